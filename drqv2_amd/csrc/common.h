// Shared declarations for the DrQ-v2 HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRQ_OK 0
#define DRQ_EARG (-1)      // bad argument / unsupported shape
#define DRQ_EWS (-2)       // workspace too small

// every launcher returns 0 or the hipError_t of the launch (positive)
#define DRQ_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// Encoder geometry (drqv2.py:55-59): 84 -(k3,s2)-> 41 -> 39 -> 37 -> 35, 32 channels.
static constexpr int kEncH[5] = {84, 41, 39, 37, 35};
static constexpr int kCout = 32;

static inline int drq_num_cus() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    n = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return n;
}
