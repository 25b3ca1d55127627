// Shared declarations for the DrQ-v2 HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The library is built with -fvisibility=hidden: only the entry points declared in include/drqv2_hip.h carry
// DRQ_API and are exported; everything else (kernels' host stubs, cross-file helpers) stays internal.
#define DRQ_API __attribute__((visibility("default")))

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

// Per-device host-side caches (the CU count, "kernel attribute already set" flags) are indexed by the current HIP
// device: an agent on cuda:1 must not reuse what was established for cuda:0.
static constexpr int kMaxDevices = 64;
static inline int drq_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  return dev;
}

static inline int drq_num_cus() {
  static int n[kMaxDevices] = {};
  const int dev = drq_device();
  if (n[dev] == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    n[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return n[dev];
}
