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

// Metrics mirror (DrqStep.sums_host): eight floats and a sequence word in pinned (fine-grained, uncached) host memory,
// written by ONE lane.  A system-scope release before the sequence word would write back every dirty line of the XCD's
// L2 first (it is what makes OTHER threads' cached stores visible) -- measured: 3.5 us of the publishing launch at batch
// 256, 66 us at batch 2,048, on the path the host waits on.  Nothing cached has to become visible here: the lane's own
// system-scope stores go straight out; they are drained (vmcnt(0)) before the sequence word follows them down the same
// ordered path, and the "memory" clobbers keep the compiler from moving either across the wait.
#if defined(__HIPCC__)
__device__ __forceinline__ void drq_publish_mirror(float* host, const float (&v)[8], unsigned seq) {
#pragma unroll
  for (int i = 0; i < 8; ++i) __hip_atomic_store(host + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(reinterpret_cast<unsigned*>(host + 8), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("" ::: "memory");
}
#endif
