// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (dev probe)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 1e-7f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
static float run(int flags, float* o1, float* o2, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, 0, nullptr, nullptr, 0, o1, iters);
    hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, 0, nullptr, nullptr, flags, o2, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e3f;
}
int main() {
  float *o1, *o2; hipMalloc(&o1, 128 * 256 * 4); hipMalloc(&o2, 128 * 256 * 4);
  int iters = 40000;
  printf("two half-chip kernels, ordered     : %.1f us\n", run(0, o1, o2, iters));
  printf("two half-chip kernels, any-order   : %.1f us\n", run(hipExtAnyOrderLaunch, o1, o2, iters));
  printf("ordered again                      : %.1f us\n", run(0, o1, o2, iters));
  return 0;
}
