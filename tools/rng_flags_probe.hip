// Which hipcc flags reproduce torch's normal_() rounding?  out[i] = first value of hiprand_normal4 of Philox(seed, i, offset).
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>
__global__ void k(unsigned long long seed, unsigned long long off, int n, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  hiprandStatePhilox4_32_10_t st;
  hiprand_init(seed, (unsigned long long)i, off, &st);
  const float4 z = hiprand_normal4(&st);
  out[i] = z.x * 1.0f + 0.0f;
}
extern "C" __attribute__((visibility("default"))) int probe(unsigned long long seed, unsigned long long off, int n, float* out) {
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, seed, off, n, out);
  return (int)hipDeviceSynchronize();
}
