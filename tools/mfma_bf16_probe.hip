// bf16 MFMA sustained rate on gfx950 (register-only loop), next to the f32 MFMA: data for the "three-way bf16 split"
// option discussed in DESIGN.md (fp32-accurate products from 6 bf16 MFMAs).  Dev probe.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void kb(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed * (threadIdx.x % 7 + e + 1)); b[e] = (__bf16)(seed * (threadIdx.x % 5 + e + 2)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    dim3 grid(256 * bpc), blk(256); int iters = 20000;
    float ms = 0, total = 0; int n = 0;
    while (total < 1500.f) {
      hipEventRecord(e0); kb<4><<<grid, blk>>>(out, iters, 0.01f); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); total += ms; ++n;
    }
    double flop = (double)grid.x * 4 * iters * 8 * 4 * 2.0 * 32 * 32 * 16;
    printf("bf16 32x32x16, 4 chains, %d wave(s)/SIMD: %.0f TFLOP/s sustained (%.2f ms per launch); /6 = %.0f TF fp32-equivalent\n",
           bpc, flop / ms / 1e9, ms, flop / ms / 1e9 / 6);
  }
  return 0;
}
