// f32 MFMA peak probe (dev tool): register-only v_mfma_f32_32x32x2_f32 loops, NACC independent accumulators
// per wave, W waves per SIMD.  Prints TFLOP/s for random-ish and zero operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 * (threadIdx.x % 7 + 1), b = b0 * (threadIdx.x % 5 + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, float a0, float b0, const char* tag) {
  int cus = 256; float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
  int iters = 20000 / NACC;
  dim3 grid(cus * blocks_per_cu), blk(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<grid, blk>>>(out, 100, a0, b0); hipDeviceSynchronize();
  hipEventRecord(e0); k<NACC><<<grid, blk>>>(out, iters, a0, b0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)grid.x * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("%-8s NACC=%d waves/SIMD=%d : %.1f TFLOP/s (%.2f ms)\n", tag, NACC, blocks_per_cu, flop / ms / 1e9, ms);
  hipFree(out);
}
// conv-shaped launch: 512-thread workgroups, LDSF floats of LDS each, one accumulator chain per wave
template <int LDSF>
__global__ __launch_bounds__(512, 4) void kc(float* out, int iters, float a0, float b0) {
  __shared__ float l[LDSF];
  for (int i = threadIdx.x; i < LDSF; i += 512) l[i] = a0 * i;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = l[threadIdx.x], b = b0 * (threadIdx.x % 5 + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int LDSF>
void convshape(int grid, int iters) {
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kc<LDSF><<<grid, 512>>>(out, 10, 0.7f, 1.3f); hipDeviceSynchronize();
  float best = 1e9, ms;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); kc<LDSF><<<grid, 512>>>(out, iters, 0.7f, 1.3f); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double flop = (double)grid * 8 * iters * 16 * 2.0 * 32 * 32 * 2;
  printf("convshape LDS=%3d KB grid=%4d iters=%5d : %.1f TFLOP/s (%.3f ms)\n", LDSF * 4 / 1024, grid, iters, flop / best / 1e9, best);
  hipFree(out);
}
// random operands that change with every MFMA (registers only, no VALU in the loop)
__global__ __launch_bounds__(256) void kr(float* out, const float* rnd, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[16], b[16];
  for (int u = 0; u < 16; ++u) { a[u] = rnd[(threadIdx.x * 16 + u) % 8192]; b[u] = rnd[(threadIdx.x * 16 + u + 4096) % 8192]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 15], b[(u + 5 * i) & 15], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void randops(int blocks_per_cu) {
  float *out, *rnd; hipMalloc(&out, 256 * 1024 * 4 * 8); hipMalloc(&rnd, 8192 * 4);
  float h[8192]; unsigned x = 12345u;
  for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23) * 1e-3f; }
  hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  dim3 grid(256 * blocks_per_cu), blk(256); int iters = 2500;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, total = 0, first = 0; int n = 0;
  while (total < 3000.f) {
    hipEventRecord(e0); kr<<<grid, blk>>>(out, rnd, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); total += ms; if (n == 1) first = ms; ++n;
  }
  double flop = (double)grid.x * 4 * iters * 64 * 2.0 * 32 * 32 * 2;
  printf("randops waves/SIMD=%d : first %.1f TFLOP/s, after 3 s %.1f TFLOP/s (%.2f ms)\n", blocks_per_cu, flop / first / 1e9, flop / ms / 1e9, ms);
}
template <int NACC>
void sustained(int blocks_per_cu) {   // >= 3 s of back-to-back launches on pseudo-random operands, then time one launch
  int cus = 256; float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
  int iters = 40000 / NACC; dim3 grid(cus * blocks_per_cu), blk(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, total = 0; int n = 0;
  while (total < 3000.f) {
    hipEventRecord(e0); k<NACC><<<grid, blk>>>(out, iters, 0.73519f, 1.31873f); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); total += ms; ++n;
  }
  double flop = (double)grid.x * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("sustained NACC=%d waves/SIMD=%d : %.1f TFLOP/s after %d launches (%.2f ms each)\n", NACC, blocks_per_cu, flop / ms / 1e9, n, ms);
  hipFree(out);
}
int main() {
  randops(1); randops(2);
  convshape<9216>(512, 54); convshape<9216>(512, 5400); convshape<15360>(512, 54); convshape<15360>(512, 5400);
  convshape<9216>(256, 108); convshape<9216>(1024, 27); convshape<9216>(2048, 14); convshape<9216>(3042, 9);

  run<1>(1, 1.0f, 0.5f, "rand"); run<4>(1, 1.0f, 0.5f, "rand"); run<1>(2, 1.0f, 0.5f, "rand"); run<1>(4, 1.0f, 0.5f, "rand");
  run<4>(2, 1.0f, 0.5f, "rand"); run<4>(1, 0.f, 0.f, "zero"); run<1>(4, 0.f, 0.f, "zero");
  return 0;
}
