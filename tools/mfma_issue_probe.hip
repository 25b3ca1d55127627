// Issue-rate probe (dev tool): one wave per SIMD, 9 independent 32x32x2 f32 accumulators, STEPS k-steps per row.
// Variants add, between the MFMAs of a step, the instruction kinds the weight-gradient kernel has there:
//   bit0: operands come from LDS (10 ds_read per step, read one step ahead)    bit1: two ds_write_b64 per step
//   bit2: a handful of VALU ops per step                                       bit3: reads issued right before use
// Prints shader cycles per MFMA (s_memtime) -- 64 is the pipe's rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int rows, int pitch) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* ring = smem + wid * 8448;
  for (int i = lane; i < 8448; i += 64) ring[i] = 0.001f * (i % 13);
  __syncthreads();
  f32x16 acc[9];
  for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int base = (lane & 31) * pitch + (lane >> 5);
  float cav = 1.f + lane, cbv[9];
  for (int t = 0; t < 9; ++t) cbv[t] = 0.5f * t + lane;
  float bsum = 0.f;
  int wofs = lane * 2;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int row = 0; row < rows; ++row) {
    const int rb0 = base + ((row + 0) & 3) * 1376, rb1 = base + ((row + 1) & 3) * 1376, rb2 = base + ((row + 2) & 3) * 1376;
    const float* da = ring + 5504 + (row & 1) * 1312 + base;
#pragma unroll
    for (int s = 0; s < 20; ++s) {
      float nav = cav, nbv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) nbv[t] = cbv[t];
      if ((V & 1) && (V & 8)) {     // reads right before use
        cav = da[2 * s];
#pragma unroll
        for (int t = 0; t < 9; ++t) cbv[t] = ring[(t / 3 == 0 ? rb0 : t / 3 == 1 ? rb1 : rb2) + t % 3 + 2 * s];
        nav = cav;
#pragma unroll
        for (int t = 0; t < 9; ++t) nbv[t] = cbv[t];
      }
      bsum += cav;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[0], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if ((V & 1) && !(V & 8)) {
        nav = da[2 * ((s + 1) % 20)];
        for (int t = 0; t < 3; ++t) nbv[t] = ring[rb0 + t + 2 * ((s + 1) % 20)];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[1], acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if ((V & 1) && !(V & 8))
        for (int t = 3; t < 6; ++t) nbv[t] = ring[rb1 + t - 3 + 2 * ((s + 1) % 20)];
      __builtin_amdgcn_sched_barrier(0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[2], acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if ((V & 1) && !(V & 8))
        for (int t = 6; t < 9; ++t) nbv[t] = ring[rb2 + t - 6 + 2 * ((s + 1) % 20)];
      __builtin_amdgcn_sched_barrier(0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[3], acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (V & 4) {
        wofs = (wofs * 3 + s) & 1023;
        asm volatile("" : "+v"(wofs));
        wofs = wofs > 512 ? wofs - 7 : wofs + 11;
        asm volatile("" : "+v"(wofs));
        wofs = (wofs ^ (s * 8)) & 1022;
        asm volatile("" : "+v"(wofs));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[4], acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[5], acc[5], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (V & 2) {
        u32x2 v = {(unsigned)s, (unsigned)row};
        *reinterpret_cast<u32x2*>(ring + ((row + 3) & 3) * 1376 + ((wofs + 64 * s) & 1022)) = v;
        *reinterpret_cast<u32x2*>(ring + 5504 + ((row + 1) & 1) * 1312 + ((wofs + 60 * s) & 1022)) = v;
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[6], acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[7], acc[7], 0, 0, 0);
      acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[8], acc[8], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      cav = nav;
#pragma unroll
      for (int t = 0; t < 9; ++t) cbv[t] = nbv[t];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sacc = bsum + wofs;
  for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) sacc += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
  if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
}
static int cmp(const void* a, const void* b) {
  unsigned long long x = *(const unsigned long long*)a, y = *(const unsigned long long*)b;
  return x < y ? -1 : x > y;
}
template <int V>
void run(int pitch, const char* tag) {
  const int blocks = 256, rows = 10;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
  const int lds = 4 * 8448 * 4;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) k<V><<<blocks, 256, lds>>>(out, cyc, rows, pitch);
  hipDeviceSynchronize();
  unsigned long long h[1024];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  qsort(h, 1024, 8, cmp);
  printf("V=%2d pitch=%d %-44s cycles/MFMA median %.1f  (p10 %.1f p90 %.1f)\n", V, pitch, tag, h[512] / (rows * 180.0),
         h[102] / (rows * 180.0), h[921] / (rows * 180.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>(42, "registers only");
  run<4>(42, "+ VALU");
  run<1>(42, "+ LDS reads one step ahead");
  run<9>(42, "+ LDS reads right before use");
  run<3>(42, "+ LDS reads ahead + ds_write_b64 x2");
  run<7>(42, "+ reads ahead + writes + VALU");
  run<1>(43, "+ LDS reads ahead, odd pitch");
  return 0;
}
