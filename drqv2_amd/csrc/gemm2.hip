// LDS-free f32 GEMM for the hidden_dim x hidden_dim layers of the policy / Q MLPs (drqv2.py:77-81,97-101) when
// every dimension is a multiple of 32 (the training shapes: batch 256/512, hidden 1024).
//
// gemm.hip stages both operands through LDS with one barrier per k-tile; at M = batch = 256 the block tiles are
// small (32x32, to fill 256 CUs) and its LDS traffic and its MFMAs end up back to back instead of overlapped
// (45-50 % of the MFMA rate whatever the tile shape or k-tile).  Here a wave owns a 32x32 output tile and feeds
// v_mfma_f32_32x32x2_f32 straight from global memory, the way the conv kernel does:
//   * lane (i = lane&31, h = lane>>5) holds row i of the A tile / column i of the B tile and, in each 32-long
//     k-step, the 16 consecutive k values [16h, 16h+16).  MFMA e of the step multiplies the e-th of them, i.e.
//     reduces k = e and k = 16+e: any k order is a valid reduction as long as A and B use the same one.
//   * k-contiguous operand (x[m][k], W[n][k]): four 16-byte buffer loads per lane and step; a lane reads 64 B
//     of its row, the two halves of the wave cover the whole 128-byte line in the same instruction pair.
//   * row-contiguous operand (W[k][n] in dgrad, dy[k][m] and x[k][n] in wgrad): 16 dword buffer loads per step,
//     each two full 128-byte row segments.
//   * no LDS, no barrier in the main loop; the loads of step s+1 are issued before the MFMAs of step s and
//     4-5 waves per SIMD cover the rest.
// K can be split over the 4 waves of a workgroup (KS = 4, summed in wave order through LDS: deterministic) so
// that M*N/1024 tiles x 4 waves fill the chip.  Epilogue: bias, ReLU, ReLU mask, and the bias gradient of the
// wgrad form (row sums of A).  drq_gemm_batched_f32 (gemm.hip) routes eligible calls here.
#include "common.h"

namespace {

constexpr int MAXB2 = 8;

struct G2Args {
  const float* A[MAXB2];
  const float* B[MAXB2];
  float* C[MAXB2];
  const float* bias[MAXB2];
  const float* aux[MAXB2];
  float* rowsum[MAXB2];
  long lda, ldb, ldc;
  int ldaux;
  int M, N, K;
  int relu;
  unsigned a_bytes, b_bytes;
};

__device__ __forceinline__ int rowmap2(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// one k-step (32 values of k) of one operand tile -> 16 registers
template <bool KC>
__device__ __forceinline__ void load_step(float (&v)[16], __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff,
                                          unsigned ldbytes) {
  if constexpr (KC) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + q * 16, 0));
      const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
      v[q * 4 + 0] = t0; v[q * 4 + 1] = t1; v[q * 4 + 2] = t2; v[q * 4 + 3] = t3;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + e * ldbytes, 0));
  }
}

template <bool A_KC, bool B_KC, int KS>
__global__ __launch_bounds__(256) void gemm2_kernel(G2Args g) {
  __shared__ float red[KS > 1 ? 3 * 16 * 64 : 1];
  __shared__ float rsred[KS > 1 ? 3 * 32 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int batch = blockIdx.z;
  const int NT = g.N >> 5;
  const int tile = KS > 1 ? blockIdx.x : blockIdx.x * 4 + wid;
  const int ntiles = (g.M >> 5) * NT;
  const bool live = tile < ntiles;                 // KS == 1: the last workgroup may hold fewer than 4 tiles
  const int tc = live ? tile : ntiles - 1;
  const int mt = tc / NT, nt = tc - mt * NT;
  const int m0 = mt * 32, n0 = nt * 32;
  const int klen = g.K / KS;                       // multiple of 32 (checked by the launcher)
  const int kbeg = KS > 1 ? wid * klen : 0;
  const int nsteps = klen >> 5;

  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g.A[batch], 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)g.B[batch], 0, g.b_bytes, 0x00020000);
  const unsigned lda4 = (unsigned)g.lda * 4, ldb4 = (unsigned)g.ldb * 4;
  // per-lane byte offset of the lane's first element of step 0; the step advances through the scalar offset
  const unsigned avoff = A_KC ? ((unsigned)(m0 + col) * lda4 + (unsigned)(kbeg + half * 16) * 4)
                              : ((unsigned)(kbeg + half * 16) * lda4 + (unsigned)(m0 + col) * 4);
  const unsigned bvoff = B_KC ? ((unsigned)(n0 + col) * ldb4 + (unsigned)(kbeg + half * 16) * 4)
                              : ((unsigned)(kbeg + half * 16) * ldb4 + (unsigned)(n0 + col) * 4);
  const unsigned astep = A_KC ? 128u : 32u * lda4;   // bytes per k-step
  const unsigned bstep = B_KC ? 128u : 32u * ldb4;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float rs = 0.f;                                  // row sum of A (bias gradient), this lane's k values
  const bool do_rs = !A_KC && g.rowsum[batch] != nullptr && nt == 0;

  float a0[16], b0[16], a1[16], b1[16];
  auto mfma_step = [&](const float (&a)[16], const float (&b)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    if constexpr (!A_KC) {
      if (do_rs) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rs += a[e];
      }
    }
  };
  load_step<A_KC>(a0, ars, avoff, 0u, lda4);
  load_step<B_KC>(b0, brs, bvoff, 0u, ldb4);
  int s = 0;
  for (; s + 2 <= nsteps; s += 2) {
    load_step<A_KC>(a1, ars, avoff, (unsigned)(s + 1) * astep, lda4);
    load_step<B_KC>(b1, brs, bvoff, (unsigned)(s + 1) * bstep, ldb4);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(a0, b0);
    const int sn = s + 2 < nsteps ? s + 2 : s + 1;        // past the end: harmless re-load
    load_step<A_KC>(a0, ars, avoff, (unsigned)sn * astep, lda4);
    load_step<B_KC>(b0, brs, bvoff, (unsigned)sn * bstep, ldb4);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(a1, b1);
  }
  if (s < nsteps) mfma_step(a0, b0);

  if constexpr (!A_KC) {
    if (do_rs) rs += __shfl_xor(rs, 32);           // the two k halves of row m0+col
  }

  // ---- combine the K split in wave order
  if constexpr (KS > 1) {
    if (wid > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wid - 1) * 16 + r) * 64 + lane] = acc[r];
      if (lane < 32) rsred[(wid - 1) * 32 + lane] = rs;
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      v += red[(0 * 16 + r) * 64 + lane];
      v += red[(1 * 16 + r) * 64 + lane];
      v += red[(2 * 16 + r) * 64 + lane];
      acc[r] = v;
    }
    rs = ((rs + rsred[col]) + rsred[32 + col]) + rsred[64 + col];
  }
  if (!live) return;

  // ---- epilogue: C/D layout col = lane&31 (n), rows (r&3) + 8*(r>>2) + 4*half (m)
  const int n = n0 + col;
  const float bv = g.bias[batch] ? g.bias[batch][n] : 0.f;
  float mk[16];
  const float* ap = g.aux[batch];
#pragma unroll
  for (int r = 0; r < 16; ++r) mk[r] = ap ? ap[(long)(m0 + rowmap2(r, half)) * g.ldaux + n] : 1.f;
  float* c = g.C[batch];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bv;
    if (g.relu) v = v > 0.f ? v : 0.f;
    v = mk[r] > 0.f ? v : 0.f;
    c[(long)(m0 + rowmap2(r, half)) * g.ldc + n] = v;
  }
  if constexpr (!A_KC) {
    if (do_rs && half == 0) g.rowsum[batch][m0 + col] = rs;
  }
}

template <bool A_KC, bool B_KC>
int launch2(const G2Args& g, int nbatch, hipStream_t st) {
  const int tiles = (g.M / 32) * (g.N / 32);
  // split K over the waves of a workgroup when the tiles alone leave the chip under-filled
  const bool split = (long)tiles * nbatch < 4L * drq_num_cus() * 2 && g.K % 128 == 0 && g.K >= 512;
  if (split) {
    hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, 4>), dim3(tiles, 1, nbatch), dim3(256), 0, st, g);
  } else {
    hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, 1>), dim3((tiles + 3) / 4, 1, nbatch), dim3(256), 0, st, g);
  }
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // namespace

// Returns DRQ_EARG when the problem is not eligible (the caller then uses the LDS-tiled kernel).
int drq_gemm2(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb, int b_kc,
              float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
              const float* const* aux, int ldaux, float* const* rowsum, hipStream_t st) {
  if (nbatch <= 0 || nbatch > MAXB2 || M % 32 || N % 32 || K % 32 || K < 64 || M < 32 || N < 32) return DRQ_EARG;
  if (!a_kc && b_kc) return DRQ_EARG;
  if (rowsum && a_kc) return DRQ_EARG;
  // forward form (both operands k-contiguous): every 16-byte lane load touches its own cache line, 32 lines per
  // instruction, and the texture addresser becomes the limit (measured 37 vs 31 us on 4 x 256x1024x1024); the
  // LDS-tiled kernel keeps that form.  With one row-contiguous operand this kernel is 1.4-1.6x faster.
  if (a_kc && b_kc) return DRQ_EARG;
  const size_t ab = (a_kc ? (size_t)M * lda : (size_t)K * lda) * 4, bb = (b_kc ? (size_t)N * ldb : (size_t)K * ldb) * 4;
  if (ab >= (1ull << 31) || bb >= (1ull << 31)) return DRQ_EARG;
  if (a_kc && lda % 4) return DRQ_EARG;
  if (b_kc && ldb % 4) return DRQ_EARG;
  G2Args g{};
  for (int b = 0; b < nbatch; ++b) {
    if ((a_kc && ((uintptr_t)A[b] & 15)) || (b_kc && ((uintptr_t)B[b] & 15))) return DRQ_EARG;
    g.A[b] = A[b]; g.B[b] = B[b]; g.C[b] = C[b];
    g.bias[b] = bias ? bias[b] : nullptr;
    g.aux[b] = aux ? aux[b] : nullptr;
    g.rowsum[b] = rowsum ? rowsum[b] : nullptr;
  }
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.M = M; g.N = N; g.K = K; g.relu = relu;
  g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  if (a_kc && b_kc) return launch2<true, true>(g, nbatch, st);
  if (a_kc && !b_kc) return launch2<true, false>(g, nbatch, st);
  return launch2<false, false>(g, nbatch, st);
}
