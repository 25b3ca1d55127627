// LDS-free f32 GEMM for the hidden_dim x hidden_dim layers of the policy / Q MLPs (drqv2.py:77-81,97-101) when
// every dimension is a multiple of 32 (the training shapes: batch 256/512, hidden 1024).
//
// gemm.hip stages both operands through LDS with one barrier per k-tile; at M = batch = 256 the block tiles are
// small (32x32, to fill 256 CUs) and its LDS traffic and its MFMAs end up back to back instead of overlapped
// (45-50 % of the MFMA rate whatever the tile shape or k-tile).  Here a wave owns a 32x32 output tile and feeds
// v_mfma_f32_32x32x2_f32 straight from global memory, the way the conv kernel does:
//   * lane (i = lane&31, h = lane>>5) holds row i of the A tile / column i of the B tile and, in each 32-long
//     k-step, the 16 k values 8q + 4h + {0..3}, q = 0..3.  MFMA e = 4q + j of the step multiplies the e-th of
//     them, i.e. reduces k = 8q + j and k = 8q + 4 + j: any k order is a valid reduction as long as A and B use
//     the same one.
//   * k-contiguous operand (x[m][k], W[n][k]): four 16-byte buffer loads per lane and step; load q of the two
//     lanes of a row covers 32 contiguous bytes, so an instruction touches one 64-byte sector per row (with the
//     lane's 16 values contiguous instead, the two lanes hit both sectors of the line and the L1 does twice the
//     sector reads).
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
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + q * 32, 0));
      const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
      v[q * 4 + 0] = t0; v[q * 4 + 1] = t1; v[q * 4 + 2] = t2; v[q * 4 + 3] = t3;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      v[e] = __uint_as_float(
          __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + (8 * (e >> 2) + (e & 3)) * ldbytes, 0));
  }
}

// bx / batch: the workgroup's position (blockIdx.x / blockIdx.z of a plain launch; gemm2_pair_kernel remaps them)
template <bool A_KC, bool B_KC, int KS>
__device__ __forceinline__ void gemm2_body(const G2Args& g, int bx, int batch) {
  __shared__ float red[KS > 1 ? 3 * 16 * 64 : 1];
  __shared__ float rsred[KS > 1 ? 3 * 32 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int NT = g.N >> 5;
  const int tile = KS > 1 ? bx : bx * 4 + wid;
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
  const unsigned avoff = A_KC ? ((unsigned)(m0 + col) * lda4 + (unsigned)(kbeg + half * 4) * 4)
                              : ((unsigned)(kbeg + half * 4) * lda4 + (unsigned)(m0 + col) * 4);
  const unsigned bvoff = B_KC ? ((unsigned)(n0 + col) * ldb4 + (unsigned)(kbeg + half * 4) * 4)
                              : ((unsigned)(kbeg + half * 4) * ldb4 + (unsigned)(n0 + col) * 4);
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

template <bool A_KC, bool B_KC, int KS>
__global__ __launch_bounds__(256) void gemm2_kernel(G2Args g) {
  gemm2_body<A_KC, B_KC, KS>(g, (int)blockIdx.x, (int)blockIdx.z);
}

// The weight gradient (dW = dy^T x, both operands row-contiguous) and the input gradient (dx = dy W, masked) of ONE
// layer read the same dy and do not depend on each other: one launch runs both (workgroups [0, nxw) the first,
// the rest the second) instead of two dependent launches of ~15 us each -- a launch boundary less per layer, and
// the tail of one problem overlaps the other.  Same bodies as gemm2_kernel: results are bit-identical.
template <int KSW, int KSD>
__global__ __launch_bounds__(256) void gemm2_pair_kernel(G2Args gw, G2Args gd, int nxw) {
  if ((int)blockIdx.x < nxw) gemm2_body<false, false, KSW>(gw, (int)blockIdx.x, (int)blockIdx.z);
  else gemm2_body<true, false, KSD>(gd, (int)blockIdx.x - nxw, (int)blockIdx.z);
}

inline bool gemm2_split(const G2Args& g, int nbatch) {
  const int tiles = (g.M / 32) * (g.N / 32);
  // split K over the waves of a workgroup when the tiles alone leave the chip under-filled
  return (long)tiles * nbatch < 4L * drq_num_cus() * 2 && g.K % 128 == 0 && g.K >= 512;
}

template <bool A_KC, bool B_KC>
int launch2(const G2Args& g, int nbatch, hipStream_t st) {
  const int tiles = (g.M / 32) * (g.N / 32);
  const bool split = gemm2_split(g, nbatch);
  if (split) {
    hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, 4>), dim3(tiles, 1, nbatch), dim3(256), 0, st, g);
  } else {
    hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, 1>), dim3((tiles + 3) / 4, 1, nbatch), dim3(256), 0, st, g);
  }
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// ---- trunk forward: z = feat W^T with K = repr_dim (39200), N = feature_dim (50) (drqv2.py:71-72,91-92) --------
// Both operands are k-contiguous and N fits two 32-column tiles (four for feature_dim 100, with TM = 1), so a wave
// takes TM row tiles x all column tiles and a slice of K: per 32-long k-step it loads 4*(TM+2) 16-byte pieces per lane for 32*TM MFMAs -- half (TM = 1)
// or a third (TM = 2) of the loads per MFMA of the one-tile form above, which the texture addresser keeps up
// with.  Weight rows >= N are out of the buffer's range and read as zero.  K is cut into gridDim.x slices
// (whole k-steps, sizes differing by at most one step); the four waves of a workgroup sum their shares in wave
// order through LDS and the workgroup writes ONE partial record [M][N] per (problem, blockIdx.x) in the split-K
// layout of gemm.hip, which the LayerNorm kernel sums (+ bias) in a fixed order.  No VALU work in the k loop.
struct TrunkArgs {
  const float* A[MAXB2];
  const float* B[MAXB2];
  float* part;               // [nbatch][gridDim.x][M][N]
  long lda, ldb;
  int M, N, K;
  unsigned a_bytes, b_bytes;
};

template <int TM, int TN = 2>
__global__ __launch_bounds__(256, 1) void trunk_fwd_kernel(TrunkArgs g) {
  __shared__ float red[3 * TM * TN * 16 * 64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int batch = blockIdx.z;
  const int m0 = blockIdx.y * (32 * TM);
  const int steps = g.K >> 5;
  // the workgroup's slice of k-steps, dealt round-robin to its four waves: at any time the workgroup reads
  // 512 contiguous bytes of each row instead of four distant 128-byte pieces
  const int nsl = (int)gridDim.x, sl = (int)blockIdx.x;
  const int per = steps / nsl, rem = steps - per * nsl;
  const int s0 = sl * per + (sl < rem ? sl : rem) + wid;
  const int len = per + (sl < rem ? 1 : 0);
  const int ns = len > wid ? (len - wid + 3) / 4 : 0;

  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g.A[batch], 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)g.B[batch], 0, g.b_bytes, 0x00020000);
  const unsigned lda4 = (unsigned)g.lda * 4, ldb4 = (unsigned)g.ldb * 4;
  unsigned avoff[TM], bvoff[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) avoff[i] = (unsigned)(m0 + i * 32 + col) * lda4 + (unsigned)(half * 4) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) bvoff[j] = (unsigned)(j * 32 + col) * ldb4 + (unsigned)(half * 4) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // two register stages (a third, i.e. loads issued two k-steps ahead, measured slower: 53.6 vs 51.4 us)
  float a0[TM][16], b0[TN][16], a1[TM][16], b1[TN][16];
  auto load = [&](float (&a)[TM][16], float (&b)[TN][16], int s) {
    const int sc = s < ns ? s : ns - 1;                   // past the end: harmless re-load of the last step
    const unsigned so = (unsigned)(s0 + 4 * sc) * 128u;
#pragma unroll
    for (int i = 0; i < TM; ++i) load_step<true>(a[i], ars, avoff[i], so, lda4);
#pragma unroll
    for (int j = 0; j < TN; ++j) load_step<true>(b[j], brs, bvoff[j], so, ldb4);
  };
  auto mfma_step = [&](const float (&a)[TM][16], const float (&b)[TN][16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };
  // the same step with the NEXT stage's loads spread over it, (TM+TN)/4 ahead of every MFMA group: a wave issues in
  // order and there is one wave per SIMD -- a burst of 12-20 loads that backs up in the texture addresser would hold
  // the MFMAs behind it
  auto mfma_step_load = [&](const float (&a)[TM][16], const float (&b)[TN][16], float (&na)[TM][16],
                            float (&nb)[TN][16], int s) {
    const int sc = s < ns ? s : ns - 1;
    const unsigned so = (unsigned)(s0 + 4 * sc) * 128u;
    constexpr int NL = (TM + TN) * 4;        // 16-byte loads of a stage: tile l>>2 (A tiles first), piece l&3
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int l = (e * NL) / 16; l < ((e + 1) * NL) / 16; ++l) {
        const int tile = l >> 2, q = l & 3;
        f32x4 t;
        if (tile < TM) t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff[tile], so + q * 32, 0));
        else t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, bvoff[tile - TM], so + q * 32, 0));
        const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        float* dst = tile < TM ? &na[tile][q * 4] : &nb[tile - TM][q * 4];
        dst[0] = t0; dst[1] = t1; dst[2] = t2; dst[3] = t3;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (ns > 0) {
    load(a0, b0, 0);
    int s = 0;
    if constexpr (TM == 1) {
      // one row tile per wave (the single trunk of the actor step, feature_dim 100): few MFMAs per load, the spread
      // form wins (21.5 -> 18.4 us); with two row tiles (the four trunks of the critic step) the burst is as good
      for (; s + 2 <= ns; s += 2) {
        mfma_step_load(a0, b0, a1, b1, s + 1);
        mfma_step_load(a1, b1, a0, b0, s + 2);
      }
    } else {
      for (; s + 2 <= ns; s += 2) {
        load(a1, b1, s + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a0, b0);
        load(a0, b0, s + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a1, b1);
      }
    }
    if (s < ns) mfma_step(a0, b0);
  }

  // ---- sum the four slices of the workgroup in wave order, write the partial record
  if (wid > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wid - 1) * TM + i) * TN + j) * 1024 + r * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wid > 0) return;
  float* out = g.part + ((long)batch * gridDim.x + blockIdx.x) * g.M * g.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += red[((w * TM + i) * TN + j) * 1024 + r * 64 + lane];
        if (n < g.N) out[(long)(m0 + i * 32 + rowmap2(r, half)) * g.N + n] = v;
      }
    }
}

// ---- trunk weight gradient: dW = dz^T feat, M = feature_dim (50), N = repr_dim (39200), K = batch ------------------
// A wave keeps ITS 32 rows of dz^T for the whole batch in registers (K/32 steps x 16 values) and walks a contiguous
// range of 32-column tiles of feat: 16 dword loads (two full 128-byte row segments each) per 16 MFMAs, issued three
// k-steps ahead through four rotating register stages that run on across tile boundaries; no LDS, no barrier, no
// VALU in the loop.  Waves 2i and 2i+1 take the two row tiles of the same columns (the second finds feat in L1).
struct TWArgs {
  const float* A;            // dz   [K][lda]
  const float* B;            // feat [K][ldb]
  float* C;                  // dW   [M][ldc]
  float* rowsum;             // db   [M] or null
  long lda, ldb, ldc;
  int M, N, K;
  unsigned a_bytes, b_bytes;
  // rider (optional, ln_rows > 0): ONE extra workgroup computes the LayerNorm parameter gradients of the same trunk
  // (dgamma[f] = sum_r dln[r][f] xhat[r][f], dbeta[f] = sum_r dln[r][f]) beside the weight-gradient workgroups -- they
  // only need what the LayerNorm backward just wrote, and as a launch of their own they cost ~5 us of a serial stream
  const float* ln_dln;
  const float* ln_xhat;
  float* ln_dgamma;
  float* ln_dbeta;
  int ln_rows, ln_F;
};

__device__ __forceinline__ void ln_param_rider(const TWArgs& g) {
  __shared__ float sg[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rows = g.ln_rows, F = g.ln_F;
  for (int f0 = 0; f0 < F; f0 += 64) {
    const int f = f0 + lane;
    const bool ok = f < F;
    const int fc = ok ? f : F - 1;
    float gs = 0.f, bs = 0.f;
    for (int r0 = wid; r0 < rows; r0 += 4 * 16) {       // 16 rows in flight per thread, then a fixed-order sum
      float d[16], x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int rr = min(r0 + 4 * u, rows - 1);
        d[u] = g.ln_dln[(long)rr * F + fc];
        x[u] = g.ln_xhat[(long)rr * F + fc];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (r0 + 4 * u < rows) {
          gs += d[u] * x[u];
          bs += d[u];
        }
    }
    sg[wid][lane] = gs;
    sb[wid][lane] = bs;
    __syncthreads();
    if (wid == 0 && ok) {
      g.ln_dgamma[f] = ((sg[0][lane] + sg[1][lane]) + sg[2][lane]) + sg[3][lane];
      g.ln_dbeta[f] = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
    }
    __syncthreads();
  }
}

template <int KSTEPS>
__global__ __launch_bounds__(256, 1) void trunk_wgrad_kernel(TWArgs g) {
  constexpr int NS = 4, D = 3;               // register stages, prefetch distance (k-steps)
  static_assert(KSTEPS % NS == 0, "stage index must be a compile-time constant");
  const int nblk = (int)gridDim.x - (g.ln_rows > 0 ? 1 : 0);     // the last workgroup is the rider's
  if (g.ln_rows > 0 && (int)blockIdx.x == nblk) {
    ln_param_rider(g);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int mtiles = (g.M + 31) >> 5, ntiles = g.N >> 5;
  const int gw = (int)blockIdx.x * 4 + wid, nw = nblk * 4;
  const int mt = gw % mtiles, grp = gw / mtiles, ngrp = nw / mtiles;
  if (grp >= ngrp) return;
  const int per = ntiles / ngrp, rem = ntiles - per * ngrp;
  const int t0 = grp * per + (grp < rem ? grp : rem);
  const int t1 = t0 + per + (grp < rem ? 1 : 0);
  if (t0 >= t1) return;

  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, g.b_bytes, 0x00020000);
  const unsigned lda4 = (unsigned)g.lda * 4, ldb4 = (unsigned)g.ldb * 4;
  const int m = mt * 32 + col;
  // rows of dz^T past M read as zero (offset out of the buffer's range)
  const unsigned avoff = m < g.M ? (unsigned)(half * 4) * lda4 + (unsigned)m * 4 : 0x7ffffff0u;
  const unsigned bvoff = (unsigned)(half * 4) * ldb4 + (unsigned)col * 4;

  float a[KSTEPS][16];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) load_step<false>(a[s], ars, avoff, (unsigned)(s * 32) * lda4, lda4);

  float b[NS][16];
  // flat (tile, step) index i -> loads of tile t0 + i / KSTEPS, step i % KSTEPS; past the end: re-load the last step
  const int total = (t1 - t0) * KSTEPS;
  auto loadb = [&](float (&dst)[16], int i) {
    const int ic = i < total ? i : total - 1;
    const int t = t0 + ic / KSTEPS, st = ic % KSTEPS;
    load_step<false>(dst, brs, bvoff, (unsigned)t * 128u + (unsigned)(st * 32) * ldb4, ldb4);
  };
#pragma unroll
  for (int i = 0; i < D; ++i) loadb(b[i], i);

  float* c = g.C;
  for (int t = t0; t < t1; ++t) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ibase = (t - t0) * KSTEPS;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      // the sixteen loads of the step three ahead go out one per MFMA (a burst would back up in the texture addresser
      // and hold this wave's MFMAs behind it: one wave per SIMD)
      unsigned lsoff, lld;
      {
        const int i = ibase + s + D, ic = i < total ? i : total - 1;
        const int tt = t0 + ic / KSTEPS, st = ic % KSTEPS;
        lsoff = (unsigned)tt * 128u + (unsigned)(st * 32) * ldb4;
        lld = ldb4;
      }
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        b[(s + D) % NS][e] = __uint_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(brs, bvoff, lsoff + (8 * (e >> 2) + (e & 3)) * lld, 0));
        b[(s + D) % NS][e + 1] = __uint_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(brs, bvoff, lsoff + (8 * ((e + 1) >> 2) + ((e + 1) & 3)) * lld, 0));
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][e], b[s % NS][e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][e + 1], b[s % NS][e + 1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int n = t * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + rowmap2(r, half);
      if (row < g.M) c[(long)row * g.ldc + n] = acc0[r] + acc1[r];
    }
  }
  if (g.rowsum && grp == 0) {                // bias gradient: column sums of dz, from the registers of group 0
    float rs = 0.f;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) rs += a[s][e];
    rs += __shfl_xor(rs, 32);
    if (half == 0 && m < g.M) g.rowsum[m] = rs;
  }
}

}  // namespace

// internal (gemm.hip): dW [M][ldc] = A^T B with A = dz [K][lda], B = feat [K][ldb], rowsum = column sums of A.
// Returns DRQ_EARG when the problem is not eligible (the caller then uses the LDS-tiled kernel).
// ln_*: optional rider (see TWArgs): LayerNorm parameter gradients from dln / xhat [ln_rows][ln_F]; ln_rows = 0: none.
int drq_trunk_wgrad_ln(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                       float* rowsum, const float* ln_dln, const float* ln_xhat, float* ln_dgamma, float* ln_dbeta,
                       int ln_rows, int ln_F, hipStream_t st) {
  if (M < 1 || M > 128 || N < 4096 || N % 32 || (K != 128 && K != 256 && K != 512)) return DRQ_EARG;
  if (ln_rows > 0 && (!ln_dln || !ln_xhat || !ln_dgamma || !ln_dbeta || ln_F < 1 || ln_F > 256)) return DRQ_EARG;
  const size_t ab = (size_t)K * lda * 4, bb = (size_t)K * ldb * 4;
  if (ab >= (1ull << 31) || bb >= (1ull << 31)) return DRQ_EARG;
  TWArgs g{A, B, C, rowsum, lda, ldb, ldc, M, N, K, (unsigned)ab, (unsigned)bb,
           ln_dln, ln_xhat, ln_dgamma, ln_dbeta, ln_rows > 0 ? ln_rows : 0, ln_F};
  const int blocks = drq_num_cus() + (ln_rows > 0 ? 1 : 0);   // one wave per SIMD (128 + 64 operand registers per lane)
  // K = 512 (quadruped's batch): dz^T takes 256 of the wave's 512 registers; beyond that the tiled GEMM serves
  if (K == 512) hipLaunchKernelGGL((trunk_wgrad_kernel<16>), dim3(blocks), dim3(256), 0, st, g);
  else if (K == 256) hipLaunchKernelGGL((trunk_wgrad_kernel<8>), dim3(blocks), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((trunk_wgrad_kernel<4>), dim3(blocks), dim3(256), 0, st, g);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

int drq_trunk_wgrad(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                    float* rowsum, hipStream_t st) {
  return drq_trunk_wgrad_ln(A, lda, B, ldb, C, ldc, M, N, K, rowsum, nullptr, nullptr, nullptr, nullptr, 0, 0, st);
}

// internal (gemm.hip): split-K partials [nbatch * (*splitk_out)][M][N] of A[b] B[b]^T for the trunk shape.
// Returns DRQ_EARG when the problem is not eligible (the caller then uses the LDS-tiled kernel).
int drq_trunk_fwd_partial(int nbatch, const float* const* A, long lda, const float* const* B, long ldb, int M, int N,
                          int K, float* ws, size_t ws_bytes, int* splitk_out, hipStream_t st) {
  if (nbatch <= 0 || nbatch > MAXB2 || M % 32 || M < 32 || N < 1 || N > 128 || K % 32 || K < 4096) return DRQ_EARG;
  if (lda % 4 || ldb % 4 || !ws || ((uintptr_t)ws & 15)) return DRQ_EARG;
  const size_t ab = (size_t)M * lda * 4, bb = (size_t)N * ldb * 4;
  if (ab >= (1ull << 31) || bb >= (1ull << 31)) return DRQ_EARG;
  TrunkArgs g{};
  for (int b = 0; b < nbatch; ++b) {
    if (((uintptr_t)A[b] & 15) || ((uintptr_t)B[b] & 15)) return DRQ_EARG;
    g.A[b] = A[b]; g.B[b] = B[b];
  }
  g.part = ws; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  // one workgroup per CU (one wave per SIMD): measured 51 us against 59 us with two and 66 us with four for the
  // four trunks of the critic update -- more concurrent row streams cost more in the memory system than the
  // extra waves hide
  const int steps = K / 32, cus = drq_num_cus();
  const bool wide = N > 64;               // four column tiles per wave, one row tile
#ifdef DRQ_DEV
  static const char* const dbg = getenv("DRQ_TRUNK_DBG");       // development build only (tools/trunk_bench.py)
#else
  constexpr const char* dbg = nullptr;
#endif
  const bool force_tm2 = dbg && (atoi(dbg) & 4);
  const bool tm2 = !wide && M % 64 == 0 && ((long)nbatch * (M / 32) * 4 >= 64 || force_tm2);
  const int rows = tm2 ? M / 64 : M / 32;
  int blocks_k = (cus + nbatch * rows - 1) / (nbatch * rows);
  if (blocks_k * 4 > steps) blocks_k = steps / 4;
  // small batches (one or two row tiles): the consumer (LayerNorm) sums the split-K records row by row, and beyond
  // ~64 records per element that sum costs more than the extra workgroups save here
  if (blocks_k > 64) blocks_k = 64;
  if (dbg) {
    const int d = atoi(dbg);
    if (d & 1) g.a_bytes = g.b_bytes = 0;               // every load out of range: MFMA time only
    if (d & 2) blocks_k *= 2;
  }
  if (blocks_k < 2) return DRQ_EARG;      // a split count of 1 means "result in C" to the callers
  if ((size_t)nbatch * blocks_k * M * N * sizeof(float) > ws_bytes) return DRQ_EWS;
  const dim3 grid(blocks_k, rows, nbatch);
  if (wide) hipLaunchKernelGGL((trunk_fwd_kernel<1, 4>), grid, dim3(256), 0, st, g);
  else if (tm2) hipLaunchKernelGGL((trunk_fwd_kernel<2>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((trunk_fwd_kernel<1>), grid, dim3(256), 0, st, g);
  DRQ_LAUNCH_CHECK();
  if (splitk_out) *splitk_out = blocks_k;
  return DRQ_OK;
}

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
  // (re-measured with the 32-contiguous-bytes k order: 20.4 / 38.4 / 19.7 us against 16.1 / 27.5 / 15.8 us)
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

// wgrad + dgrad of one hidden layer in one launch (see gemm2_pair_kernel).  dy [Brows][Nout] (ld lddy), x [Brows][Kin],
// w [Nout][ldw]: dW [Nout][Kin] = dy^T x, db [Nout] = column sums of dy, dx [Brows][Kin] = (dy w) * (mask > 0).
// Returns DRQ_EARG when a shape is not eligible (the caller then issues the two GEMMs separately).
int drq_gemm2_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                          float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                          long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin, hipStream_t st) {
  if (nbatch <= 0 || nbatch > MAXB2 || Brows % 32 || Nout % 32 || Kin % 32 || Brows < 64 || Nout < 64 || Kin < 32)
    return DRQ_EARG;
  if (lddy % 4) return DRQ_EARG;
  G2Args gw{}, gd{};
  for (int b = 0; b < nbatch; ++b) {
    if (!dy[b] || !x[b] || !dw[b] || !w[b] || !dx[b] || ((uintptr_t)dy[b] & 15)) return DRQ_EARG;
    gw.A[b] = dy[b]; gw.B[b] = x[b]; gw.C[b] = dw[b]; gw.rowsum[b] = db ? db[b] : nullptr;
    gd.A[b] = dy[b]; gd.B[b] = w[b]; gd.C[b] = dx[b]; gd.aux[b] = mask ? mask[b] : nullptr;
  }
  // wgrad: A(m = n_out, k = row) = dy[k*lddy + m], B(k, n = k_in) = x[k*ldx + n]
  gw.lda = lddy; gw.ldb = ldx; gw.ldc = Kin; gw.M = Nout; gw.N = Kin; gw.K = Brows;
  // dgrad: A(m = row, k = n_out) = dy[m*lddy + k], B(k, n = k_in) = w[k*ldw + n]
  gd.lda = lddy; gd.ldb = ldw; gd.ldc = lddx; gd.ldaux = ldmask; gd.M = Brows; gd.N = Kin; gd.K = Nout;
  const size_t dyb = (size_t)Brows * lddy * 4, xb = (size_t)Brows * ldx * 4, wb = (size_t)Nout * ldw * 4;
  if (dyb >= (1ull << 31) || xb >= (1ull << 31) || wb >= (1ull << 31)) return DRQ_EARG;
  gw.a_bytes = (unsigned)dyb; gw.b_bytes = (unsigned)xb;
  gd.a_bytes = (unsigned)dyb; gd.b_bytes = (unsigned)wb;
  const bool sw = gemm2_split(gw, nbatch), sd = gemm2_split(gd, nbatch);
  const int tw = (gw.M / 32) * (gw.N / 32), td = (gd.M / 32) * (gd.N / 32);
  const int nxw = sw ? tw : (tw + 3) / 4, nxd = sd ? td : (td + 3) / 4;
  const dim3 grid(nxw + nxd, 1, nbatch);
  if (sw && sd) hipLaunchKernelGGL((gemm2_pair_kernel<4, 4>), grid, dim3(256), 0, st, gw, gd, nxw);
  else if (sw) hipLaunchKernelGGL((gemm2_pair_kernel<4, 1>), grid, dim3(256), 0, st, gw, gd, nxw);
  else if (sd) hipLaunchKernelGGL((gemm2_pair_kernel<1, 4>), grid, dim3(256), 0, st, gw, gd, nxw);
  else hipLaunchKernelGGL((gemm2_pair_kernel<1, 1>), grid, dim3(256), 0, st, gw, gd, nxw);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}
