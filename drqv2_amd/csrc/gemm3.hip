// f32 GEMM for the hidden_dim x hidden_dim layers of the policy / Q MLPs (drqv2.py:77-81,103-111), round 3.
//
// gemm.hip stages operands through LDS with register staging and one barrier per k-tile (MFMA busy 37 %), gemm2.hip
// feeds a 32x32 tile per wave straight from global memory (26-30 %): at M = batch = 256 both are bound by how they
// FEED the matrix pipe, not by it.  Here the feed is an LDS-DMA ring:
//   * a workgroup (4 waves, one per SIMD) owns a (32*WM) x (32*WN) output tile; wave = one 32x32 sub-tile on
//     v_mfma_f32_32x32x2_f32 (one accumulator chain: issue interval = dependent latency = 64 cycles), optionally with
//     the k-steps of every k-tile split over WK wave groups (summed in wave order through LDS: deterministic);
//   * k-tiles of 32 are copied global -> LDS by `buffer_load_dwordx4 ... lds` (no registers, no VALU, no LDS-write
//     instructions), NS stages deep: the copy of tile t+NS-1 is issued while tile t is multiplied, one counted
//     `s_waitcnt vmcnt` + one raw `s_barrier` per k-tile; slots past the last k-tile are issued through a descriptor
//     with zero records (dropped by the range check) so that the wait count is the same in every iteration;
//   * a k-contiguous operand (x[m][k], W[n][k]) lies in LDS as [row][32 k] with the 16-byte chunk index XOR-ed with
//     (row >> 1) & 7 -- applied on the SOURCE address of the copy, the LDS side of an LDS-DMA is lane-linear -- so
//     that the `ds_read_b128` operand reads (lane = row, 4 consecutive k) are conflict-free; a row-contiguous
//     operand (W[k][n], dy[k][m], x[k][n]) lies as [32 k][rows] and is read with conflict-free `ds_read_b32`;
//   * lane (i = lane & 31, h = lane >> 5) feeds MFMA 4q+e of a k-tile with k = 8q + 4h + e from both operands
//     (any k order is a valid reduction as long as both operands use the same one);
//   * operand registers are double-buffered: the reads of tile t+1 are issued before the MFMAs of tile t.
// Forms (operand addressing of gemm.hip): 0 forward  y = x W^T (+bias, ReLU, optional partial dots with the
// next layer's single output row: the Q head's Linear(hidden, 1) needs no launch of its own);  1 dgrad
// dx = (dy W) masked;  2 wgrad  dW = dy^T x with the bias gradient (row sums of dy^T) fused.
// The weight gradient and the input gradient of one layer run as block ranges of ONE launch (gemm3_pair_kernel).
#include "common.h"
#include "../../include/drqv2_hip.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int MAXB3 = 8;

struct G3Args {
  const float* A[MAXB3];
  const float* B[MAXB3];
  float* C[MAXB3];
  const float* bias[MAXB3];
  const float* aux[MAXB3];      // ReLU mask source [M][ldaux] or null
  float* rowsum[MAXB3];         // form 2: [M] row sums of A (bias gradient) or null
  const float* qw[MAXB3];       // form 0: weight row [N] of a following Linear(N, 1) or null
  float* qpart[MAXB3];          //         its partial dots [M][N / BN] (column tile j holds sum over the tile's columns)
  long lda, ldb, ldc;
  int ldaux;
  int M, N, K;
  int relu;
  unsigned a_bytes, b_bytes;
  int dbg;                      // development build only: 1 = issue no copies, 2 = no barrier in the k loop
};

__device__ __forceinline__ int rowmap3(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int FORM, int WM, int WN, int WK, int NS>
struct G3Cfg {
  static_assert(WM * WN * WK == 4, "four waves per workgroup");
  static constexpr bool A_KC = FORM < 2, B_KC = FORM == 0;
  static constexpr int BM = 32 * WM, BN = 32 * WN;
  static constexpr int A_FLOATS = BM * 32, B_FLOATS = BN * 32, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int NI_A = BM / 8, NI_B = BN / 8;       // 1-KiB copy instructions per operand tile
  static_assert(NI_A % 4 == 0 && NI_B % 4 == 0, "every wave issues the same number of copies");
  static constexpr int JA = NI_A / 4, JB = NI_B / 4, IPW = JA + JB;
  static constexpr int QN = 4 / WK;                          // groups of 8 k per wave and k-tile
  static constexpr int RED = (WK > 1 ? 4 * 16 * 64 : 0) + 4 * 32 + 4 * 32;   // epilogue scratch (floats)
  static constexpr int LDS_FLOATS = NS * STAGE > RED ? NS * STAGE : RED;
};

// per-lane source byte offset of copy instruction ii (1 KiB of the LDS tile) of an operand tile
template <bool KC, int BR>
__device__ __forceinline__ unsigned dma_voff(int ii, int lane, int row0, unsigned ld4) {
  if constexpr (KC) {
    const int r = 8 * ii + (lane >> 3);                   // 8 rows of 128 bytes per instruction
    const int c = (lane & 7) ^ ((r >> 1) & 7);            // this LDS slot holds chunk c of the row
    return (unsigned)(row0 + r) * ld4 + (unsigned)c * 16u;
  } else {
    constexpr int LPR = BR / 4, RPI = 64 / LPR;           // lanes per k-row, k-rows per instruction
    const int k = RPI * ii + lane / LPR;
    return (unsigned)k * ld4 + (unsigned)(row0 * 4 + (lane % LPR) * 16);
  }
}

// One piece of the wave's operand values of a k-tile, v[4q + e] = element (row sub*32 + i, k = 8(q0+q) + 4h + e):
// k-contiguous operand: piece p = the ds_read_b128 of q = p (QN pieces); row-contiguous: piece p = the ds_read_b32 of
// v[p] (4*QN pieces).  Pieces are issued one or two per MFMA gap (see gemm3_body).
template <bool KC, int QN>
constexpr int n_pieces() { return KC ? QN : 4 * QN; }

template <bool KC, int BR, int QN>
__device__ __forceinline__ void read_piece(float (&v)[4 * QN], const float* tile, int sub, int q0, int i, int h, int p) {
  if constexpr (KC) {
    const int row = sub * 32 + i;
    const int sw = (row >> 1) & 7;
    const int c = 2 * (q0 + p) + h;
    const f32x4 t = *reinterpret_cast<const f32x4*>(tile + (row * 8 + (c ^ sw)) * 4);
    const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    v[4 * p + 0] = t0; v[4 * p + 1] = t1; v[4 * p + 2] = t2; v[4 * p + 3] = t3;
  } else {
    v[p] = tile[(8 * (q0 + (p >> 2)) + 4 * h + (p & 3)) * BR + sub * 32 + i];
  }
}

// one 1-KiB LDS-DMA copy: lane l's 16 bytes at (voff + soff) of the buffer land at lds + 16*l.  (The host pass of hipcc
// checks the builtin's size argument against the HOST target: the call exists in the device pass only.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tile: the workgroup's output tile (m fastest); smem: the workgroup's LDS array (G3Cfg::LDS_FLOATS)
template <int FORM, int WM, int WN, int WK, int NS>
__device__ __forceinline__ void gemm3_body(const G3Args& g, int tile, int batch, float* smem) {
  using Cfg = G3Cfg<FORM, WM, WN, WK, NS>;
  constexpr bool A_KC = Cfg::A_KC, B_KC = Cfg::B_KC;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, QN = Cfg::QN, STAGE = Cfg::STAGE, JA = Cfg::JA, JB = Cfg::JB;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wk = wid / (WM * WN), wr = wid % (WM * WN), wm = wr / WN, wn = wr % WN;
  const int MT = g.M / BM;
  const int mt = tile % MT, ntile = tile / MT;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int nkt = g.K >> 5;

  const unsigned lda4 = (unsigned)g.lda * 4, ldb4 = (unsigned)g.ldb * 4;
  unsigned avoff[JA], bvoff[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) avoff[j] = dma_voff<A_KC, BM>(wid + 4 * j, lane, m0, lda4);
#pragma unroll
  for (int j = 0; j < JB; ++j) bvoff[j] = dma_voff<B_KC, BN>(wid + 4 * j, lane, n0, ldb4);
  const unsigned astep = A_KC ? 128u : 32u * lda4, bstep = B_KC ? 128u : 32u * ldb4;
  const float* Ap = g.A[batch];
  const float* Bp = g.B[batch];

  // one 1-KiB piece (of the wave's JA + JB) of the copy of k-tile s into ring buffer `buf`; s >= nkt: zero records, the
  // copy is dropped by the range check (same instruction count in every iteration: the wait counts stay exact)
  auto dma_piece = [&](int s, int buf, int j) {
#ifdef DRQ_DEV
    if (g.dbg & 1) return;
#endif
    const bool live = s < nkt;
    float* st = smem + buf * STAGE;
    if (j < JA) {
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, live ? g.a_bytes : 0u, 0x00020000);
      dma16(ars, st + (wid + 4 * j) * 256, avoff[j < JA ? j : 0], (unsigned)s * astep);
    } else {
      const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, live ? g.b_bytes : 0u, 0x00020000);
      dma16(brs, st + Cfg::A_FLOATS + (wid + 4 * (j - JA)) * 256, bvoff[j >= JA ? j - JA : 0], (unsigned)s * bstep);
    }
  };

  // two accumulator chains (even / odd k-steps of a tile), added in the epilogue: the next MFMA never waits for the
  // previous one's result
  f32x16 acc, accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
  float rs = 0.f;                                  // form 2: row sum of A (bias gradient), this lane's k values
  const bool do_rs = FORM == 2 && g.rowsum[batch] != nullptr && ntile == 0 && wn == 0;

  constexpr int NM = 4 * QN;                       // MFMAs per k-tile and wave
  constexpr int NA = n_pieces<A_KC, QN>(), NB = n_pieces<B_KC, QN>(), NF = NA + NB + JA + JB;
  float a0[NM], b0[NM], a1[NM], b1[NM];
  const int q0 = wk * QN;
  // filler f of a k-tile: operand read pieces of tile t+1 first (A, then B), then the copy pieces of tile t+NS-1
  auto filler = [&](int f, float (&na)[NM], float (&nb)[NM], int bnext, int s_fill, int bfill) {
    const float* st = smem + bnext * STAGE;
    if (f < NA) read_piece<A_KC, BM, QN>(na, st, wm, q0, i, h, f);
    else if (f < NA + NB) read_piece<B_KC, BN, QN>(nb, st + Cfg::A_FLOATS, wn, q0, i, h, f - NA);
    else dma_piece(s_fill, bfill, f - NA - NB);
  };

  // ---- prologue: NS-1 tiles in flight, the first one into registers
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
#pragma unroll
    for (int j = 0; j < JA + JB; ++j) dma_piece(s, s, j);
  wait_vmcnt<Cfg::IPW*(NS - 2)>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < NA + NB; ++f) filler(f, a0, b0, 0, 0, 0);

  // Iteration t: tile t+1 has landed (counted wait + barrier); then the MFMAs of tile t with, in the 64-cycle shadow of
  // each, one or two fillers: the operand reads of tile t+1 and the copy of tile t+NS-1 into the buffer tile t-1 was read
  // from (its readers passed the barrier above).  A wave issues in order: fillers in a block in front of the MFMAs
  // (an LDS-DMA piece takes 60-180 cycles to issue) would leave the matrix pipe idle for a third of the iteration.
  int bnext = 1, bfill = NS - 1;                   // ring indices of tile t+1 and tile t+NS-1
  auto iter = [&](int t, const float (&a)[NM], const float (&b)[NM], float (&na)[NM], float (&nb)[NM]) {
    wait_vmcnt<Cfg::IPW*(NS - 3)>();
#ifdef DRQ_DEV
    if (!(g.dbg & 2))
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NM; ++e) {
      if (e & 1) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], accb, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
#pragma unroll
      for (int f = (e * NF + NM - 1) / NM; f < ((e + 1) * NF + NM - 1) / NM; ++f) filler(f, na, nb, bnext, t + NS - 1, bfill);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FORM == 2) {
      if (do_rs) {
#pragma unroll
        for (int e = 0; e < NM; ++e) rs += a[e];
      }
    }
    bnext = bnext + 1 == NS ? 0 : bnext + 1;
    bfill = bfill + 1 == NS ? 0 : bfill + 1;
  };
  auto mfmas = [&](const float (&a)[NM], const float (&b)[NM]) {
#pragma unroll
    for (int e = 0; e < NM; ++e) {
      if (e & 1) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], accb, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
    if constexpr (FORM == 2) {
      if (do_rs) {
#pragma unroll
        for (int e = 0; e < NM; ++e) rs += a[e];
      }
    }
  };
  int t = 0;
  for (; t + 2 <= nkt; t += 2) {
    iter(t, a0, b0, a1, b1);
    iter(t + 1, a1, b1, a0, b0);
  }
  if (t < nkt) mfmas(a0, b0);
  // every outstanding copy is a dropped one; drain them before the LDS is reused below
  wait_vmcnt<0>();
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += accb[r];

  if constexpr (FORM == 2) {
    if (do_rs) rs += __shfl_xor(rs, 32);
  }

  // ---- K split over wave groups: sum in wave-group order through LDS
  if constexpr (WK > 1) {
    __builtin_amdgcn_s_barrier();                  // the last tile's operand reads are done in every wave
    float* red = smem;
    float* rsred = smem + 4 * 16 * 64;
    if (wk > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wid - WM * WN) * 16 + r) * 64 + lane] = acc[r];
      if (h == 0) rsred[(wid - WM * WN) * 32 + i] = rs;
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
#pragma unroll
        for (int w = 0; w < WK - 1; ++w) v += red[((w * WM * WN + wr) * 16 + r) * 64 + lane];
        acc[r] = v;
      }
#pragma unroll
      for (int w = 0; w < WK - 1; ++w) rs += rsred[(w * WM * WN + wr) * 32 + i];
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout column = lane & 31 (n), rows (r&3) + 8*(r>>2) + 4*half (m)
  const int n = n0 + wn * 32 + i;
  const int mrow = m0 + wm * 32;
  const bool writer = wk == 0;
  float v[16];
  if (writer) {
    const float bv = g.bias[batch] ? g.bias[batch][n] : 0.f;
    float mk[16];
    const float* ap = g.aux[batch];
#pragma unroll
    for (int r = 0; r < 16; ++r) mk[r] = ap ? ap[(long)(mrow + rowmap3(r, h)) * g.ldaux + n] : 1.f;
    float* c = g.C[batch];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = acc[r] + bv;
      if (g.relu) x = x > 0.f ? x : 0.f;
      x = mk[r] > 0.f ? x : 0.f;
      v[r] = x;
      c[(long)(mrow + rowmap3(r, h)) * g.ldc + n] = x;
    }
    if constexpr (FORM == 2) {
      if (do_rs && h == 0) g.rowsum[batch][mrow + i] = rs;
    }
  }
  if constexpr (FORM == 0) {
    // partial dots with the next layer's single weight row: sum over the workgroup's BN columns, per output row
    if (g.qw[batch]) {                             // uniform over the workgroup
      float* qred = smem + (WK > 1 ? 4 * 16 * 64 + 4 * 32 : 0);
      if constexpr (WK == 1) __builtin_amdgcn_s_barrier();      // operand reads of the last tile are done
      if (writer) {
        const float wv = g.qw[batch][n];
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = v[r] * wv;
        // sixteen independent butterflies, step by step: the cross-lane moves of a step are in flight together
#pragma unroll
        for (int o = 1; o < 32; o <<= 1)
#pragma unroll
          for (int r = 0; r < 16; ++r) p[r] += __shfl_xor(p[r], o);
        if (i == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) qred[(wm * WN + wn) * 32 + rowmap3(r, h)] = p[r];
        }
      }
      __syncthreads();
      const int tid = threadIdx.x;
      if (tid < BM) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) s += qred[((tid >> 5) * WN + w) * 32 + (tid & 31)];
        g.qpart[batch][(long)(m0 + tid) * (g.N / BN) + ntile] = s;
      }
    }
  }
}

// Linear workgroup id -> (problem, tile).  The workgroups of one XCD (ids equal mod 8 under the dispatcher's round-robin:
// a speed assumption only, never correctness) get a CONTIGUOUS range of the (problem, tile) space, m fastest: at four
// problems of 64 tiles, an XCD works on half the column tiles of ONE problem -- 2 MB of weight rows and that problem's
// 1 MB of activations fit its 4 MB L2, instead of a slice of every problem (6 MB) that does not.
__device__ __forceinline__ void xcd_decode(int id, int ntiles, int total, int& batch, int& tile) {
  const int l = (total & 7) ? id : (id & 7) * (total >> 3) + (id >> 3);
  batch = l / ntiles;
  tile = l - batch * ntiles;
}

template <int FORM, int WM, int WN, int WK, int NS>
__global__ __launch_bounds__(256) void gemm3_kernel(G3Args g) {
  __shared__ __attribute__((aligned(16))) float smem[G3Cfg<FORM, WM, WN, WK, NS>::LDS_FLOATS];
  int batch, tile;
  xcd_decode((int)(blockIdx.x + gridDim.x * blockIdx.z), (int)gridDim.x, (int)(gridDim.x * gridDim.z), batch, tile);
  gemm3_body<FORM, WM, WN, WK, NS>(g, tile, batch, smem);
}

// weight gradient (form 2) in workgroups [0, nxw) of every problem, input gradient (form 1) in the rest: they read the
// same dy and do not depend on each other
template <int W_WM, int W_WN, int W_WK, int W_NS, int D_WM, int D_WN, int D_WK, int D_NS>
__global__ __launch_bounds__(256) void gemm3_pair_kernel(G3Args gw, G3Args gd, int nxw) {
  constexpr int LW = G3Cfg<2, W_WM, W_WN, W_WK, W_NS>::LDS_FLOATS, LD = G3Cfg<1, D_WM, D_WN, D_WK, D_NS>::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[LW > LD ? LW : LD];
  const int bx = (int)blockIdx.x, nz = (int)gridDim.z, z = (int)blockIdx.z;
  int batch, tile;
  if (bx < nxw) {
    xcd_decode(bx + nxw * z, nxw, nxw * nz, batch, tile);
    gemm3_body<2, W_WM, W_WN, W_WK, W_NS>(gw, tile, batch, smem);
  } else {
    const int nxd = (int)gridDim.x - nxw;
    xcd_decode(bx - nxw + nxd * z, nxd, nxd * nz, batch, tile);
    gemm3_body<1, D_WM, D_WN, D_WK, D_NS>(gd, tile, batch, smem);
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// shape / alignment checks shared by the entries; fills the argument block
int fill(G3Args& g, int form, int nbatch, const float* const* A, long lda, const float* const* B, long ldb,
         float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu, const float* const* aux,
         int ldaux, float* const* rowsum) {
  if (nbatch <= 0 || nbatch > MAXB3 || M % 64 || N % 32 || K % 32 || M < 64 || N < 32 || K < 64) return DRQ_EARG;
  if (lda % 4 || ldb % 4) return DRQ_EARG;
  const bool a_kc = form < 2, b_kc = form == 0;
  const size_t ab = (a_kc ? (size_t)M * lda : (size_t)K * lda) * 4, bb = (b_kc ? (size_t)N * ldb : (size_t)K * ldb) * 4;
  if (ab >= (1ull << 31) || bb >= (1ull << 31)) return DRQ_EARG;
  for (int b = 0; b < nbatch; ++b) {
    if (!A[b] || !B[b] || !C[b] || !al16(A[b]) || !al16(B[b])) return DRQ_EARG;
    g.A[b] = A[b]; g.B[b] = B[b]; g.C[b] = C[b];
    g.bias[b] = bias ? bias[b] : nullptr;
    g.aux[b] = aux ? aux[b] : nullptr;
    g.rowsum[b] = rowsum ? rowsum[b] : nullptr;
  }
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux; g.M = M; g.N = N; g.K = K; g.relu = relu;
  g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  return DRQ_OK;
}

}  // namespace

// internal (gemm.hip, step.hip).  Returns DRQ_EARG when the problem is not eligible (the caller then uses another kernel).
// Forward form with optional partial dots: qw[b] = weight row [N] of a following Linear(N,1), qpart[b] = [M][*nq_out]
// partial sums (the consumer adds them in index order, plus that layer's bias).
int drq_gemm3_fwd(int nbatch, const float* const* A, long lda, const float* const* B, long ldb, float* const* C,
                  long ldc, int M, int N, int K, const float* const* bias, int relu, const float* const* qw,
                  float* const* qpart, int* nq_out, hipStream_t st) {
  G3Args g{};
  const int rc = fill(g, 0, nbatch, A, lda, B, ldb, C, ldc, M, N, K, bias, relu, nullptr, 0, nullptr);
  if (rc != DRQ_OK) return rc;
  if (N % 64) return DRQ_EARG;
  for (int b = 0; b < nbatch; ++b) {
    g.qw[b] = qw ? qw[b] : nullptr;
    g.qpart[b] = qpart ? qpart[b] : nullptr;
    if ((g.qw[b] != nullptr) != (g.qpart[b] != nullptr)) return DRQ_EARG;
  }
  const long t64 = (long)(M / 64) * (N / 64) * nbatch;
#ifdef DRQ_DEV
  static const char* const dbg = getenv("DRQ_G3_DBG");          // development build only (tools/gemm3_bench.py)
  const int d = dbg ? atoi(dbg) : 0;
  if (d & 1) g.a_bytes = g.b_bytes = 0;                         // every copy dropped: MFMA + LDS + barrier time only
  g.dbg = d >> 4;
  if (d & 2) {                                                  // deeper ring
    if (nq_out) *nq_out = N / 64;
    hipLaunchKernelGGL((gemm3_kernel<0, 2, 2, 1, 9>), dim3((M / 64) * (N / 64), 1, nbatch), dim3(256), 0, st, g);
    DRQ_LAUNCH_CHECK();
    return DRQ_OK;
  }
  if (d & 4) {                                                  // shallow ring
    if (nq_out) *nq_out = N / 64;
    hipLaunchKernelGGL((gemm3_kernel<0, 2, 2, 1, 4>), dim3((M / 64) * (N / 64), 1, nbatch), dim3(256), 0, st, g);
    DRQ_LAUNCH_CHECK();
    return DRQ_OK;
  }
#endif
  // enough 64x64 tiles to give every CU one: one sub-tile per wave; else 64x32 tiles with the k-steps of a k-tile
  // split over two wave groups (twice the workgroups, half the MFMAs per wave)
  if (t64 >= drq_num_cus()) {
    if (nq_out) *nq_out = N / 64;
    hipLaunchKernelGGL((gemm3_kernel<0, 2, 2, 1, 6>), dim3((M / 64) * (N / 64), 1, nbatch), dim3(256), 0, st, g);
  } else {
    if (nq_out) *nq_out = N / 32;
    hipLaunchKernelGGL((gemm3_kernel<0, 2, 1, 2, 6>), dim3((M / 64) * (N / 32), 1, nbatch), dim3(256), 0, st, g);
  }
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// dgrad: dx [M][N] = (dy [M][K] W [K][ldb]) * (mask > 0)
int drq_gemm3_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* w, long ldw, float* const* dx,
                    long lddx, int M, int N, int K, const float* const* mask, int ldmask, hipStream_t st) {
  G3Args g{};
  const int rc = fill(g, 1, nbatch, dy, lddy, w, ldw, dx, lddx, M, N, K, nullptr, 0, mask, ldmask, nullptr);
  if (rc != DRQ_OK) return rc;
  if (N % 64) return DRQ_EARG;
  const long t64 = (long)(M / 64) * (N / 64) * nbatch;
  if (t64 >= drq_num_cus())
    hipLaunchKernelGGL((gemm3_kernel<1, 2, 2, 1, 6>), dim3((M / 64) * (N / 64), 1, nbatch), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm3_kernel<1, 2, 1, 2, 6>), dim3((M / 64) * (N / 32), 1, nbatch), dim3(256), 0, st, g);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// wgrad + dgrad of one hidden layer in one launch.  dy [Brows][Nout] (ld lddy), x [Brows][Kin] (ld ldx), w [Nout][ldw]:
// dW [Nout][Kin] = dy^T x, db [Nout] = column sums of dy, dx [Brows][Kin] = (dy w) * (mask > 0)
int drq_gemm3_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                          float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                          long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin, hipStream_t st) {
  G3Args gw{}, gd{};
  // wgrad: A(m = n_out, k = row) = dy[k*lddy + m], B(k, n = k_in) = x[k*ldx + n]
  int rc = fill(gw, 2, nbatch, dy, lddy, x, ldx, dw, Kin, Nout, Kin, Brows, nullptr, 0, nullptr, 0, db);
  if (rc != DRQ_OK) return rc;
  // dgrad: A(m = row, k = n_out) = dy[m*lddy + k], B(k, n = k_in) = w[k*ldw + n]
  rc = fill(gd, 1, nbatch, dy, lddy, w, ldw, dx, lddx, Brows, Kin, Nout, nullptr, 0, mask, ldmask, nullptr);
  if (rc != DRQ_OK) return rc;
  if (Kin % 64 || Nout % 64 || Brows % 64) return DRQ_EARG;
  const int nxw = (Nout / 64) * (Kin / 64);
  const long td64 = (long)(Brows / 64) * (Kin / 64) * nbatch;
  if (td64 >= drq_num_cus()) {
    const int nxd = (Brows / 64) * (Kin / 64);
    hipLaunchKernelGGL((gemm3_pair_kernel<2, 2, 1, 4, 2, 2, 1, 6>), dim3(nxw + nxd, 1, nbatch), dim3(256), 0, st, gw, gd,
                       nxw);
  } else {
    const int nxd = (Brows / 64) * (Kin / 32);
    hipLaunchKernelGGL((gemm3_pair_kernel<2, 2, 1, 4, 2, 1, 2, 6>), dim3(nxw + nxd, 1, nbatch), dim3(256), 0, st, gw, gd,
                       nxw);
  }
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

extern "C" {

// See include/drqv2_hip.h
DRQ_API int drq_mlp_fwd(int nbatch, const float* const* x, long ldx, const float* const* w, long ldw, float* const* y,
                        long ldy, int M, int N, int K, const float* const* bias, int relu, const float* const* qw,
                        float* const* qpart, int* nq_out, drq_stream_t stream) {
  if (!x || !w || !y) return DRQ_EARG;
  return drq_gemm3_fwd(nbatch, x, ldx, w, ldw, y, ldy, M, N, K, bias, relu, qw, qpart, nq_out, (hipStream_t)stream);
}

DRQ_API int drq_mlp_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* w, long ldw,
                          float* const* dx, long lddx, int M, int N, int K, const float* const* mask, int ldmask,
                          drq_stream_t stream) {
  if (!dy || !w || !dx) return DRQ_EARG;
  return drq_gemm3_dgrad(nbatch, dy, lddy, w, ldw, dx, lddx, M, N, K, mask, ldmask, (hipStream_t)stream);
}

DRQ_API int drq_mlp_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                                float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                                long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin,
                                drq_stream_t stream) {
  if (!dy || !x || !dw || !w || !dx) return DRQ_EARG;
  return drq_gemm3_wgrad_dgrad(nbatch, dy, lddy, x, ldx, dw, db, w, ldw, dx, lddx, mask, ldmask, Brows, Nout, Kin,
                               (hipStream_t)stream);
}

}  // extern "C"
