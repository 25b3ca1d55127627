// f32 GEMM for the fully connected layers of DrQ-v2 (trunk, policy, Q heads; drqv2.py:74-81,100-111)
// on v_mfma_f32_16x16x4_f32 (exact f32 fma chain, k-ordered).
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(k,n) )
//
// Operand addressing covers the three shapes autograd needs without materialising transposes:
//   A_KC: A(m,k) = A[m*lda + k]   else   A(m,k) = A[k*lda + m]
//   B_KC: B(k,n) = B[n*ldb + k]   else   B(k,n) = B[k*ldb + n]
//   forward  y = x W^T : A_KC, B_KC      dgrad  dx = dy W : A_KC, !B_KC      wgrad  dW = dy^T x : !A_KC, !B_KC
// Workgroup = 4 waves (2x2); wave tile = TMxTN MFMA tiles of 16x16; block tile (32*TM)x(32*TN), BK = 32.
// Global -> registers (prefetched one k-tile ahead) -> LDS [row][k] with pitch 34 (bank = 2*row + k:
// conflict-free MFMA operand reads) -> MFMA.  Split-K writes dense partials; a second kernel sums them
// in a fixed order (deterministic) and applies the epilogue.
#include "common.h"
#include <type_traits>

namespace {

constexpr int BK0 = 32;   // k-tile of the default configurations (split-K chunks are multiples of 64: both fit)

constexpr int MAXB = 8;      // problems per launch (same shape, independent pointers)

struct Epilogue {
  const float* bias[MAXB];   // [N] or null
  const float* aux[MAXB];    // relu mask source [M][ldaux] or null : v = aux>0 ? v : 0
  float* rowsum[MAXB];       // [M] or null: sum_k A(m,k) (bias gradient of a wgrad GEMM); needs !A_KC, splitk == 1
  int ldaux;
  int relu;
  int scatter_hw;            // >0: C index = padded NCHW scatter (trunk dgrad -> conv4 grad layout, pad 2)
};

struct GemmArgs {
  const float* A[MAXB];
  const float* B[MAXB];
  float* C[MAXB];
  long lda, ldb, ldc;
  int M, N, K;
  int nbatch, splitk, kchunk;
  float* part;               // split-K partials [nbatch*splitk][M][N]
  Epilogue ep;
};

__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int batch, int m, int n, float v) {
  const Epilogue& e = g.ep;
  if (e.bias[batch]) v += e.bias[batch][n];
  if (e.relu) v = v > 0.f ? v : 0.f;
  if (e.aux[batch]) v = (e.aux[batch][(long)m * e.ldaux + n] > 0.f) ? v : 0.f;
  float* c = g.C[batch];
  if (e.scatter_hw > 0) {
    const int hw = e.scatter_hw, hp = hw + 4;
    const int ch = n / (hw * hw);
    const int r = n - ch * hw * hw;
    const int y = r / hw, x = r - y * hw;
    c[(((long)m * 32 + ch) * hp + (y + 2)) * hp + (x + 2)] = v;
  } else {
    c[(long)m * g.ldc + n] = v;
  }
}

// ---- global -> register tile loaders (BR rows of the block tile, BK columns of k).
// Loads are unconditional on clamped (always valid) addresses and carry NO dependent ALU: the out-of-range
// mask is applied when the tile is written to LDS (store_tile).  A load behind a per-lane branch, or a
// select right after it, makes the compiler wait for it at once and serialises the register queue.
// Requires rows >= 1 and kend > kbeg (checked by the caller).
// FULL: every tile of the launch lies inside the matrix (no clamps here, no selects in store_tile: a select is a
// VALU instruction per staged element, and VALU work between MFMAs costs matrix-pipe cycles -- DESIGN.md section 3)
template <int BR, bool KC, bool V4, int BK, bool FULL = false>
__device__ __forceinline__ void load_tile(const float* P, long ld, int row0, int rows, int k0, int kend,
                                          float (&reg)[BR * BK / 256], int tid) {
  if constexpr (KC && V4) {
    // BK/4 threads x float4 along k per row, 1024/BK rows per pass (kend % 4 == 0, kend >= 4)
    constexpr int TPR = BK / 4, RPP = 256 / TPR;
    const int r = tid / TPR, kq = (tid % TPR) * 4;
    const int kc = FULL ? k0 + kq : min(k0 + kq, kend - 4);
#pragma unroll
    for (int p = 0; p < BR / RPP; ++p) {
      const int row = FULL ? row0 + p * RPP + r : min(row0 + p * RPP + r, rows - 1);
      f32x4 v;
      if constexpr (FULL) {
        // uniform base (advances with k0: scalar) + loop-invariant 32-bit lane offset: no address VALU per tile
        // (a buffer load: with a plain pointer hipcc adds k0 to a per-lane 64-bit address on the VALU every tile)
        const unsigned voff = (unsigned)(((long)row * ld + kq) * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, 0xffffffffu, 0x00020000);
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)k0 * 4u, 0));
      } else {
        v = *reinterpret_cast<const f32x4*>(P + (long)row * ld + kc);
      }
      reg[p * 4 + 0] = v[0]; reg[p * 4 + 1] = v[1]; reg[p * 4 + 2] = v[2]; reg[p * 4 + 3] = v[3];
    }
  } else if constexpr (KC) {
    // BK lanes along k per row, 256/BK rows per pass
    constexpr int RPP = 256 / BK;
    const int r = tid / BK, kk = tid % BK;
    const int kc = min(k0 + kk, kend - 1);
#pragma unroll
    for (int p = 0; p < BR / RPP; ++p) reg[p] = P[(long)min(row0 + p * RPP + r, rows - 1) * ld + kc];
  } else {
    // row index contiguous in memory: lanes along rows
    const int r = tid % BR, kk0 = tid / BR;
    constexpr int KSTEP = 256 / BR;
    const int rc = min(row0 + r, rows - 1);
#pragma unroll
    for (int p = 0; p < BR * BK / 256; ++p) reg[p] = P[(long)min(k0 + kk0 + p * KSTEP, kend - 1) * ld + rc];
  }
}

// registers -> LDS [row][k] (pitch PK), zeroing what lies outside the matrix
template <int BR, bool KC, bool V4, int BK, bool FULL = false>
__device__ __forceinline__ void store_tile(float* S, const float (&reg)[BR * BK / 256], int row0, int rows, int k0,
                                           int kend, int tid) {
  constexpr int PK = BK + 2;
  if constexpr (KC && V4) {
    constexpr int TPR = BK / 4, RPP = 256 / TPR;
    const int r = tid / TPR, kq = (tid % TPR) * 4;
    const bool kok = FULL || k0 + kq < kend;
#pragma unroll
    for (int p = 0; p < BR / RPP; ++p) {
      const bool ok = FULL || (kok && row0 + p * RPP + r < rows);
      float* d = S + (p * RPP + r) * PK + kq;
      d[0] = ok ? reg[p * 4 + 0] : 0.f; d[1] = ok ? reg[p * 4 + 1] : 0.f;
      d[2] = ok ? reg[p * 4 + 2] : 0.f; d[3] = ok ? reg[p * 4 + 3] : 0.f;
    }
  } else if constexpr (KC) {
    constexpr int RPP = 256 / BK;
    const int r = tid / BK, kk = tid % BK;
    const bool kok = k0 + kk < kend;
#pragma unroll
    for (int p = 0; p < BR / RPP; ++p) S[(p * RPP + r) * PK + kk] = (kok && row0 + p * RPP + r < rows) ? reg[p] : 0.f;
  } else {
    const int r = tid % BR, kk0 = tid / BR;
    constexpr int KSTEP = 256 / BR;
    const bool rok = row0 + r < rows;
#pragma unroll
    for (int p = 0; p < BR * BK / 256; ++p)
      S[r * PK + kk0 + p * KSTEP] = (rok && k0 + kk0 + p * KSTEP < kend) ? reg[p] : 0.f;
  }
}

// masked sum of this thread's staged A values (row-contiguous layout only): piece of sum_k A(m,k)
template <int BR, int BK>
__device__ __forceinline__ float tile_rowsum(const float (&reg)[BR * BK / 256], int row0, int rows, int k0, int kend,
                                             int tid) {
  const int r = tid % BR, kk0 = tid / BR;
  constexpr int KSTEP = 256 / BR;
  float s = 0.f;
  if (row0 + r < rows) {
#pragma unroll
    for (int p = 0; p < BR * BK / 256; ++p)
      if (k0 + kk0 + p * KSTEP < kend) s += reg[p];
  }
  return s;
}

template <int TM, int TN, bool A_KC, bool B_KC, bool V4, int BK = BK0, bool FULL = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int PK = BK + 2;      // LDS pitch: bank = 2*row + k, conflict-free MFMA operand reads
  constexpr int BM = 32 * TM, BN = 32 * TN;
  constexpr int RA = BM * BK / 256, RB = BN * BK / 256;   // staging registers per thread and k-tile
  // Depth of the register queue of k-tiles in flight from global memory.  Small block tiles do little MFMA
  // work per k-tile (8 MFMAs/wave), so they keep more tiles in flight to cover an L2/HBM round trip.
  constexpr int D = (TM * TN == 1 && BK == 32) ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float As[2][BM * PK];   // double-buffered: one barrier per k-tile
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * PK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int z = blockIdx.z;
  const int batch = z / g.splitk, ks = z - batch * g.splitk;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const float* A = g.A[batch];
  const float* B = g.B[batch];

  float ra[D][RA], rb[D][RB];

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int arow = (wm * TM * 16 + (lane & 15)) * PK + (lane >> 4);
  const int brow = (wn * TN * 16 + (lane & 15)) * PK + (lane >> 4);
  const int nsteps = (kend - kbeg + BK - 1) / BK;

  // tile t lives in register slot t % D until it is written to LDS buffer t & 1 during step t-1
  auto fetch = [&](int t, float (&fa)[RA], float (&fb)[RB]) {
    const int tc = t < nsteps ? t : nsteps - 1;   // past the end: harmless re-load, never written to LDS
    load_tile<BM, A_KC, V4, BK, FULL>(A, g.lda, m0, g.M, kbeg + tc * BK, kend, fa, tid);
    load_tile<BN, B_KC, V4, BK, FULL>(B, g.ldb, n0, g.N, kbeg + tc * BK, kend, fb, tid);
  };
  // fused bias gradient of a wgrad GEMM: the workgroups of the first column tile also sum their A rows over k
  bool do_rs = false;
  if constexpr (!A_KC) do_rs = g.ep.rowsum[batch] != nullptr && blockIdx.x == 0;
  float rs = 0.f;
  if (nsteps > 0) {
    fetch(0, ra[0], rb[0]);
    if constexpr (!A_KC)
      if (do_rs) rs += tile_rowsum<BM, BK>(ra[0], m0, g.M, kbeg, kend, tid);
    store_tile<BM, A_KC, V4, BK, FULL>(As[0], ra[0], m0, g.M, kbeg, kend, tid);
    store_tile<BN, B_KC, V4, BK, FULL>(Bs[0], rb[0], n0, g.N, kbeg, kend, tid);
#pragma unroll
    for (int d = 1; d <= D; ++d) fetch(d, ra[d % D], rb[d % D]);
  }
  __syncthreads();
  // one k-tile: MFMAs from LDS buffer (s&1), then move tile t+1 from its register slot to the other
  // buffer (last read in step t-1; every wave has passed that barrier) and refill the slot with tile t+1+D.
  // No conditionals inside: a tile past the end is a harmless store into a buffer nobody reads again.
  auto step = [&](int t, auto sc) {
    constexpr int s = decltype(sc)::value;
    const float* as = As[s & 1];      // D is even: t & 1 == s & 1
    const float* bs = Bs[s & 1];
    // all operand reads of the k-tile first (one LDS latency per tile, not one per MFMA pair)
    float a[BK / 4][TM], b[BK / 4][TN];
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[kk][i] = as[arow + i * 16 * PK + kk * 4];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[kk][j] = bs[brow + j * 16 * PK + kk * 4];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    if constexpr (!A_KC)
      if (do_rs) rs += tile_rowsum<BM, BK>(ra[(s + 1) % D], m0, g.M, kbeg + (t + 1) * BK, kend, tid);
    store_tile<BM, A_KC, V4, BK, FULL>(As[(s + 1) & 1], ra[(s + 1) % D], m0, g.M, kbeg + (t + 1) * BK, kend, tid);
    store_tile<BN, B_KC, V4, BK, FULL>(Bs[(s + 1) & 1], rb[(s + 1) % D], n0, g.N, kbeg + (t + 1) * BK, kend, tid);
    fetch(t + 1 + D, ra[(s + 1) % D], rb[(s + 1) % D]);
    __syncthreads();
  };
  int t0 = 0;
  for (; t0 + D <= nsteps; t0 += D) {       // full groups: straight-line body, exact vmcnt bookkeeping
    step(t0 + 0, std::integral_constant<int, 0>{});
    step(t0 + 1, std::integral_constant<int, 1>{});
    if constexpr (D == 4) {
      step(t0 + 2, std::integral_constant<int, 2>{});
      step(t0 + 3, std::integral_constant<int, 3>{});
    }
  }
  if (t0 + 0 < nsteps) step(t0 + 0, std::integral_constant<int, 0>{});
  if constexpr (D == 4) {
    if (t0 + 1 < nsteps) step(t0 + 1, std::integral_constant<int, 1>{});
    if (t0 + 2 < nsteps) step(t0 + 2, std::integral_constant<int, 2>{});
  }

  if constexpr (!A_KC) {
    if (do_rs) {           // 256/BM threads hold pieces of each row sum: combine in a fixed order through LDS
      float* red = As[0];
      red[tid] = rs;
      __syncthreads();
      if (tid < BM && m0 + tid < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 256 / BM; ++q) t += red[tid + q * BM];
        g.ep.rowsum[batch][m0 + tid] = t;
      }
    }
  }

  // ---- store: C/D layout col = lane&15, row = (lane>>4)*4 + reg.  Two phases so that every bias / mask
  // load of the thread is in flight before the first dependent store (no per-element round trips).
  const int mrow0 = m0 + wm * TM * 16 + (lane >> 4) * 4;
  const int ncol0 = n0 + wn * TN * 16 + (lane & 15);
  if (g.splitk > 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mrow0 + i * 16 + r, n = ncol0 + j * 16;
          if (m < g.M && n < g.N) g.part[((long)z * g.M + m) * g.N + n] = acc[i][j][r];
        }
    return;
  }
  const Epilogue& e = g.ep;
  // loads use clamped (always valid) coordinates so that none of them sits behind a per-lane branch
  float bv[TN];
  float av[TM][TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = 0.f;
  if (e.bias[batch]) {
    const float* bp = e.bias[batch];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = bp[min(ncol0 + j * 16, g.N - 1)];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) av[i][j][r] = 1.f;
  if (e.aux[batch]) {
    const float* ap = e.aux[batch];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          av[i][j][r] = ap[(long)min(mrow0 + i * 16 + r, g.M - 1) * e.ldaux + min(ncol0 + j * 16, g.N - 1)];
  }
  float* c = g.C[batch];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = ncol0 + j * 16;
      long cbase = n;
      if (e.scatter_hw > 0) {   // zero-padded (pad 2) NCHW gradient layout, column n = (ch, y, x)
        const int hw = e.scatter_hw, hp = hw + 4;
        const int ch = n / (hw * hw);
        const int rr = n - ch * hw * hw;
        const int y = rr / hw, x = rr - y * hw;
        cbase = ((long)ch * hp + (y + 2)) * hp + (x + 2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mrow0 + i * 16 + r;
        if (m < g.M && n < g.N) {
          float v = acc[i][j][r] + bv[j];
          if (e.relu) v = v > 0.f ? v : 0.f;
          v = av[i][j][r] > 0.f ? v : 0.f;
          if (e.scatter_hw > 0) c[(long)m * 32 * (e.scatter_hw + 4) * (e.scatter_hw + 4) + cbase] = v;
          else c[(long)m * g.ldc + cbase] = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-MFMA variant (BASELINE configs[4], "bf16": the MFMA fc path).  Same GemmArgs, operand layouts, split-K
// partial format and epilogue as gemm_kernel; operands are read as fp32 from global memory, rounded to bf16 (nearest
// even) on the way into LDS ([row][k], 72-element pitch: aligned, conflict-free 16-byte fragment reads), products are
// exact, accumulation is fp32 on v_mfma_f32_32x32x16_bf16.  64x64 block tile (2x2 waves of 32x32), k-tile 64.
// The fused bias gradient (rowsum) sums the UNROUNDED fp32 values.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16_pair(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int BF_BM = 64, BF_BK = 64, BF_PD = 36;   // rows per block tile, k per tile, dword pitch of an LDS row

// Staging is split into a LOAD (global fp32 -> 16 registers, issued one k-tile ahead so that it flies under the
// MFMAs of the current tile) and a STORE (round to bf16, write LDS).
// k-contiguous operand: thread -> (row = tid/4, 16 consecutive k); out-of-range elements are zero
template <bool V4>
__device__ __forceinline__ void bf_load_kc(const float* P, long ld, int row0, int rows, int k0, int kend,
                                           float (&v)[16], int tid) {
  const int r = tid >> 2, kq = (tid & 3) * 16;
  const int row = row0 + r;
  if (V4 && row < rows && k0 + kq + 16 <= kend) {
    const f32x4* p = reinterpret_cast<const f32x4*>(P + (long)row * ld + k0 + kq);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 t = p[i];
      v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
    }
  } else {
    const int rc = row < rows ? row : rows - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = k0 + kq + i;
      const float t = P[(long)rc * ld + (k < kend ? k : kend - 1)];
      v[i] = (row < rows && k < kend) ? t : 0.f;
    }
  }
}
__device__ __forceinline__ void bf_store_kc(const float (&v)[16], unsigned* S, int tid) {
  const int r = tid >> 2, kq = (tid & 3) * 16;
  unsigned* d = S + r * BF_PD + (kq >> 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4_t pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = pack_bf16_pair(v[h * 8 + 2 * e], v[h * 8 + 2 * e + 1]);
    *reinterpret_cast<u32x4_t*>(d + h * 4) = pk;
  }
}

// row-contiguous operand (element (row, k) at P[k*ld + row]): thread -> (4 consecutive rows, 2 x 2 consecutive k):
// v[p*8 + i] = (row i, k), v[p*8 + 4 + i] = (row i, k+1) of pass p
template <bool V4>
__device__ __forceinline__ void bf_load_rc(const float* P, long ld, int row0, int rows, int k0, int kend,
                                           float (&v)[16], int tid) {
  const int rc = (tid & 15) * 4, kp = tid >> 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int k = k0 + 2 * (kp + 16 * p);
    if (V4 && row0 + rc + 4 <= rows && k + 1 < kend) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(P + (long)k * ld + row0 + rc);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(P + (long)(k + 1) * ld + row0 + rc);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[p * 8 + i] = t0[i]; v[p * 8 + 4 + i] = t1[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + rc + i;
        const int rr = row < rows ? row : rows - 1;
        const float t0 = P[(long)(k < kend ? k : kend - 1) * ld + rr];
        const float t1 = P[(long)(k + 1 < kend ? k + 1 : kend - 1) * ld + rr];
        v[p * 8 + i] = (row < rows && k < kend) ? t0 : 0.f;
        v[p * 8 + 4 + i] = (row < rows && k + 1 < kend) ? t1 : 0.f;
      }
    }
  }
}
// rs (optional): += the thread's unrounded values per row (bias gradient of a wgrad GEMM)
__device__ __forceinline__ void bf_store_rc(const float (&v)[16], unsigned* S, int tid, float* rs) {
  const int rc = (tid & 15) * 4, kp = tid >> 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int kk = 2 * (kp + 16 * p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      S[(rc + i) * BF_PD + (kk >> 1)] = pack_bf16_pair(v[p * 8 + i], v[p * 8 + 4 + i]);
      if (rs) rs[i] += v[p * 8 + i] + v[p * 8 + 4 + i];
    }
  }
}

// TN: 32-column tiles per wave (block tile 64 x 64*TN).  TN = 2 serves 64 < N <= 128 (the trunk layer at
// feature_dim 100): the long A operand is then read once instead of once per column tile.
template <bool A_KC, bool B_KC, bool V4, int TN = 1>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned As[BF_BM * BF_PD];
  __shared__ __attribute__((aligned(16))) unsigned Bs[TN * BF_BM * BF_PD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int col = lane & 31, half = lane >> 5;
  const int z = blockIdx.z;
  const int batch = z / g.splitk, ks = z - batch * g.splitk;
  const int m0 = blockIdx.y * BF_BM, n0 = blockIdx.x * (BF_BM * TN);
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const float* A = g.A[batch];
  const float* B = g.B[batch];
  bool do_rs = false;
  if constexpr (!A_KC) do_rs = g.ep.rowsum[batch] != nullptr && blockIdx.x == 0;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned* ap = As + (wm * 32 + col) * BF_PD + half * 4;
  const unsigned* bp = Bs + (wn * 32 * TN + col) * BF_PD + half * 4;
  float va[16], vb[TN][16];
  auto fetch = [&](int k0) {
    if constexpr (A_KC) bf_load_kc<V4>(A, g.lda, m0, g.M, k0, kend, va, tid);
    else bf_load_rc<V4>(A, g.lda, m0, g.M, k0, kend, va, tid);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (B_KC) bf_load_kc<V4>(B, g.ldb, n0 + j * BF_BM, g.N, k0, kend, vb[j], tid);
      else bf_load_rc<V4>(B, g.ldb, n0 + j * BF_BM, g.N, k0, kend, vb[j], tid);
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BF_BK) {
    if constexpr (A_KC) bf_store_kc(va, As, tid);
    else bf_store_rc(va, As, tid, do_rs ? rs : nullptr);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (B_KC) bf_store_kc(vb[j], Bs + j * BF_BM * BF_PD, tid);
      else bf_store_rc(vb[j], Bs + j * BF_BM * BF_PD, tid, nullptr);
    }
    __syncthreads();
    if (k0 + BF_BK < kend) fetch(k0 + BF_BK);            // in flight under this tile's MFMAs
#pragma unroll
    for (int kk = 0; kk < BF_BK / 16; ++kk) {
      const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(ap + kk * 8));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8_t b =
            __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(bp + j * 32 * BF_PD + kk * 8));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if constexpr (!A_KC) {
    if (do_rs) {           // 16 k-lanes hold pieces of each row sum: combine in a fixed order through LDS
      float* red = reinterpret_cast<float*>(As);     // [16 kp][64 rows]
      const int rc = (tid & 15) * 4, kp = tid >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) red[kp * 64 + rc + i] = rs[i];
      __syncthreads();
      if (tid < 64 && m0 + tid < g.M) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 64 + tid];
        g.ep.rowsum[batch][m0 + tid] = t;
      }
    }
  }

  // C/D layout of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < g.M && n < g.N) {
        if (g.splitk > 1) g.part[((long)z * g.M + m) * g.N + n] = acc[j][r];
        else epilogue_store(g, batch, m, n, acc[j][r]);
      }
    }
  }
}

__global__ void splitk_reduce_kernel(GemmArgs g) {
  const long mn = (long)g.M * g.N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int batch = blockIdx.y;
  if (i >= mn) return;
  const float* p = g.part + (long)batch * g.splitk * mn + i;
  // fixed association (4 interleaved chains, then a tree): deterministic, 8 loads in flight
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
#pragma unroll 2
  for (; k + 3 < g.splitk; k += 4) {
    s0 += p[(long)k * mn];
    s1 += p[(long)(k + 1) * mn];
    s2 += p[(long)(k + 2) * mn];
    s3 += p[(long)(k + 3) * mn];
  }
  for (; k < g.splitk; ++k) s0 += p[(long)k * mn];
  const int m = (int)(i / g.N), n = (int)(i - (long)m * g.N);
  epilogue_store(g, batch, m, n, (s0 + s1) + (s2 + s3));
}

// set by drq_gemm_batched_partial around one call: keep the split-K partials, the caller reduces them itself
thread_local bool g_leave_partials = false;   // call-scoped flag of drq_gemm_batched_partial (per host thread)
thread_local int g_last_splitk = 1;

template <int TM, int TN, bool A_KC, bool B_KC, bool V4, int BK = BK0>
int launch(const GemmArgs& g, hipStream_t st) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nbatch * g.splitk);
  // the forward form of the hidden layers (all dimensions multiples of the tiles): the variant without masks
  constexpr bool HAS_FULL = TM == 1 && TN == 1 && A_KC && B_KC && V4 && BK == BK0;
  const bool full = g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0 && g.kchunk % BK == 0;
  if constexpr (HAS_FULL) {
    if (full) hipLaunchKernelGGL((gemm_kernel<TM, TN, A_KC, B_KC, V4, BK, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<TM, TN, A_KC, B_KC, V4, BK>), grid, dim3(256), 0, st, g);
  } else
  hipLaunchKernelGGL((gemm_kernel<TM, TN, A_KC, B_KC, V4, BK>), grid, dim3(256), 0, st, g);
  DRQ_LAUNCH_CHECK();
  g_last_splitk = g.splitk;
  if (g.splitk > 1 && !g_leave_partials) {
    const long mn = (long)g.M * g.N;
    dim3 rg((unsigned)((mn + 255) / 256), g.nbatch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, st, g);
    DRQ_LAUNCH_CHECK();
  }
  return DRQ_OK;
}

template <int TM, int TN, int BK = BK0>
int dispatch(const GemmArgs& g, int a_kc, int b_kc, bool v4, hipStream_t st) {
  if (a_kc && b_kc)
    return v4 ? launch<TM, TN, true, true, true, BK>(g, st) : launch<TM, TN, true, true, false, BK>(g, st);
  if (a_kc && !b_kc)
    return v4 ? launch<TM, TN, true, false, true, BK>(g, st) : launch<TM, TN, true, false, false, BK>(g, st);
  if (!a_kc && !b_kc) return launch<TM, TN, false, false, false, BK>(g, st);
  return DRQ_EARG;
}

template <bool A_KC, bool B_KC>
int launch_bf16(const GemmArgs& g, bool v4, hipStream_t st) {
  // 64 < N <= 128 with a long reduction (the trunk layer at feature_dim 100): one 128-column tile reads A once
  const bool wide = g.N > 64 && g.N <= 128 && g.K >= 4096;
  const int bn = wide ? 2 * BF_BM : BF_BM;
  dim3 grid((g.N + bn - 1) / bn, (g.M + BF_BM - 1) / BF_BM, g.nbatch * g.splitk);
  if (wide) {
    if (v4) hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, true, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, false, 2>), grid, dim3(256), 0, st, g);
  } else if (v4) hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, false>), grid, dim3(256), 0, st, g);
  DRQ_LAUNCH_CHECK();
  g_last_splitk = g.splitk;
  if (g.splitk > 1 && !g_leave_partials) {
    const long mn = (long)g.M * g.N;
    dim3 rg((unsigned)((mn + 255) / 256), g.nbatch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, st, g);
    DRQ_LAUNCH_CHECK();
  }
  return DRQ_OK;
}

}  // namespace

// elementwise.hip (C ABI): out[n] = sum_m dy[m*ld + n]
extern "C" int drq_colsum(const float* dy, long ld, long dy_bs, float* out, long out_bs, int M, int N, int nbatch,
                          hipStream_t st);
// skinny.hip
int drq_skinny_dgrad(const float* dz, long lda, const float* w, long ldb, float* c, long ldc, int M, int N, int K,
                     const float* aux, int ldaux, int scatter_hw, hipStream_t st);

// gemm2.hip
int drq_trunk_wgrad(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                    float* rowsum, hipStream_t st);
int drq_trunk_fwd_partial(int nbatch, const float* const* A, long lda, const float* const* B, long ldb, int M, int N,
                          int K, float* ws, size_t ws_bytes, int* splitk_out, hipStream_t st);
int drq_gemm2(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb, int b_kc,
              float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
              const float* const* aux, int ldaux, float* const* rowsum, hipStream_t st);

// bf16 != 0: the bf16-MFMA kernel for every shape (fp32 storage, operands rounded when staged); see gemm_bf16_kernel
int drq_gemm_batched_any(int bf16, int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                         int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                         const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int tile,
                         int splitk, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || nbatch <= 0 || nbatch > MAXB) return DRQ_EARG;
  for (int b = 0; b < nbatch; ++b)
    if (!A[b] || !B[b] || !C[b]) return DRQ_EARG;
  if (rowsum && a_kc) return DRQ_EARG;
  if (bf16) {
    if (!a_kc && b_kc) return DRQ_EARG;
    const int cus = drq_num_cus();
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * nbatch;
    // a weight-gradient GEMM with few output tiles (the first layers: N = feature_dim (+ action_dim)) needs split-K
    // to fill the chip, and the fused bias gradient cannot be split: it gets its own pass (column sums of dy)
    if (rowsum && splitk == 0 && tiles < 2L * cus && K >= 512) {
      for (int b = 0; b < nbatch; ++b)
        if (rowsum[b]) {
          const int rc = drq_colsum(A[b], lda, 0, rowsum[b], 0, K, M, 1, st);
          if (rc != DRQ_OK) return rc;
        }
      rowsum = nullptr;
    }
    if (rowsum) splitk = 1;
    if (splitk == 0) {
      splitk = 1;
      if (tiles < 2L * cus && K >= 512) {
        splitk = (int)((3L * cus + tiles - 1) / tiles);
        const int maxs = K / 256;                    // keep >= 4 k-tiles per split
        if (splitk > maxs) splitk = maxs;
        if (splitk < 1) splitk = 1;
      }
    }
    int kchunk = ((K + splitk - 1) / splitk + 63) / 64 * 64;
    splitk = (K + kchunk - 1) / kchunk;
    if (splitk > 1 && (!ws || (size_t)nbatch * splitk * M * N * sizeof(float) > ws_bytes)) return DRQ_EWS;
    GemmArgs g{};
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    bool v4 = lda % 4 == 0 && ldb % 4 == 0;
    for (int b = 0; b < nbatch; ++b) {
      g.A[b] = A[b]; g.B[b] = B[b]; g.C[b] = C[b];
      g.ep.bias[b] = bias ? bias[b] : nullptr;
      g.ep.aux[b] = aux ? aux[b] : nullptr;
      g.ep.rowsum[b] = rowsum ? rowsum[b] : nullptr;
      v4 = v4 && al16(A[b]) && al16(B[b]);
    }
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.nbatch = nbatch; g.splitk = splitk; g.kchunk = kchunk; g.part = ws;
    g.ep.ldaux = ldaux; g.ep.relu = relu; g.ep.scatter_hw = scatter_hw;
    if (a_kc && b_kc) return launch_bf16<true, true>(g, v4, st);
    if (a_kc && !b_kc) return launch_bf16<true, false>(g, v4, st);
    return launch_bf16<false, false>(g, v4, st);
  }
  // the skinny dgrad shape of the trunk layer (K = feature_dim, N = 39200) has its own kernel; an explicit
  // tile / splitk request keeps the generic path (tests compare the two)
  if (nbatch == 1 && tile == 0 && splitk == 0 && N >= 4096 && N % 32 == 0 && !bias && !relu && a_kc && !b_kc &&
      K <= 128 && !rowsum) {
    const int rc =
        drq_skinny_dgrad(A[0], lda, B[0], ldb, C[0], ldc, M, N, K, aux ? aux[0] : nullptr, ldaux, scatter_hw, st);
    if (rc != DRQ_EARG) return rc;
  }
  // the trunk layer's weight gradient (M = feature_dim, N = 39200, K = batch) likewise
  if (nbatch == 1 && tile == 0 && splitk == 0 && scatter_hw == 0 && !a_kc && !b_kc && M <= 128 && N >= 4096 && !bias &&
      !relu && !aux) {
    const int rc = drq_trunk_wgrad(A[0], lda, B[0], ldb, C[0], ldc, M, N, K, rowsum ? rowsum[0] : nullptr, st);
    if (rc != DRQ_EARG) return rc;
  }
  // hidden x hidden layers at multiple-of-32 shapes: the LDS-free kernel (gemm2.hip)
  if (tile == 0 && splitk == 0 && scatter_hw == 0) {
    const int rc = drq_gemm2(nbatch, A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, relu, aux, ldaux, rowsum, st);
    if (rc != DRQ_EARG) return rc;
  }
  const int cus = drq_num_cus();
  const long t_big = (long)((M + 63) / 64) * ((N + 63) / 64) * nbatch;
  const long t_small = (long)((M + 31) / 32) * ((N + 31) / 32) * nbatch;
  if (tile == 0) tile = (t_big >= 2L * cus) ? 2 : 1;
  const long tiles = tile == 2 ? t_big : (tile >= 3 ? (t_big + t_small) / 2 : t_small);
  if (rowsum) splitk = 1;
  if (splitk == 0) {
    splitk = 1;
    if (tiles < 2L * cus && K >= 256) {
      splitk = (int)((3L * cus + tiles - 1) / tiles);
      const int maxs = K / 128;                      // keep >= 4 k-tiles per split
      if (splitk > maxs) splitk = maxs;
      if (splitk < 1) splitk = 1;
    }
  }
  int kchunk = ((K + splitk - 1) / splitk + 63) / 64 * 64;
  splitk = (K + kchunk - 1) / kchunk;
  if (splitk > 1 && (!ws || (size_t)nbatch * splitk * M * N * sizeof(float) > ws_bytes)) return DRQ_EWS;
  GemmArgs g{};
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  bool v4 = (K % 4 == 0) && (a_kc || b_kc);
  if (a_kc) v4 = v4 && lda % 4 == 0;
  if (b_kc) v4 = v4 && ldb % 4 == 0;
  for (int b = 0; b < nbatch; ++b) {
    g.A[b] = A[b]; g.B[b] = B[b]; g.C[b] = C[b];
    g.ep.bias[b] = bias ? bias[b] : nullptr;
    g.ep.aux[b] = aux ? aux[b] : nullptr;
    g.ep.rowsum[b] = rowsum ? rowsum[b] : nullptr;
    if (a_kc) v4 = v4 && al16(A[b]);
    if (b_kc) v4 = v4 && al16(B[b]);
  }
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.nbatch = nbatch; g.splitk = splitk; g.kchunk = kchunk; g.part = ws;
  g.ep.ldaux = ldaux; g.ep.relu = relu; g.ep.scatter_hw = scatter_hw;
  if (tile == 3) return dispatch<2, 1>(g, a_kc, b_kc, v4, st);   // 64x32 block tile
  if (tile == 4) return dispatch<1, 2>(g, a_kc, b_kc, v4, st);   // 32x64
  if (tile == 5) return dispatch<1, 1, 64>(g, a_kc, b_kc, v4, st);   // 32x32, k-tile 64 (one barrier per 64 k)
  if (tile == 6) return dispatch<2, 2, 64>(g, a_kc, b_kc, v4, st);   // 64x64, k-tile 64
  return tile == 2 ? dispatch<2, 2>(g, a_kc, b_kc, v4, st) : dispatch<1, 1>(g, a_kc, b_kc, v4, st);
}

extern "C" {

// See include/drqv2_hip.h.  Host arrays of nbatch (<= 8) device pointers; bias/aux/rowsum arrays may be null.
DRQ_API int drq_gemm_batched_f32(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                         int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                         const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int tile,
                         int splitk, float* ws, size_t ws_bytes, hipStream_t st) {
  return drq_gemm_batched_any(0, nbatch, A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, relu, aux, ldaux, rowsum,
                              scatter_hw, tile, splitk, ws, ws_bytes, st);
}

DRQ_API int drq_gemm_batched_bf16(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                          int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                          const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int splitk,
                          float* ws, size_t ws_bytes, hipStream_t st) {
  return drq_gemm_batched_any(1, nbatch, A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, relu, aux, ldaux, rowsum,
                              scatter_hw, 0, splitk, ws, ws_bytes, st);
}

}  // extern "C"

// internal (step.hip): the forward form with the split-K sum left to the caller, in either precision: a split-K
// result stays in `ws` as partials [nbatch*splitk][M][N] (no bias, no epilogue) and *splitk_out says how many there
// are per problem; 1 = the result went to C as usual
int drq_gemm_batched_partial_any(int bf16, int nbatch, const float* const* A, long lda, int a_kc, const float* const* B,
                                 long ldb, int b_kc, float* const* C, long ldc, int M, int N, int K,
                                 const float* const* bias, float* ws, size_t ws_bytes, int* splitk_out, hipStream_t st) {
  // the fp32 trunk forward (k-contiguous operands, N <= 64, long K) has its own kernel
#ifdef DRQ_DEV
  static const bool no_trunk = getenv("DRQ_NO_TRUNK_KERNEL") != nullptr;     // development build: A/B against the tiled GEMM
#else
  constexpr bool no_trunk = false;
#endif
  if (!bf16 && a_kc && b_kc && N <= 128 && K >= 4096 && ldc == N && !no_trunk) {
    int sk = 1;
    const int rc = drq_trunk_fwd_partial(nbatch, A, lda, B, ldb, M, N, K, ws, ws_bytes, &sk, st);
    if (rc == DRQ_OK) {
      if (splitk_out) *splitk_out = sk;
      return DRQ_OK;
    }
    if (rc != DRQ_EARG) return rc;
  }
  g_leave_partials = true;
  g_last_splitk = 1;
  const int rc = drq_gemm_batched_any(bf16, nbatch, A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, 0, nullptr, 0,
                                      nullptr, 0, bf16 ? 0 : 1, 0, ws, ws_bytes, st);
  g_leave_partials = false;
  if (splitk_out) *splitk_out = g_last_splitk;
  return rc;
}

extern "C" {

DRQ_API int drq_gemm_batched_partial(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                             int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias,
                             float* ws, size_t ws_bytes, int* splitk_out, hipStream_t st) {
  return drq_gemm_batched_partial_any(0, nbatch, A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, bias, ws, ws_bytes,
                                      splitk_out, st);
}

// strided-batch form of the same call
DRQ_API int drq_gemm_f32(const float* A, long lda, int a_kc, const float* B, long ldb, int b_kc, float* C, long ldc,
                 int M, int N, int K, int nbatch, long a_bs, long b_bs, long c_bs, const float* bias, long bias_bs,
                 int relu, const float* aux, int ldaux, long aux_bs, int scatter_hw, int tile, int splitk,
                 float* ws, size_t ws_bytes, hipStream_t st) {
  if (!A || !B || !C || nbatch <= 0 || nbatch > MAXB) return DRQ_EARG;
  const float *Ap[MAXB], *Bp[MAXB], *bp[MAXB], *xp[MAXB];
  float* Cp[MAXB];
  for (int b = 0; b < nbatch; ++b) {
    Ap[b] = A + b * a_bs; Bp[b] = B + b * b_bs; Cp[b] = C + b * c_bs;
    bp[b] = bias ? bias + b * bias_bs : nullptr;
    xp[b] = aux ? aux + b * aux_bs : nullptr;
  }
  return drq_gemm_batched_f32(nbatch, Ap, lda, a_kc, Bp, ldb, b_kc, Cp, ldc, M, N, K, bias ? bp : nullptr, relu,
                              aux ? xp : nullptr, ldaux, nullptr, scatter_hw, tile, splitk, ws, ws_bytes, st);
}

}  // extern "C"
