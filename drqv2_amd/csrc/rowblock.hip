// Row-local stages of the actor / critic heads fused with the first MLP layer that consumes them (round 3).
//
// Between the trunk GEMM (K = 39200) and the hidden x hidden layers the reference runs, per row of the batch, only
// row-local work: LayerNorm+tanh of the trunk output (drqv2.py:74-75,100-101), the [h, action] concatenation (:117),
// the policy's output layer Linear(hidden, A) with tanh and the clipped-noise sample (:81,88-92, utils.py:117-126),
// and the first layers Linear(F or F+A, hidden)+ReLU of the policy / Q MLPs (:77,103,108) whose reduction is only
// 50..121 long.  As launches of their own each of these is a 5-8 us dependent step (measured floor of a launch on this
// stream: 4.7 us); here a workgroup owns 16 rows and does the row-local stage into LDS, then multiplies the rows by a
// 256-column slice of the first-layer weights on v_mfma_f32_16x16x4_f32 (16 rows = one MFMA tile):
//   lnl1_kernel    LayerNorm+tanh (from the trunk's split-K records, same arithmetic and association as
//                  ln_tanh_fwd_kernel: bit-identical h, xhat, rstd) [+ action columns] -> up to two first layers
//   polout_kernel  policy output layer (16 rows x A on the MFMA, K = hidden split over the four waves, summed in wave
//                  order) + tanh + sample -> for the next_obs rows the target critic's two first layers on [h, a']
// The first-layer product is a k-ordered f32 fma chain per output (lane's k values 4V*q + V*kq + e, both operands).
#include "common.h"
#include "../../include/drqv2_hip.h"

namespace {

constexpr int RB_ROWS = 16;        // rows per workgroup
constexpr int RB_COLS = 256;       // first-layer columns per workgroup (64 per wave: four 16-column MFMA tiles)
constexpr int RB_KPMAX = 128;      // padded reduction length of the first layer (F + A <= 128)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum_rb(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// First layer on the wave's 64 columns starting at c0:  y[row0 + r][n] = relu(b[n] + sum_k xs[r][k] w[n][k]), n < H,
// r < nrows.  xs: LDS [16][KPT] zero-padded; w [H][K] row-major.  Two steps so that the weight loads (which depend on
// nothing the kernel computes) are in flight while the row-local stage runs: a kernel of this size is a chain of
// dependent memory round trips of 1-2 us each, and every load issued late is one more of them.  Everything is unrolled
// over compile-time counts and no load sits behind a branch (a branch around a load makes hipcc wait for it at once).
template <int KPT>
struct L1W {
  float bv[4][KPT / 4];          // this lane's weights of the wave's four 16-column tiles
  float bias[4];
};

// V = floats per lane load (K % V == 0; 16-byte aligned rows for V = 4, 8-byte for V = 2)
template <int V, int KPT>
__device__ __forceinline__ void l1_load(L1W<KPT>& W, const float* w, int K, int H, const float* b, int c0, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  constexpr int NQ = KPT / (4 * V);                          // vector loads per lane and tile
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((long)H * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, b ? H * 4 : 0, 0x00020000);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int col = c0 + u * 16 + n;
    // columns past H and k past the row's end (it then reads the next row: finite, times a zero of xs) are harmless;
    // past the buffer the range check returns 0
    const unsigned voff = (unsigned)(((long)col * K + V * kq) * 4);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if constexpr (V == 4) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, (unsigned)(16 * V * q), 0));
        W.bv[u][4 * q] = t[0]; W.bv[u][4 * q + 1] = t[1]; W.bv[u][4 * q + 2] = t[2]; W.bv[u][4 * q + 3] = t[3];
      } else if constexpr (V == 2) {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs, voff, (unsigned)(16 * V * q), 0));
        W.bv[u][2 * q] = t[0]; W.bv[u][2 * q + 1] = t[1];
      } else {
        W.bv[u][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, voff, (unsigned)(16 * V * q), 0));
      }
    }
    W.bias[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)col * 4u, 0, 0));   // 0 past H / no bias
  }
}

template <int V, int KPT>
__device__ __forceinline__ void l1_compute(const L1W<KPT>& W, const float* xs, int H, float* y, long ldy, int row0,
                                           int nrows, int c0, int lane) {
  const int n = lane & 15, kq = lane >> 4;
  constexpr int NQ = KPT / (4 * V);
  // A operand: row n of xs (the MFMA's row index is lane & 15), this lane's k values 4V*q + V*kq + e
  float av[KPT / 4];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float* p = xs + n * KPT + 4 * V * q + V * kq;
    if constexpr (V == 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p);
      av[4 * q] = t[0]; av[4 * q + 1] = t[1]; av[4 * q + 2] = t[2]; av[4 * q + 3] = t[3];
    } else if constexpr (V == 2) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(p);
      av[2 * q] = t[0]; av[2 * q + 1] = t[1];
    } else {
      av[q] = p[0];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int col = c0 + u * 16 + n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < KPT / 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], W.bv[u][e], acc, 0, 0, 0);
    if (col < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;                          // C/D layout: column = lane & 15, row = 4*(lane>>4) + r
        if (row < nrows) {
          const float v = acc[r] + W.bias[u];
          y[(long)(row0 + row) * ldy + col] = v > 0.f ? v : 0.f;
        }
      }
    }
  }
}

// vk: 0 / 1 / 2 = 4 / 2 / 1 floats per weight load (uniform over the workgroup)
template <int KPT>
__device__ __forceinline__ void l1_load_any(int vk, L1W<KPT>& W, const float* w, int K, int H, const float* b, int c0, int lane) {
  if (vk == 0) l1_load<4, KPT>(W, w, K, H, b, c0, lane);
  else if (vk == 1) l1_load<2, KPT>(W, w, K, H, b, c0, lane);
  else l1_load<1, KPT>(W, w, K, H, b, c0, lane);
}
template <int KPT>
__device__ __forceinline__ void l1_compute_any(int vk, const L1W<KPT>& W, const float* xs, int H, float* y, long ldy,
                                               int row0, int nrows, int c0, int lane) {
  if (vk == 0) l1_compute<4, KPT>(W, xs, H, y, ldy, row0, nrows, c0, lane);
  else if (vk == 1) l1_compute<2, KPT>(W, xs, H, y, ldy, row0, nrows, c0, lane);
  else l1_compute<1, KPT>(W, xs, H, y, ldy, row0, nrows, c0, lane);
}
__device__ __forceinline__ int vec_kind(int K) { return (K & 3) == 0 ? 0 : ((K & 1) == 0 ? 1 : 2); }

// ---- LayerNorm + tanh (+ tail columns) -> first layer(s) ---------------------------------------------------------
struct LnL1Job {
  const float* part;      // split-K records of this problem: element (s, row, f) at part[s*slab + row*F + f]
  const float* z;         // or (splitk == 0) the pre-norm input [rows][F] with the bias already added
  const float* bias;      // trunk bias [F] added to the summed records (may be null)
  const float* gamma;
  const float* beta;
  float* out;             // [rows][ldo]: tanh(LN) in columns [0,F), the tail in [F, F+tail_n)
  float* xhat;            // [rows][F] or null
  float* rstd;            // [rows] or null
  const float* tail;      // [rows][tail_ld] or null
  int ldo, tail_ld, tail_n;
  int rows;
  int nheads;             // first layers fed by [out row]: 0, 1 or 2
  const float* w[2];      // [H][F + tail_n]
  const float* b[2];
  float* y[2];            // [rows][H]
  int blk0, ncg;          // first workgroup of the job, column groups per row block (nheads * ceil(H/256), >= 1)
};
struct LnL1Args {
  LnL1Job job[4];
  int njobs, F, H, splitk;
  long slab;              // floats between consecutive split-K records of one problem
};

// The split-K sum of ln_tanh_fwd_kernel (elementwise.hip sum_partials: four interleaved chains over the records, then a
// tree) for FOUR rows of one lane at once.  p[r]: the element in record 0 (a valid address also for rows / features
// that do not exist: the caller clamps, and discards the result).  All 4 x 16 loads of a batch are in flight together.
__device__ __forceinline__ void sum_partials4(const float* const (&p)[4], long slab, int splitk, float (&out)[4]) {
  float s[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s[r][c] = 0.f;
  for (int k0 = 0; k0 < splitk; k0 += 16) {
    float v[4][16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 16; ++u) v[r][u] = p[r][(long)min(k0 + u, splitk - 1) * slab];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 16; ++u) s[r][u & 3] += (k0 + u < splitk) ? v[r][u] : 0.f;     // x + 0 = x: the same sums
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = (s[r][0] + s[r][1]) + (s[r][2] + s[r][3]);
}

// KPT: padded reduction length of the first layers (64 or 128); NQF: 64-feature chunks of a LayerNorm row (F <= 64*NQF)
template <int KPT, int NQF>
__global__ __launch_bounds__(256, 1) void lnl1_kernel(LnL1Args a) {
  __shared__ __attribute__((aligned(16))) float xs[RB_ROWS * KPT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ji = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < a.njobs && (int)blockIdx.x >= a.job[j].blk0) ji = j;
  const LnL1Job& J = a.job[ji];
  const int lb = (int)blockIdx.x - J.blk0;
  const int rb = lb / J.ncg, cg = lb - rb * J.ncg;
  const int row0 = rb * RB_ROWS;
  const int nrows = min(RB_ROWS, J.rows - row0);
  const int F = a.F, K = F + J.tail_n;
  const bool writer = cg == 0;                     // one column group of the row block writes the row-local results
  // the first layer's weights of this wave's columns: requested before anything else
  const int cgh = (a.H + RB_COLS - 1) / RB_COLS;
  const int head = J.nheads > 0 ? cg / cgh : 0, c0 = (cg - head * cgh) * RB_COLS + wid * 64;
  const int vk = vec_kind(K);
  const bool do_l1 = J.nheads > 0 && c0 < a.H;
  L1W<KPT> W;
  if (do_l1) l1_load_any<KPT>(vk, W, J.w[head], K, a.H, J.b[head], c0, lane);

  // ---- LayerNorm + tanh: wave w takes rows 4w .. 4w+3 of the block, lane = feature f + 64 q (ln_tanh_fwd_kernel's
  // arithmetic: identical results).  Loads use clamped (always valid) rows and features; results of rows / features
  // that do not exist are discarded.
  bool ok[4];
  int rr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ok[r] = 4 * wid + r < nrows;
    rr[r] = row0 + min(4 * wid + r, nrows - 1);
  }
  float v[4][NQF], gm[NQF], bt[NQF], tl[4];
  {
    const int tc = min(lane, J.tail_n > 0 ? J.tail_n - 1 : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) tl[r] = J.tail ? J.tail[(long)rr[r] * J.tail_ld + tc] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NQF; ++q) {
    const int f = lane + 64 * q;
    const int fc = min(f, F - 1);
    gm[q] = J.gamma[fc];
    bt[q] = J.beta[fc];
    float t[4];
    if (a.splitk > 0) {                            // uniform over the launch
      const float* pp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[r] = J.part + (long)rr[r] * F + fc;
      sum_partials4(pp, a.slab, a.splitk, t);
      const float bb = J.bias ? J.bias[fc] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] += bb;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = J.z[(long)rr[r] * F + fc];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r][q] = f < F ? t[r] : 0.f;
  }
  for (int i = tid; i < RB_ROWS * KPT; i += 256) xs[i] = 0.f;
  __syncthreads();
  float mean[4], rstd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NQF; ++q) s += v[r][q];
    mean[r] = wave_sum_rb(s) / (float)F;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NQF; ++q) {
      const float d = lane + 64 * q < F ? v[r][q] - mean[r] : 0.f;
      ss += d * d;
    }
    // the sum of squares is a ROUNDED value before it enters the butterfly (as in ln_tanh_fwd_kernel): without this
    // hipcc's cross-statement contraction folds the last product into the first butterfly add as an fma with an
    // unrounded square on one side (1 ulp of rstd in ~2 % of the rows)
    asm volatile("" : "+v"(ss));
    const float var = wave_sum_rb(ss) / (float)F;
    rstd[r] = 1.0f / sqrtf(var + 1e-5f);
  }
#pragma unroll
  for (int q = 0; q < NQF; ++q) {
    const int f = lane + 64 * q;
    if (f < F) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ok[r]) {
          const float xh = (v[r][q] - mean[r]) * rstd[r];
          const float y = tanhf(xh * gm[q] + bt[q]);
          xs[(4 * wid + r) * KPT + f] = y;
          if (writer) {
            J.out[(long)rr[r] * J.ldo + f] = y;
            if (J.xhat) J.xhat[(long)rr[r] * F + f] = xh;
          }
        }
      }
    }
  }
  if (writer && lane == 0 && J.rstd) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[r]) J.rstd[rr[r]] = rstd[r];
  }
  if (J.tail && lane < J.tail_n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ok[r]) {
        xs[(4 * wid + r) * KPT + F + lane] = tl[r];
        if (writer) J.out[(long)rr[r] * J.ldo + F + lane] = tl[r];
      }
    }
  }
  if (J.nheads == 0) return;
  __syncthreads();
  // ---- first layer(s): column group cg = (head, 256-column slice); wave = 64 columns of it
  if (do_l1) l1_compute_any<KPT>(vk, W, xs, a.H, J.y[head], a.H, row0, nrows, c0, lane);
}

// ---- policy output layer + sample -> first layers of the target critic -------------------------------------------
struct PolOutL1Args {
  const float* p2;        // [rows][H] post-ReLU input of the policy's output layer
  const float* w3;        // [A][H]
  const float* b3;        // [A]
  float* p3;              // [rows][A] pre-tanh output
  int rows, srow0;        // rows >= srow0: job "hi", rows < srow0: job "lo"
  int H, A, F;
  float std, clip;
  int use_clip;
  // hi rows (next_obs): a' = sample(noise_hi) -> columns [F, F+A) of ha_hi (row r - srow0); mu_hi optional
  const float* noise_hi;
  float* mu_hi;
  float* ha_hi;           // [rows - srow0][lda_hi]: columns [0,F) already hold h (LayerNorm output)
  long lda_hi;
  // lo rows (obs): the actor update's own draw from the same policy output (may be null)
  const float* noise_lo;
  float* mu_lo;
  float* ha_lo;           // action columns only are written (row stride lda_lo, column offset F)
  long lda_lo;
  // first layers on the hi rows' [h, a']: nheads = 0 or 2
  int nheads;
  const float* w[2];      // [H][F + A]
  const float* b[2];
  float* y[2];            // [rows - srow0][H]
  int nblk_lo, ncg;       // workgroups of the lo rows (one per row block); column groups per hi row block
};

// KPT as above; KW16 = (H / 4) / 16: 16-byte loads per lane of a wave's share of the output layer's reduction;
// NT = 16-wide tiles of action outputs (1: A <= 16, 2: A <= 32)
template <int KPT, int KW16, int NT>
__global__ __launch_bounds__(256, 1) void polout_kernel(PolOutL1Args a) {
  __shared__ __attribute__((aligned(16))) float xs[RB_ROWS * KPT];
  __shared__ float red[3 * 2 * 4 * 64];            // partial output tiles of waves 1..3: [wave - 1][tile][r][lane]
  __shared__ float p3s[RB_ROWS][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = (int)blockIdx.x;
  const bool hi = bx >= a.nblk_lo;
  const int lb = hi ? bx - a.nblk_lo : bx;
  const int rb = hi ? lb / a.ncg : lb, cg = hi ? lb - rb * a.ncg : 0;
  const int jrows = hi ? a.rows - a.srow0 : a.srow0;
  const int jrow0 = rb * RB_ROWS;                  // row inside the job
  const int nrows = min(RB_ROWS, jrows - jrow0);
  const int grow0 = (hi ? a.srow0 : 0) + jrow0;    // row of p2 / p3
  const int H = a.H, A = a.A, F = a.F, K = F + A;
  const bool writer = cg == 0;
  // everything the kernel reads that it does not compute itself is requested first: the target critic's first-layer
  // weights of this wave's columns, then this wave's share of the output layer's operands (16 + 16*NT loads per lane)
  const int cgh = (H + RB_COLS - 1) / RB_COLS;
  const bool want_l1 = hi && a.nheads > 0;
  const int head = want_l1 ? cg / cgh : 0, c0 = (cg - head * cgh) * RB_COLS + wid * 64;
  const bool do_l1 = want_l1 && c0 < H;
  const int vk = vec_kind(K);
  L1W<KPT> W;
  if (do_l1) l1_load_any<KPT>(vk, W, a.w[head], K, H, a.b[head], c0, lane);
  // ---- p3 = p2 W3^T: 16 rows x (A <= 32) outputs on the MFMA, the reduction over H split over the four waves
  const int n = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)a.p2, 0, (int)((long)a.rows * H * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, (int)((long)A * H * 4), 0x00020000);
  constexpr int KW = 16 * KW16;                    // this wave's share of the reduction (H = 4 * KW)
  const int arow = min(grow0 + n, a.rows - 1);
  const unsigned avoff = (unsigned)(((long)arow * H + wid * KW + 4 * kq) * 4);
  f32x4 acc[NT];
  unsigned bvoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bvoff[t] = (unsigned)(((long)(16 * t + n) * H + wid * KW + 4 * kq) * 4);   // rows >= A: out of range -> 0
  }
  {
    f32x4 av[KW16], bv[NT][KW16];
#pragma unroll
    for (int q = 0; q < KW16; ++q) {
      av[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, (unsigned)(64 * q), 0));
#pragma unroll
      for (int t = 0; t < NT; ++t)
        bv[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, bvoff[t], (unsigned)(64 * q), 0));
    }
    // the h part of the hi rows' [h, a'] (written by the LayerNorm launch) and the noise: requested now as well
    if (want_l1) {
      for (int i = tid; i < RB_ROWS * KPT; i += 256) {
        const int r = i / KPT, f = i - r * KPT;
        xs[i] = (r < nrows && f < F) ? a.ha_hi[(long)(jrow0 + r) * a.lda_hi + f] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < KW16; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], bv[t][q][e], acc[t], 0, 0, 0);
  }
  if (wid > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wid - 1) * 2 + t) * 4 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = 16 * t + n;                    // output index (C/D layout: column = lane & 15, row = 4*(lane>>4) + r)
      const float bj = a.b3[min(j, A - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[t][r];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += red[((w * 2 + t) * 4 + r) * 64 + lane];
        if (j < A) p3s[4 * kq + r][j] = v + bj;
      }
    }
  }
  __syncthreads();
  // ---- tanh, clipped noise, straight-through clamp (utils.py:117-126; sample_action_kernel's arithmetic)
  for (int i = tid; i < RB_ROWS * A; i += 256) {
    const int r = i / A, j = i - r * A;
    if (r < nrows) {
      const float pre = p3s[r][j];
      if (writer) a.p3[(long)(grow0 + r) * A + j] = pre;
      const float* nz = hi ? a.noise_hi : a.noise_lo;
      if (nz) {
        const long jr = jrow0 + r;
        const float mu = tanhf(pre);
        float eps = nz[jr * A + j] * a.std;
        if (a.use_clip) eps = fminf(fmaxf(eps, -a.clip), a.clip);
        const float lo = (float)(-1.0 + 1e-6), hic = (float)(1.0 - 1e-6);
        const float act = fminf(fmaxf(mu + eps, lo), hic);
        float* mo = hi ? a.mu_hi : a.mu_lo;
        float* ao = hi ? a.ha_hi : a.ha_lo;
        const long ld = hi ? a.lda_hi : a.lda_lo;
        if (writer) {
          if (mo) mo[jr * A + j] = mu;
          ao[jr * ld + F + j] = act;
        }
        if (want_l1) xs[r * KPT + F + j] = act;
      }
    }
  }
  if (!want_l1) return;
  __syncthreads();
  // ---- [h, a'] rows -> first layers of the two target heads
  if (do_l1) l1_compute_any<KPT>(vk, W, xs, H, a.y[head], H, jrow0, nrows, c0, lane);
}

}  // namespace

// internal (step.hip) + C ABI wrappers below.  njobs (<= 4) LayerNorm+tanh problems of one (F, splitk, slab) with their
// first layers; DRQ_EARG = shape not eligible (the caller issues the separate launches).
int drq_lnl1_fwd(int njobs, const float* const* part, const float* const* z, const float* const* bias,
                 const float* const* gamma, const float* const* beta, float* const* out, const int* ldo,
                 float* const* xhat, float* const* rstd, const float* const* tail, const int* tail_ld, const int* tail_n,
                 const int* rows, const int* nheads, const float* const* w, const float* const* b, float* const* y, int F,
                 int H, int splitk, long slab, hipStream_t st) {
  if (njobs <= 0 || njobs > 4 || !gamma || !beta || !out || !ldo || !rows || !nheads || F <= 0 || F > 256 || H <= 0)
    return DRQ_EARG;
  if (splitk < 0 || (splitk > 0 && !part) || (splitk == 0 && !z)) return DRQ_EARG;
  LnL1Args a{};
  a.njobs = njobs; a.F = F; a.H = H; a.splitk = splitk; a.slab = slab;
  int blk = 0;
  const int cgh = (H + RB_COLS - 1) / RB_COLS;
  for (int j = 0; j < njobs; ++j) {
    LnL1Job& J = a.job[j];
    J.part = part ? part[j] : nullptr;
    J.z = z ? z[j] : nullptr;
    if ((splitk > 0 && !J.part) || (splitk == 0 && !J.z) || !gamma[j] || !beta[j] || !out[j] || rows[j] <= 0)
      return DRQ_EARG;
    J.bias = bias ? bias[j] : nullptr;
    J.gamma = gamma[j]; J.beta = beta[j]; J.out = out[j]; J.ldo = ldo[j];
    J.xhat = xhat ? xhat[j] : nullptr;
    J.rstd = rstd ? rstd[j] : nullptr;
    J.tail = tail ? tail[j] : nullptr;
    J.tail_ld = (tail && tail_ld) ? tail_ld[j] : 0;
    J.tail_n = (J.tail && tail_n) ? tail_n[j] : 0;
    if (J.tail_n < 0 || J.tail_n > 64) return DRQ_EARG;
    J.rows = rows[j];
    J.nheads = nheads[j];
    if (J.nheads < 0 || J.nheads > 2) return DRQ_EARG;
    if (J.nheads > 0 && F + J.tail_n > RB_KPMAX) return DRQ_EARG;
    for (int h = 0; h < J.nheads; ++h) {
      if (!w || !y || !w[2 * j + h] || !y[2 * j + h]) return DRQ_EARG;
      if (((uintptr_t)w[2 * j + h] & 15)) return DRQ_EARG;
      J.w[h] = w[2 * j + h]; J.b[h] = b ? b[2 * j + h] : nullptr; J.y[h] = y[2 * j + h];
    }
    if (J.nheads > 0 && (long)H * (F + J.tail_n) * 4 >= (1L << 31)) return DRQ_EARG;
    J.blk0 = blk;
    J.ncg = J.nheads > 0 ? J.nheads * cgh : 1;
    blk += ((J.rows + RB_ROWS - 1) / RB_ROWS) * J.ncg;
  }
  int kmax = 0;
  for (int j = 0; j < njobs; ++j)
    if (a.job[j].nheads > 0) kmax = kmax > F + a.job[j].tail_n ? kmax : F + a.job[j].tail_n;
  if (kmax <= 64 && F <= 64) hipLaunchKernelGGL((lnl1_kernel<64, 1>), dim3(blk), dim3(256), 0, st, a);
  else if (F <= 64) hipLaunchKernelGGL((lnl1_kernel<128, 1>), dim3(blk), dim3(256), 0, st, a);
  else if (F <= 128) hipLaunchKernelGGL((lnl1_kernel<128, 2>), dim3(blk), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((lnl1_kernel<128, 4>), dim3(blk), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// policy output layer + sampling (+ the two first layers of a critic on the hi rows' [h, a'])
int drq_polout_l1_fwd(const float* p2, const float* w3, const float* b3, float* p3, int rows, int srow0, int H, int A,
                      int F, float std, float clip, int use_clip, const float* noise_hi, float* mu_hi, float* ha_hi,
                      long lda_hi, const float* noise_lo, float* mu_lo, float* ha_lo, long lda_lo, int nheads,
                      const float* const* w, const float* const* b, float* const* y, hipStream_t st) {
  if (!p2 || !w3 || !b3 || !p3 || rows <= 0 || srow0 < 0 || srow0 > rows || H <= 0 || H % 256 || A <= 0 || A > 32 ||
      F <= 0 || F + A > RB_KPMAX)
    return DRQ_EARG;
  if (((uintptr_t)p2 & 15) || ((uintptr_t)w3 & 15)) return DRQ_EARG;
  if ((long)rows * H * 4 >= (1L << 31)) return DRQ_EARG;
  if (nheads != 0 && nheads != 2) return DRQ_EARG;
  if (srow0 < rows && (!noise_hi || !ha_hi)) return DRQ_EARG;
  if (noise_lo && !ha_lo) return DRQ_EARG;
  PolOutL1Args a{};
  a.p2 = p2; a.w3 = w3; a.b3 = b3; a.p3 = p3; a.rows = rows; a.srow0 = srow0; a.H = H; a.A = A; a.F = F;
  a.std = std; a.clip = clip; a.use_clip = use_clip;
  a.noise_hi = noise_hi; a.mu_hi = mu_hi; a.ha_hi = ha_hi; a.lda_hi = lda_hi;
  a.noise_lo = noise_lo; a.mu_lo = mu_lo; a.ha_lo = ha_lo; a.lda_lo = lda_lo;
  a.nheads = nheads;
  for (int h = 0; h < nheads; ++h) {
    if (!w || !y || !w[h] || !y[h] || ((uintptr_t)w[h] & 15)) return DRQ_EARG;
    a.w[h] = w[h]; a.b[h] = b ? b[h] : nullptr; a.y[h] = y[h];
  }
  if (nheads > 0 && (long)H * (F + A) * 4 >= (1L << 31)) return DRQ_EARG;
  const int cgh = (H + RB_COLS - 1) / RB_COLS;
  a.nblk_lo = (srow0 + RB_ROWS - 1) / RB_ROWS;
  a.ncg = nheads > 0 ? nheads * cgh : 1;
  const int nblk_hi = ((rows - srow0 + RB_ROWS - 1) / RB_ROWS) * a.ncg;
  if (a.nblk_lo + nblk_hi <= 0) return DRQ_EARG;
  const dim3 grid(a.nblk_lo + nblk_hi);
  const bool k64 = F + A <= 64;
#define POL_LAUNCH(KW16, NT)                                                                            \
  do {                                                                                                  \
    if (k64) hipLaunchKernelGGL((polout_kernel<64, KW16, NT>), grid, dim3(256), 0, st, a);             \
    else hipLaunchKernelGGL((polout_kernel<128, KW16, NT>), grid, dim3(256), 0, st, a);                \
  } while (0)
  if (H == 1024) { if (A <= 16) POL_LAUNCH(16, 1); else POL_LAUNCH(16, 2); }
  else if (H == 512) { if (A <= 16) POL_LAUNCH(8, 1); else POL_LAUNCH(8, 2); }
  else if (H == 256) { if (A <= 16) POL_LAUNCH(4, 1); else POL_LAUNCH(4, 2); }
  else return DRQ_EARG;
#undef POL_LAUNCH
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

extern "C" {

// See include/drqv2_hip.h
DRQ_API int drq_ln_l1_fwd(int njobs, const float* const* part, const float* const* z, const float* const* bias,
                          const float* const* gamma, const float* const* beta, float* const* out, const int* ldo,
                          float* const* xhat, float* const* rstd, const float* const* tail, const int* tail_ld,
                          const int* tail_n, const int* rows, const int* nheads, const float* const* w,
                          const float* const* b, float* const* y, int F, int H, int splitk, long slab,
                          drq_stream_t stream) {
  return drq_lnl1_fwd(njobs, part, z, bias, gamma, beta, out, ldo, xhat, rstd, tail, tail_ld, tail_n, rows, nheads, w, b,
                      y, F, H, splitk, slab, (hipStream_t)stream);
}

DRQ_API int drq_policy_out_l1_fwd(const float* p2, const float* w3, const float* b3, float* p3, int rows, int srow0, int H,
                                  int A, int F, float std, float clip, int use_clip, const float* noise_hi, float* mu_hi,
                                  float* ha_hi, long lda_hi, const float* noise_lo, float* mu_lo, float* ha_lo,
                                  long lda_lo, int nheads, const float* const* w, const float* const* b,
                                  float* const* y, drq_stream_t stream) {
  return drq_polout_l1_fwd(p2, w3, b3, p3, rows, srow0, H, A, F, std, clip, use_clip, noise_hi, mu_hi, ha_hi, lda_hi,
                           noise_lo, mu_lo, ha_lo, lda_lo, nheads, w, b, y, (hipStream_t)stream);
}

}  // extern "C"
