// 3x3 "valid" convolution of the DrQ-v2 encoder on the f32 matrix cores of gfx950.
//
// Reference op: nn.Conv2d(Cin,32,3,stride) + ReLU, drqv2.py:55-59, and its autograd
// backward (dgrad for conv2..4, wgrad for conv1..4).
//
// Forward / dgrad (conv3x3_kernel): implicit GEMM  D[cout][pixel] += W[cout][k] * X[k][pixel] with
// v_mfma_f32_32x32x2_f32 (bit-exact f32 fma chain).  k is ordered (channel pair c, tap t): lanes 0-31
// take channel 2c, lanes 32-63 channel 2c+1.  A = weights, staged ONCE per workgroup into LDS in
// MFMA-lane order ([step][lane]: one conflict-free ds_read_b32 per MFMA); B = input, one dword per lane
// per MFMA through a bounds-checked buffer load (each element is re-used by 32 output channels inside
// the MFMA and by the 3x3 taps through L1).  A "pixel tile" is 32 consecutive output pixels of the
// flattened (sample,y,x) index, so tiles are always full; a workgroup owns a contiguous run of tiles
// (neighbouring tiles share input rows in L1/L2).  Few registers per wave (16 accumulators + loads in
// flight) -> 4+ waves per SIMD hide the memory latency; the MFMA pipe is the only shared resource.
// The dgrad of a stride-1 3x3 valid conv is the same kernel run on the zero-padded (pad 2) output
// gradient with W transposed and flipped (gather mode 1), epilogue = ReLU mask.
//
// Wgrad (conv3x3_wgrad_kernel): D[cout][col] += dY[cout][pixel] * X[pixel][col], reduction over all
// B*Hout*Wout pixels.  Each wave owns a contiguous run of output rows; per row it stages the dY row and
// the three input rows through a wave-private LDS tile (row-per-instruction coalesced global loads into
// registers, issued one row ahead so they fly under the MFMA loop; odd LDS pitches make the
// channel-strided operand reads bank-conflict free) and keeps all 9 taps x 32x32 accumulators in
// registers.  Partials: 4 waves -> LDS -> one record per workgroup -> fixed-order reduction kernel
// (deterministic, no float atomics).
#include "common.h"
#include <stdlib.h>

namespace {

struct ConvArgs {
  const float* x;      // [NB][CIN][HIN][HIN]
  const float* w;      // canonical [32][CINW][3][3]
  const float* bias;   // [32] or null
  const float* mask;   // [NB][32][HOUT][HOUT] or null : out *= (mask > 0)
  float* y;
  long y_bs, y_cs, y_rs, y_off;   // output strides (elements)
  unsigned x_bytes, y_bytes, mask_bytes;
  int nb;
  int relu;
  int wmode;           // 0 forward gather, 1 dgrad gather (transposed + flipped)
  unsigned long long* stamps;   // development only (ABL & 4): per-wave s_memtime stamps, 32 per wave
};

#ifdef DRQ_DEV   // development build only (tools/): per-wave time stamps, see drq_dev_* at the end of the file
unsigned long long* g_conv_stamps = nullptr;
int g_conv_variant = -1;          // drq_dev_conv_variant overrides DRQ_CONV_VARIANT
#endif

// BLK = workgroups per CU the kernel is tuned for (8 waves each): 2 -> 4 waves/SIMD (<=128 VGPR)
// ABL (development only): 1 = skip the input loads, 2 = skip the LDS weight reads, 3 = both (timing ablations)
// MASK: the epilogue multiplies by (mask > 0) (dgrad through the ReLU of the layer below)
template <int CIN, int HIN, int STRIDE, int BLK, int TPW, int ABL = 0, bool MASK = false>
__global__ __launch_bounds__(512, 2 * BLK) void conv3x3_kernel(ConvArgs a) {
  constexpr int CP = (CIN + 1) / 2;
  constexpr int NS = CP * 9;                 // MFMA steps per tile
  constexpr int HOUT = (HIN - 3) / STRIDE + 1;
  constexpr int P = HOUT * HOUT;
  constexpr int WP = 73;                     // LDS pitch of one MFMA step (64 lanes + bank-spreading pad)
  __shared__ float wl[NS * WP];              // A operand, [step][lane]
  __shared__ float bl[32];                   // bias

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  unsigned long long* stamp = nullptr;
  int nstamp = 0;
  auto mark = [&]() {
    if constexpr (ABL & 4) {
      if (stamp && nstamp < 30 && lane == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  if constexpr (ABL & 4) {
    if (a.stamps) {
      stamp = a.stamps + ((size_t)blockIdx.x * 8 + wid) * 32;
      if (lane == 0) {
        stamp[30] = __builtin_amdgcn_s_memrealtime();
        stamp[31] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
      }
    }
  }
  mark();
  const int col = lane & 31;   // MFMA: A row (cout) for the weights, B column (pixel) for the input
  const int half = lane >> 5;  // k parity -> channel parity

  // weights -> LDS.  Global reads are contiguous (thread i takes canonical elements i, i+512, ...: a lane-ordered
  // GATHER from global memory touches one 128-byte line per lane and cost several us per workgroup); the
  // scatter into MFMA-lane order happens on the LDS side, where the step pitch WP = 73 (= 9 mod 64) spreads the
  // (tap, channel) neighbours of one wave over distinct banks in both gather modes.
  {
    constexpr int NW = 32 * CIN * 9;
    constexpr int NIT = (NW + 511) / 512;
    float wv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) wv[it] = a.w[min(it * 512 + (int)threadIdx.x, NW - 1)];
    if constexpr (CIN & 1) {   // the missing odd channel of the last pair multiplies real inputs: must be 0
      for (int i = threadIdx.x; i < 9 * 32; i += 512) wl[((CP - 1) * 9 + (i >> 5)) * WP + 32 + (i & 31)] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 512 + (int)threadIdx.x;
      if (idx < NW) {
        const int t = idx % 9, q = idx / 9;
        int kc, row, tt;
        if (a.wmode == 0) { row = q / CIN; kc = q - row * CIN; tt = t; }      // w[row=cout][kc=cin][t]
        else              { kc = q >> 5;   row = q & 31;       tt = 8 - t; }  // w[kc=cout][row=cin][8-t]
        wl[((kc >> 1) * 9 + tt) * WP + (kc & 1) * 32 + row] = wv[it];
      }
    }
  }
  if (threadIdx.x < 32) bl[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
  __syncthreads();
  mark();

  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  // output / mask through buffer descriptors too: one 32-bit per-lane offset + scalar channel offsets
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, 0, a.mask ? a.mask_bytes : 0, 0x00020000);
  const int ycs4 = (int)a.y_cs * 4;
  const int total = a.nb * P;
  const int ntiles = (total + 31) >> 5;
  // contiguous run of tiles per workgroup, waves interleaved inside the run
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_beg = blockIdx.x * per;
  const int t_end = min(ntiles, t_beg + per);
  const float* wlane = wl + lane;

  // per-lane byte offset of a tile's pixel (clamped for the ragged last tile)
  auto tile_voff = [&](int tile) {
    int p = tile * 32 + col;
    if (p >= total) p = total - 1;
    const int b = p / P;
    const int rem = p - b * P;
    const int oy = rem / HOUT;
    const int ox = rem - oy * HOUT;
    return (((b * CIN + half) * HIN + oy * STRIDE) * HIN + ox * STRIDE) * 4;
  };
  // the 9 taps of channel pair c (c may be a runtime value: it only moves the scalar offset)
  auto load_group = [&](float (&dst)[9], int voff, int c) {
    const int cbase = c * (2 * HIN * HIN * 4);
    // one 12-byte load per kernel row: the three kx taps of a lane are adjacent in memory
    // (3x fewer vector-memory instructions than per-tap dword loads; the TA was the limiter)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if constexpr (ABL & 1) {
        dst[ky * 3 + 0] = __uint_as_float(voff + cbase);
        dst[ky * 3 + 1] = __uint_as_float(voff + ky);
        dst[ky * 3 + 2] = __uint_as_float(cbase + ky);
        continue;
      }
      const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, cbase + ky * HIN * 4, 0);
      // (elements are copied to scalars first: __builtin_bit_cast on a vector-element lvalue
      //  reads element 0 for every index with this clang)
      const unsigned e0 = v[0], e1 = v[1], e2 = v[2];
      dst[ky * 3 + 0] = __uint_as_float(e0);
      dst[ky * 3 + 1] = __uint_as_float(e1);
      dst[ky * 3 + 2] = __uint_as_float(e2);
    }
  };
  // TPW pixel tiles per wave iteration: TPW independent accumulator chains share one A (weight) read,
  // so a wave always has an MFMA that does not depend on the one in flight.
  auto mfma_group = [&](f32x16 (&acc)[TPW], const float (&xv)[TPW][9], int c) {
    const float* wp = wlane + c * (9 * WP);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float wv = (ABL & 2) ? __uint_as_float((unsigned)(c * 9 + t) + lane) : wp[t * WP];
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv[j][t], acc[j], 0, 0, 0);
    }
  };
  // ---- epilogue pieces -----------------------------------------------------------------------------------------
  // The waves of a SIMD run in lockstep (same work, round-robin MFMA issue): whatever one wave waits for between
  // its last MFMA of a tile and its first of the next, the MFMA pipe waits for too.  vmcnt retires in order and
  // the compiler can only count exactly in straight-line code, so
  //   * the channel-pair loop of a tile is fully unrolled (every s_waitcnt carries an exact count),
  //   * a finished tile is moved to 16 "out" registers (ReLU / mask applied) and its stores are issued two
  //     stages into the NEXT tile, unconditionally (invalid lanes get an out-of-range offset: the buffer
  //     descriptor drops them) -- a branch would merge two counter states and force a conservative wait,
  //   * the mask loads and the address arithmetic (integer divisions) are issued in mid-tile stages.
  constexpr int kDrop = (int)0x80000000u;   // beyond any num_records (< 2^31): the store is discarded
  auto tile_out = [&](int tile, int& yoff, int& moff) {
    const int p0 = tile * 32 + col;
    const bool ok = tile < t_end && p0 < total;
    const int p = p0 < total ? p0 : total - 1;
    const int b = p / P;
    const int rem = p - b * P;
    const int oy = rem / HOUT;
    const int ox = rem - oy * HOUT;
    // register r holds cout = (r&3) + 8*(r>>2) + 4*half: the 4*half part rides in the lane offset
    yoff = ok ? ((int)a.y_off + b * (int)a.y_bs + oy * (int)a.y_rs + ox) * 4 + half * 4 * ycs4 : kDrop;
    moff = ((b * 32 + 4 * half) * P + rem) * 4;
  };
  auto load_mask = [&](float (&mv)[16], int moff) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrsrc, moff, ((r & 3) + 8 * (r >> 2)) * P * 4, 0));
  };
  auto store_out = [&](const float (&o)[16], int yoff) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), yrsrc, yoff, ((r & 3) + 8 * (r >> 2)) * ycs4, 0);
  };

  // software pipeline over channel pairs ("stages"): the taps of the next stage (or of the next tile's first
  // stage) are in flight while the MFMAs of the current one issue; the other waves of the SIMD fill the rest.
  float X[2][TPW][9];
  float out[TPW][16], mv[TPW][16];
  int voff[TPW], nvoff[TPW], yoff[TPW], moff[TPW], out_yoff[TPW];
  int tile = t_beg + wid * TPW;
  auto clampt = [&](int t) { return t < t_end ? t : t_end - 1; };   // ragged end: harmless re-load
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    out_yoff[j] = kDrop;
    nvoff[j] = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[j][r] = 0.f;
  }
  if (tile < t_end) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      voff[j] = tile_voff(clampt(tile + j));
      load_group(X[0][j], voff[j], 0);
    }
  }
  constexpr int kStoreStage = CP > 4 ? 2 : 0;              // previous tile's stores
  constexpr int kPrepStage = CP > 4 ? CP - 4 : 0;          // next-tile / output addresses (needed at stage CP-1)
  constexpr int kMaskStage = CP > 4 ? CP - 3 : 0;
  for (; tile < t_end; tile += 8 * TPW) {
    // accumulator row (cout) of register r: (r&3) + 8*(r>>2) + 4*half; C-in = bias
    f32x16 acc[TPW];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = bl[(r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j][r] = bv;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      if (c == kPrepStage) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          nvoff[j] = tile_voff(clampt(tile + 8 * TPW + j));
          tile_out(tile + j, yoff[j], moff[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        if (c + 1 < CP) load_group(X[(c + 1) & 1][j], voff[j], c + 1);
        else load_group(X[(c + 1) & 1][j], nvoff[j], 0);
      }
      if (c == kStoreStage) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) store_out(out[j], out_yoff[j]);
      }
      if constexpr (MASK) {
        if (c == kMaskStage) {
#pragma unroll
          for (int j = 0; j < TPW; ++j) load_mask(mv[j], moff[j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // memory instructions stay ahead of this stage's MFMAs
      mfma_group(acc, X[c & 1], c);
    }
    if constexpr (CP & 1) {                // odd stage count: the next tile's first taps landed in X[1]
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) X[0][j][t] = X[1][j][t];
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        if constexpr (MASK) v = mv[j][r] > 0.f ? v : 0.f;
        out[j][r] = v;
      }
      out_yoff[j] = yoff[j];
      voff[j] = nvoff[j];
    }
    if constexpr (ABL & 8) mark();
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) store_out(out[j], out_yoff[j]);   // the last tile of this wave
  if constexpr (ABL & 4) {
    __builtin_amdgcn_s_waitcnt(0);
    mark();
    if (stamp && lane == 0) stamp[29] = __builtin_amdgcn_s_memrealtime();
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;     // layer input  [NB][CIN][HIN][HIN]
  const float* dy;    // grad of the pre-activation, [NB][32][dy rows][dy cols] addressed with strides
  long dy_bs, dy_cs, dy_rs, dy_off;
  float* part;        // [nblocks][PART]
  unsigned x_bytes, dy_bytes;
  int nb;
};

constexpr int even_odd_half(int n) {   // smallest even p >= n with p/2 odd  (LDS row pitch, see below)
  int p = (n + 1) & ~1;
  return (p / 2) % 2 ? p : p + 2;
}

template <int CIN, int HIN, int STRIDE>
struct WgradGeom {
  static constexpr int HOUT = (HIN - 3) / STRIDE + 1;
  static constexpr int KS = (HOUT + 1) / 2;                 // k-steps (pixel pairs) per output row
  static constexpr int XW = (2 * KS - 1) * STRIDE + 3;      // input columns touched (>= HIN)
  // Rows are staged with 8-byte loads/LDS writes (3x fewer instructions than dwords), so row starts must be
  // 8-byte aligned: even pitch.  pitch/2 odd keeps the channel-strided MFMA operand reads at 2-way conflicts.
  static constexpr int XPAIRS = (HIN + 1) / 2;              // 8-byte pieces per input row
  static constexpr int DPAIRS = (HOUT + 1) / 2;
  static constexpr int XP = even_odd_half(XW > 2 * XPAIRS ? XW : 2 * XPAIRS);
  static constexpr int DP = even_odd_half(2 * KS + 1 > 2 * DPAIRS ? 2 * KS + 1 : 2 * DPAIRS);
  static constexpr int XRPI = 64 / XPAIRS;                  // input rows per load instruction
  static constexpr int DRPI = 64 / DPAIRS;
  static constexpr int XJ = (CIN + XRPI - 1) / XRPI;        // load instructions per kernel row ky
  static constexpr int DJ = (32 + DRPI - 1) / DRPI;
  static constexpr bool SMALL = (CIN * 3 <= 32);            // conv1: columns = (ci,kx), tiles = ky
  static constexpr int NT = SMALL ? 3 : 9;
  static constexpr int XS = 3 * CIN * XP;                   // floats of X per wave
  static constexpr int WAVE_LDS = XS + 32 * DP;             // floats per wave
  static constexpr int PART = NT * 1024 + 64;               // floats per partial record
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifdef DRQ_DEV
unsigned long long* g_wgrad1_stamps = nullptr;
#else
constexpr unsigned long long* g_wgrad1_stamps = nullptr;
#endif

template <int CIN, int HIN, int STRIDE, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_kernel(WgradArgs a, unsigned long long* stamps = nullptr) {
  using G = WgradGeom<CIN, HIN, STRIDE>;
  constexpr int HOUT = G::HOUT, KS = G::KS, XP = G::XP, DP = G::DP, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row bookkeeping stays scalar
  const int col = lane & 31;
  const int half = lane >> 5;
  float* xs = smem + wid * G::WAVE_LDS;
  float* ds = xs + G::XS;
  unsigned long long* stamp = nullptr;            // development only: 32 per wave; [30], [29] = wall clock start / end
  int nstamp = 0;
  if (STAMP) {
    stamp = stamps + ((size_t)blockIdx.x * 4 + wid) * 32;
    if (lane == 0) stamp[30] = __builtin_amdgcn_s_memrealtime();
  }
  auto mark = [&]() {
    if (STAMP) {
      if (nstamp < 28 && lane == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  mark();

  // zero the wave-private tile once: pad columns must hold finite values (they meet dY == 0)
  if constexpr (G::WAVE_LDS % 4 == 0) {
    float4* x4 = reinterpret_cast<float4*>(xs);
#pragma unroll 4
    for (int i = lane; i < G::WAVE_LDS / 4; i += 64) x4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int i = lane; i < G::WAVE_LDS; i += 64) xs[i] = 0.f;
  }

  // this lane's operand base addresses (floats)
  int bbase;   // B operand: X[pixel][column]
  if (G::SMALL) {
    const int ci = col < CIN * 3 ? col / 3 : 0;
    const int kx = col < CIN * 3 ? col % 3 : 0;
    bbase = ci * XP + kx + half * STRIDE;
  } else {
    bbase = (col < CIN ? col : 0) * XP + half * STRIDE;
  }
  const int abase = col * DP + half;   // A operand: dY[cout][pixel]

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // (sample, output row) units split evenly over the waves, the first `rem` waves take one more (all scalar)
  const int units = a.nb * HOUT;
  const int nw = (int)gridDim.x * 4;
  const int gw = (int)blockIdx.x * 4 + wid;
  const int per = units / nw, rem = units - per * nw;
  const int u0 = gw * per + (gw < rem ? gw : rem);
  const int u1 = u0 + per + (gw < rem ? 1 : 0);

  // ---- register staging with 8-byte pieces: lane = (row within the instruction, pair within the row)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const int xr = lane / G::XPAIRS, xpr = lane - xr * G::XPAIRS;      // xr >= XRPI: idle lane
  const int dr = lane / G::DPAIRS, dpr = lane - dr * G::DPAIRS;
  const bool xact = xr < G::XRPI, dact = dr < G::DRPI;
  const int xg_lane = (xr * HIN * HIN + 2 * xpr) * 4;                // global byte offsets (lane part)
  const int dg_lane = (int)(dr * a.dy_cs + 2 * dpr) * 4;
  const int xl_lane = xr * XP + 2 * xpr;                             // LDS float offsets (lane part)
  const int dl_lane = dr * DP + 2 * dpr;
  u32x2 rx[3 * G::XJ], rd[G::DJ];
  // per-lane offsets are loop invariants (a disabled lane sits far out of the buffer's range and stays there when
  // the wave-uniform row offset is added: 0x7ffffff0 + offset < 2^32); only the last piece of a row group is partial
  const int xv_full = xact ? xg_lane : 0x7ffffff0;
  const int xv_last = xact && (G::XJ - 1) * G::XRPI + xr < CIN ? xg_lane : 0x7ffffff0;
  const int dv_full = dact ? dg_lane : 0x7ffffff0;
  const int dv_last = dact && (G::DJ - 1) * G::DRPI + dr < 32 ? dg_lane : 0x7ffffff0;
  auto issue_loads = [&](int u) {
    const int b = u / HOUT;
    const int oy = u - b * HOUT;
    const int xbase = ((b * CIN * HIN + oy * STRIDE) * HIN) * 4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int j = 0; j < G::XJ; ++j) {
        // rows (ky, ci = j*XRPI + xr)
        const int xv = j == G::XJ - 1 ? xv_last : xv_full;
        const int row_off = (j * G::XRPI * HIN * HIN + ky * HIN) * 4;
        if constexpr (HIN % 2 == 0) {
          // even rows: no 8-byte piece reaches past its row, so the uniform part may ride in the scalar offset
          // (which the range check does not see) and the load needs no VALU instruction at all
          rx[ky * G::XJ + j] = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, xv, xbase + row_off, 0);
        } else {
          rx[ky * G::XJ + j] = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, xv + xbase, row_off, 0);
        }
      }
    const int dbase = (int)(a.dy_off + (long)b * a.dy_bs + (long)oy * a.dy_rs) * 4;
#pragma unroll
    for (int j = 0; j < G::DJ; ++j) {
      const int dv = j == G::DJ - 1 ? dv_last : dv_full;
      rd[j] = __builtin_amdgcn_raw_buffer_load_b64(drsrc, dv + (dbase + (int)(j * G::DRPI * a.dy_cs) * 4), 0, 0);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int j = 0; j < G::XJ; ++j)
        if (xact && j * G::XRPI + xr < CIN)
          *reinterpret_cast<u32x2*>(xs + (ky * CIN + j * G::XRPI) * XP + xl_lane) = rx[ky * G::XJ + j];
#pragma unroll
    for (int j = 0; j < G::DJ; ++j)
      if (dact && j * G::DRPI + dr < 32)
        *reinterpret_cast<u32x2*>(ds + j * G::DRPI * DP + dl_lane) = rd[j];
  };

  if (u0 < u1) issue_loads(u0);
  mark();
  for (int u = u0; u < u1; ++u) {
    mark();
    // (single wave: LDS operations of one wave complete in order, no barrier needed)
    write_lds();
    if (STAMP) {                          // odd stamps: the staged row is in LDS (loads waited for, stores issued)
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      mark();
    }
    if (u + 1 < u1) issue_loads(u + 1);   // in flight under the MFMA loop below
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float av = ds[abase + 2 * s];
      // an odd row's last pixel pair is (HOUT-1, pad): the pad's dY (the next row's first value in LDS) is zero
      if ((HOUT & 1) && s == KS - 1) av = half ? 0.f : av;
      bsum += av;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int off = G::SMALL ? (t * CIN * XP + 2 * s * STRIDE)
                                 : ((t / 3) * CIN * XP + (t % 3) + 2 * s * STRIDE);
        const float bv = xs[bbase + off];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- reduce the 4 waves of the block through LDS, one partial record per block
  mark();
  __syncthreads();
  mark();
  float* red = smem;   // [4][PART]  (fits: checked on the host)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid * G::PART + t * 1024 + r * 64 + lane] = acc[t][r];
  red[wid * G::PART + NT * 1024 + lane] = bsum;
  __syncthreads();
  float* out = a.part + (long)blockIdx.x * G::PART;
  for (int i = threadIdx.x; i < G::PART; i += 256)
    out[i] = (red[i] + red[G::PART + i]) + (red[2 * G::PART + i] + red[3 * G::PART + i]);
  mark();
  if (STAMP && lane == 0) {
    stamp[29] = __builtin_amdgcn_s_memrealtime();
    stamp[28] = (unsigned long long)nstamp;
  }
}

// ---- wgrad v2: the four waves of a workgroup take four consecutive output rows of one sample and SHARE
// the staged input rows (3+3*STRIDE rows instead of 12); the LDS footprint (<= 74 KB incl. the final
// reduction) lets two workgroups live on a CU, so one workgroup's staging/barriers hide under the other's
// MFMA loop.
template <int CIN, int HIN, int STRIDE>
struct Wgrad2Geom : WgradGeom<CIN, HIN, STRIDE> {
  using G = WgradGeom<CIN, HIN, STRIDE>;
  static constexpr int NR = 3 * STRIDE + 3;                    // input rows per group of 4 output rows
  static constexpr int NG = (G::HOUT + 3) / 4;                 // groups per sample
  static constexpr int XQ = (NR * G::XJ + 3) / 4;              // X load instructions per wave and group
  static constexpr int XS2 = NR * CIN * G::XP;                 // floats of the shared input tile
  static constexpr int TILE = XS2 + 4 * 32 * G::DP;            // + one dY row per wave
  static constexpr int LDS_FLOATS = TILE > 2 * G::PART ? TILE : 2 * G::PART;
};

template <int CIN, int HIN, int STRIDE>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad2_kernel(WgradArgs a) {
  using G = Wgrad2Geom<CIN, HIN, STRIDE>;
  constexpr int HOUT = G::HOUT, KS = G::KS, XP = G::XP, DP = G::DP, NT = G::NT;
  constexpr int XJ = G::XJ, DJ = G::DJ, XRPI = G::XRPI, DRPI = G::DRPI, NR = G::NR, XQ = G::XQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int col = lane & 31;
  const int half = lane >> 5;
  float* xs = smem;
  float* ds = smem + G::XS2 + wid * 32 * DP;

  // pad columns must hold finite values (they meet dY == 0): zero the tile once
  for (int i = threadIdx.x; i < G::TILE; i += 256) smem[i] = 0.f;

  int bbase;   // B operand: X[pixel][column], rows of this wave start at input row wid*STRIDE of the tile
  if (G::SMALL) {
    const int ci = col < CIN * 3 ? col / 3 : 0;
    const int kx = col < CIN * 3 ? col % 3 : 0;
    bbase = (wid * STRIDE * CIN + ci) * XP + kx + half * STRIDE;
  } else {
    bbase = (wid * STRIDE * CIN + (col < CIN ? col : 0)) * XP + half * STRIDE;
  }
  const int abase = col * DP + half;   // A operand: dY[cout][pixel]

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int ngroups = a.nb * G::NG;
  const int g0 = (int)((long)ngroups * blockIdx.x / gridDim.x);
  const int g1 = (int)((long)ngroups * (blockIdx.x + 1) / gridDim.x);

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const int xr = lane / G::XPAIRS, xpr = lane - xr * G::XPAIRS;      // xr >= XRPI: idle lane
  const int dr = lane / G::DPAIRS, dpr = lane - dr * G::DPAIRS;
  const bool xact = xr < XRPI, dact = dr < DRPI;
  const int xg_lane = (xr * HIN * HIN + 2 * xpr) * 4;                // global byte offsets (lane part)
  const int dg_lane = (int)(dr * a.dy_cs + 2 * dpr) * 4;
  const int xl_lane = xr * XP + 2 * xpr;                             // LDS float offsets (lane part)
  const int dl_lane = dr * DP + 2 * dpr;
  const bool d_last_odd = (HOUT & 1) && dpr == G::DPAIRS - 1;

  u32x2 rx[XQ], rd[DJ];
  // instruction q of this wave covers tile row (q*4+wid) / XJ and channel block (q*4+wid) % XJ
  auto issue_loads = [&](int g) {
    const int b = g / G::NG;
    const int oy0 = (g - b * G::NG) * 4;
    const int xbase = ((b * CIN * HIN + oy0 * STRIDE) * HIN) * 4;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int it = q * 4 + wid;
      const int trow = it / XJ, j = it - trow * XJ;
      const bool ok = xact && it < NR * XJ && j * XRPI + xr < CIN && oy0 * STRIDE + trow < HIN;
      const int voff = ok ? xbase + xg_lane + (j * XRPI * HIN * HIN + trow * HIN) * 4 : 0x7ffffff0;
      rx[q] = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, 0, 0);
    }
    const int oy = oy0 + wid;
    const int dbase = (int)(a.dy_off + (long)b * a.dy_bs + (long)oy * a.dy_rs) * 4;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const bool ok = dact && j * DRPI + dr < 32 && oy < HOUT;
      const int voff = ok ? dbase + dg_lane + (int)(j * DRPI * a.dy_cs) * 4 : 0x7ffffff0;
      rd[j] = __builtin_amdgcn_raw_buffer_load_b64(drsrc, voff, 0, 0);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int it = q * 4 + wid;
      const int trow = it / XJ, j = it - trow * XJ;
      if (xact && it < NR * XJ && j * XRPI + xr < CIN)
        *reinterpret_cast<u32x2*>(xs + (trow * CIN + j * XRPI) * XP + xl_lane) = rx[q];
    }
#pragma unroll
    for (int j = 0; j < DJ; ++j)
      if (dact && j * DRPI + dr < 32) {
        u32x2 v = rd[j];
        if (d_last_odd) v[1] = 0u;      // column HOUT of an odd row must stay zero (it pairs with the pad pixel)
        *reinterpret_cast<u32x2*>(ds + j * DRPI * DP + dl_lane) = v;
      }
  };

  __syncthreads();                       // tile zeroed
  for (int g = g0; g < g1; ++g) {
    // no cross-group register prefetch (144 accumulators leave no room at 2 waves/SIMD): the load latency
    // of this workgroup is covered by the MFMA loop of the other workgroup on the CU
    issue_loads(g);
    write_lds();
    __syncthreads();                     // tile of group g complete
    const int oy = (g % G::NG) * 4 + wid;
    if (oy < HOUT) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float av = ds[abase + 2 * s];
        bsum += av;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int off = G::SMALL ? (t * CIN * XP + 2 * s * STRIDE)
                                   : ((t / 3) * CIN * XP + (t % 3) + 2 * s * STRIDE);
          const float bv = xs[bbase + off];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();                     // every wave done reading the tile
  }

  // ---- 4 waves -> 1 record, two rounds through LDS (2 records = 74 KB), fixed order
  float* red = smem;
  auto put = [&](int slot) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[slot * G::PART + t * 1024 + r * 64 + lane] = acc[t][r];
    red[slot * G::PART + NT * 1024 + lane] = bsum;
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[slot * G::PART + t * 1024 + r * 64 + lane];
    bsum += red[slot * G::PART + NT * 1024 + lane];
  };
  if (wid >= 2) put(wid - 2);
  __syncthreads();
  if (wid < 2) add(wid);                 // wave0 += wave2, wave1 += wave3
  __syncthreads();
  if (wid == 1) put(0);
  __syncthreads();
  if (wid == 0) {
    add(0);
    float* out = a.part + (long)blockIdx.x * G::PART;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[t * 1024 + r * 64 + lane] = acc[t][r];
    out[NT * 1024 + lane] = bsum;
  }
}

// ---- wgrad v3 (stride 1, 32 input channels): wave-private ROLLING tile.  Consecutive output rows of a sample
// share two of their three input rows, so the wave keeps a ring of 4 input rows and a double-buffered dY row in
// LDS and fetches only ONE new input row + ONE dY row per output row (22 instead of 44 load / LDS-write
// instructions).  Those are spread over the k-steps of the current row's MFMA loop (loads early, LDS writes
// 7 steps later, into the ring slot / dY buffer the loop does not read), so staging hides under the MFMAs
// although the wave is alone on its SIMD.  A disabled load gets a zero-length buffer descriptor (returns 0)
// instead of being branched around (branches would force vmcnt(0) waits).
#ifdef DRQ_DEV
unsigned long long* g_wgrad_stamps = nullptr;
#else
constexpr unsigned long long* g_wgrad_stamps = nullptr;
#endif

// The k-step loop is written for a wave that is alone on its SIMD.  Measured there (tools/mfma_issue_probe.hip):
// every VALU instruction between the MFMAs costs ~5 cycles of matrix-pipe time (LDS and scalar instructions cost
// ~1), so the loop keeps the vector ALU out of the staging path altogether: all per-row / per-piece address
// arithmetic is wave-uniform and lives in SGPRs (each staging load gets its own buffer descriptor: base and byte
// count advanced by scalar adds, which also keeps the hardware range check exact), the per-lane parts of the
// global and LDS addresses are loop invariants, and lanes that have nothing to load mirror a neighbour (same
// address, same value, same LDS destination) instead of being masked.
template <int HIN, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad3_kernel(WgradArgs a, unsigned long long* stamps = nullptr) {
  constexpr int CIN = 32;
  using G = WgradGeom<CIN, HIN, 1>;
  constexpr int HOUT = G::HOUT, KS = G::KS, XP = G::XP, DP = G::DP, NT = 9;
  constexpr int XJ = G::XJ, DJ = G::DJ, XRPI = G::XRPI, DRPI = G::DRPI;
  constexpr int RS = CIN * XP;            // floats per ring slot (one input row, all channels)
  constexpr int DS = 32 * DP;             // floats per dY buffer
  constexpr int WAVE = 4 * RS + 2 * DS + 128;
  static_assert(KS >= 18, "staging schedule needs 18 k-steps");
  static_assert(WAVE % 4 == 0, "16-byte zero fill");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31;
  const int half = lane >> 5;
  float* ring = smem + wid * WAVE;
  float* dyb = ring + 4 * RS;
  unsigned long long* stamp = nullptr;            // development only: 32 per wave; [30], [29] = wall clock start / end
  int nstamp = 0;
  if (STAMP) {
    stamp = stamps + ((size_t)blockIdx.x * 4 + wid) * 32;
    if (lane == 0) stamp[30] = __builtin_amdgcn_s_memrealtime();
  }
  auto mark = [&]() {
    if (STAMP) {
      if (nstamp < 28 && lane == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  mark();
  auto zero_fill = [&]() {                        // pad columns stay finite / zero
    float4* r4 = reinterpret_cast<float4*>(ring);
#pragma unroll 4
    for (int i = lane; i < WAVE / 4; i += 64) r4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  };

  const int bbase = col * XP + half;      // B operand: X[pixel][ci]
  const int abase = col * DP + half;      // A operand: dY[cout][pixel]

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // rows of this wave: units split evenly, the first `rem` waves take one more (all scalar)
  const int units = a.nb * HOUT;
  const int nw = (int)gridDim.x * 4;
  const int gw = (int)blockIdx.x * 4 + wid;
  const int per = units / nw, rem = units - per * nw;
  const int u0 = gw * per + (gw < rem ? gw : rem);
  const int u1 = u0 + per + (gw < rem ? 1 : 0);

  // lane -> (row of the piece, 8-byte pair of the row); lanes past the piece and, in the last piece, rows past
  // channel 31 mirror the last valid lane
  constexpr int XLASTR = CIN - 1 - (XJ - 1) * XRPI, DLASTR = 31 - (DJ - 1) * DRPI;   // last valid row of the last piece
  const int xr0 = lane / G::XPAIRS, dr0 = lane / G::DPAIRS;
  const int xr = xr0 < XRPI ? xr0 : XRPI - 1, xpr = xr0 < XRPI ? lane - xr0 * G::XPAIRS : G::XPAIRS - 1;
  const int dr = dr0 < DRPI ? dr0 : DRPI - 1, dpr = dr0 < DRPI ? lane - dr0 * G::DPAIRS : G::DPAIRS - 1;
  const int xrl = xr < XLASTR ? xr : XLASTR, drl = dr < DLASTR ? dr : DLASTR;
  const int xg_full = (xr * HIN * HIN + 2 * xpr) * 4, xg_last = (xrl * HIN * HIN + 2 * xpr) * 4;   // global, bytes
  const int dg_full = (int)(dr * a.dy_cs + 2 * dpr) * 4, dg_last = (int)(drl * a.dy_cs + 2 * dpr) * 4;
  const int xl_full = xr * XP + 2 * xpr, xl_last = xrl * XP + 2 * xpr;                            // LDS, floats
  const int dl_full = dr * DP + 2 * dpr, dl_last = drl * DP + 2 * dpr;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const unsigned long xaddr = (unsigned long)a.x, daddr = (unsigned long)a.dy;
  const unsigned x_bytes = a.x_bytes, dy_bytes = a.dy_bytes;
  constexpr unsigned XSTEP = XRPI * HIN * HIN * 4;          // bytes between the pieces of an input row
  const unsigned d_step = (unsigned)a.dy_cs * 4u * DRPI;
  const unsigned d_bs4 = (unsigned)a.dy_bs * 4u, d_rs4 = (unsigned)a.dy_rs * 4u, d_off4 = (unsigned)a.dy_off * 4u;

  // Hand-placed staging loads (inline asm, invisible to hipcc's scheduler and waitcnt pass, which otherwise sinks
  // every load next to its LDS store and waits for it at once).  Their completion is counted by hand below:
  // loads and only loads are outstanding inside the k-step loop, vmcnt retires them in issue order.
  // `off` = byte offset of the piece (wave-uniform); a disabled load gets a zero-length buffer and returns 0.
  auto ld_asm = [&](u32x2& dst, unsigned long base, unsigned bytes, unsigned off, int lane_off, bool on) {
    const unsigned long p = base + off;
    const u32x4v srd = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, on && off < bytes ? bytes - off : 0u, 0x00020000u};
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(lane_off), "s"(srd) : "memory");
  };
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  // (re)start of a sample: three input rows into ring slots 0..2 (not overlapped; once per sample and wave start)
  auto prime_x = [&](int b, int oy, bool first) {
    u32x2 t[3][XJ];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        t[k][j] = __builtin_amdgcn_raw_buffer_load_b64(
            xrsrc, (j == XJ - 1 ? xg_last : xg_full) + j * (int)XSTEP, ((b * CIN * HIN + oy + k) * HIN) * 4, 0);
    if (first) {                          // the zero fill of the tile runs while the first loads are in flight
      __builtin_amdgcn_sched_barrier(0);
      zero_fill();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        *reinterpret_cast<u32x2*>(ring + k * RS + j * XRPI * XP + (j == XJ - 1 ? xl_last : xl_full)) = t[k][j];
  };

  int slot = 0, dbuf = 0;
  if (u0 < u1) {
    const int b = u0 / HOUT, oy = u0 - b * HOUT;
    u32x2 t[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j)
      t[j] = __builtin_amdgcn_raw_buffer_load_b64(drsrc, (j == DJ - 1 ? dg_last : dg_full) + (int)(j * d_step),
                                                  (int)(d_off4 + b * d_bs4 + oy * d_rs4), 0);
    prime_x(b, oy, true);
#pragma unroll
    for (int j = 0; j < DJ; ++j)
      *reinterpret_cast<u32x2*>(dyb + j * DRPI * DP + (j == DJ - 1 ? dl_last : dl_full)) = t[j];
  } else {
    zero_fill();
  }
  mark();
  for (int u = u0; u < u1; ++u) {
    mark();
    const int b = u / HOUT, oy = u - b * HOUT;
    const bool has_next = u + 1 < u1;
    const bool same = has_next && oy + 1 < HOUT;          // next row belongs to the same sample
    const int nb = same ? b : b + 1, noy = same ? oy + 1 : 0;
    const int wslot = (slot + 3) & 3;
    const int rb0 = bbase + ((slot + 0) & 3) * RS, rb1 = bbase + ((slot + 1) & 3) * RS,
              rb2 = bbase + ((slot + 2) & 3) * RS;
    const float* da = dyb + dbuf * DS + abase;
    const unsigned xrow = (unsigned)((b * CIN * HIN + oy + 3) * HIN) * 4u;     // staged input row (same sample only)
    const unsigned drow = d_off4 + nb * d_bs4 + noy * d_rs4;                   // staged dY row
    float* px_full = ring + wslot * RS + xl_full;
    float* px_last = ring + wslot * RS + xl_last;
    float* pd_full = dyb + (dbuf ^ 1) * DS + dl_full;
    float* pd_last = dyb + (dbuf ^ 1) * DS + dl_last;
    u32x2 sx[XJ], sd[DJ];
    // MFMA operands are read from LDS one k-step ahead (register double buffer), in three groups placed between
    // the MFMAs of the current step; the staging instructions sit between later MFMAs.  sched_barrier pins that
    // order (left alone, hipcc issues the reads of a step right before their first use).
    auto rd_row = [&](int ky, int s, float* bv) {
      const int rb = ky == 0 ? rb0 : (ky == 1 ? rb1 : rb2);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) bv[ky * 3 + kx] = ring[rb + kx + 2 * s];
    };
    float cav = da[0], cbv[NT];
    rd_row(0, 0, cbv);
    rd_row(1, 0, cbv);
    rd_row(2, 0, cbv);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      static_assert(XJ == DJ && XJ == 11, "hand-counted schedule assumes 11 pieces per row");
      float nav = 0.f, nbv[NT];
      const bool more = s + 1 < KS;
      bsum += cav;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[0], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        nav = da[2 * (s + 1)];
        // an odd row's last pixel pair is (HOUT-1, pad): the pad's dY (the next row's first value in LDS) is zero
        if ((HOUT & 1) && s + 1 == KS - 1) nav = half ? 0.f : nav;
        rd_row(0, s + 1, nbv);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[1], acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) rd_row(1, s + 1, nbv);
      __builtin_amdgcn_sched_barrier(0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[2], acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) rd_row(2, s + 1, nbv);
      __builtin_amdgcn_sched_barrier(0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[3], acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- staging for the next unit: piece s of the input row and of the dY row is loaded in step s (s < 11)
      // and written to LDS in step s+7; N = loads issued after the pair being retired
      if (s < XJ) {
        ld_asm(sx[s], xaddr, x_bytes, xrow + s * XSTEP, s == XJ - 1 ? xg_last : xg_full, same);
        ld_asm(sd[s], daddr, dy_bytes, drow + s * d_step, s == DJ - 1 ? dg_last : dg_full, has_next);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[4], acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[5], acc[5], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s >= 7 && s - 7 < XJ) {
        constexpr int LAST = XJ - 1;                       // last step that issues loads
        const int n_after = 2 * ((s < LAST ? s : LAST) - (s - 7));
        switch (n_after) {                                 // s is a compile-time constant after unrolling
          case 14: asm volatile("s_waitcnt vmcnt(14)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 12: asm volatile("s_waitcnt vmcnt(12)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 10: asm volatile("s_waitcnt vmcnt(10)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
          default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(sx[s - 7]), "+v"(sd[s - 7])); break;
        }
        // a disabled load returned zeros: storing them is harmless (the ring slot / dY buffer written here is
        // re-primed or rewritten before it is read again)
        const int j = s - 7;
        *reinterpret_cast<u32x2*>((j == XJ - 1 ? px_last : px_full) + j * XRPI * XP) = sx[j];
        *reinterpret_cast<u32x2*>((j == DJ - 1 ? pd_last : pd_full) + j * DRPI * DP) = sd[j];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[6], acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[7], acc[7], 0, 0, 0);
      acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(cav, cbv[8], acc[8], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      cav = nav;
#pragma unroll
      for (int t = 0; t < NT; ++t) cbv[t] = nbv[t];
    }
    if (has_next && !same) {
      prime_x(nb, 0, false);
      slot = 0;
    } else {
      slot = (slot + 1) & 3;
    }
    dbuf ^= 1;
  }

  // ---- reduce the 4 waves of the block through LDS, one partial record per block
  mark();
  __syncthreads();
  mark();
  float* red = smem;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid * G::PART + t * 1024 + r * 64 + lane] = acc[t][r];
  red[wid * G::PART + NT * 1024 + lane] = bsum;
  __syncthreads();
  static_assert(G::PART % 4 == 0, "16-byte record reduction");
  float4* out = reinterpret_cast<float4*>(a.part + (long)blockIdx.x * G::PART);
  const float4* r4 = reinterpret_cast<const float4*>(red);
  for (int i = threadIdx.x; i < G::PART / 4; i += 256) {
    const float4 p = r4[i], q = r4[G::PART / 4 + i], v = r4[2 * (G::PART / 4) + i], w = r4[3 * (G::PART / 4) + i];
    out[i] = make_float4((p.x + q.x) + (v.x + w.x), (p.y + q.y) + (v.y + w.y), (p.z + q.z) + (v.z + w.z),
                         (p.w + q.w) + (v.w + w.w));
  }
  mark();
  if (STAMP && lane == 0) {
    stamp[29] = __builtin_amdgcn_s_memrealtime();
    stamp[28] = (unsigned long long)nstamp;
  }
}

// Sums the per-block partial records in a fixed order and scatters to the canonical layouts.
// Block = 64 record elements x 16 groups of partials (1024 threads).
template <int CIN, bool SMALL>
__global__ __launch_bounds__(1024) void conv3x3_wgrad_reduce_kernel(const float* part, int nblocks, float* dw,
                                                                    float* db) {
  constexpr int NT = SMALL ? 3 : 9;
  constexpr int PART = NT * 1024 + 64;
  __shared__ float sm[16][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;        // PART is a multiple of 64
  float s = 0.f;
  for (int k = grp; k < nblocks; k += 16) s += part[(long)k * PART + i];
  sm[grp][e] = s;
  __syncthreads();
  if (grp != 0) return;
  float tot = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) tot += sm[g][e];
  if (i < NT * 1024) {
    const int t = i >> 10, r = (i >> 6) & 15, lane = i & 63;
    const int col = lane & 31, half = lane >> 5;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
    if (SMALL) {
      if (col < CIN * 3) dw[(co * CIN + col / 3) * 9 + t * 3 + col % 3] = tot;
    } else {
      if (col < CIN) dw[(co * CIN + col) * 9 + t] = tot;
    }
  } else {
    // bias partials: lane (cout, half); the two halves hold even / odd pixels
    const float other = __shfl_xor(tot, 32);
    if (e < 32) db[e] = tot + other;
  }
}

// the same reduction for up to 4 layers in one launch (blockIdx.y = layer): the encoder backward defers the four
// per-layer reductions to its end (nothing reads dW/db before the optimiser step)
struct ReduceJob {
  const float* part;
  float* dw;
  float* db;
  int nblocks, cin, small;
};
struct ReduceJobs {
  ReduceJob j[4];
};

__global__ __launch_bounds__(1024) void conv3x3_wgrad_reduce_multi_kernel(ReduceJobs js) {
  const ReduceJob J = js.j[blockIdx.y];
  const int NT = J.small ? 3 : 9;
  const int PART = NT * 1024 + 64;
  __shared__ float sm[16][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  if (blockIdx.x * 64 >= PART) return;      // uniform per workgroup
  float s = 0.f;
  for (int k0 = grp; k0 < J.nblocks; k0 += 16 * 16) {     // 16 records in flight per thread, added in order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = J.part[(long)min(k0 + 16 * u, J.nblocks - 1) * PART + i];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (k0 + 16 * u < J.nblocks) ? v[u] : 0.f;
  }
  sm[grp][e] = s;
  __syncthreads();
  if (grp != 0) return;
  float tot = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) tot += sm[g][e];
  if (i < NT * 1024) {
    const int t = i >> 10, r = (i >> 6) & 15, lane = i & 63;
    const int col = lane & 31, half = lane >> 5;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
    if (J.small) {
      if (col < J.cin * 3) J.dw[(co * J.cin + col / 3) * 9 + t * 3 + col % 3] = tot;
    } else {
      if (col < J.cin) J.dw[(co * J.cin + col) * 9 + t] = tot;
    }
  } else {
    const float other = __shfl_xor(tot, 32);
    if (e < 32) J.db[e] = tot + other;
  }
}

template <int CIN, int HIN, int STRIDE, int BLK, int TPW, int ABL = 0, bool MASK = false>
int launch_conv_v(const ConvArgs& a, hipStream_t st) {
  constexpr int HOUT = (HIN - 3) / STRIDE + 1;
  const long ntiles = ((long)a.nb * HOUT * HOUT + 31) / 32;
  long blocks = (ntiles + 8 * TPW - 1) / (8 * TPW);
  const long cap = (long)BLK * drq_num_cus();
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  // Pin the residency to exactly BLK workgroups per CU: registers and the static LDS alone would admit BLK+1, and
  // the dispatcher then packs BLK+1 on some CUs and leaves others short (a one-wave grid of equal-length
  // workgroups then runs at the pace of the over-filled CUs: measured -12 % on an MFMA-only probe).  Unused
  // dynamic LDS raises the group segment just past 160 KB / (BLK+1).
  constexpr int NS = ((CIN + 1) / 2) * 9;
  constexpr int static_lds = NS * 73 * 4 + 32 * 4;
  constexpr int want = 160 * 1024 / (BLK + 1) + 1024;
  const int pad = want > static_lds ? ((want - static_lds + 255) & ~255) : 0;
  hipLaunchKernelGGL((conv3x3_kernel<CIN, HIN, STRIDE, BLK, TPW, ABL, MASK>), dim3((unsigned)blocks), dim3(512), pad, st,
                     a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

template <int CIN, int HIN, int STRIDE>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  if (a.mask) {
    if constexpr (CIN == 32) return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 0, true>(a, st);
    else return DRQ_EARG;
  }
#ifdef DRQ_DEV
  // DRQ_CONV_VARIANT / drq_dev_conv_variant: timing ablations of the forward kernel.  Development build only
  // (tools/conv_ab.py, tools/conv_stamps.py): variants 6-8 skip loads and compute garbage on purpose.
  if (g_conv_variant < 0) {
    const char* e = getenv("DRQ_CONV_VARIANT");
    g_conv_variant = e ? atoi(e) : 0;
  }
  switch (g_conv_variant) {
    case 1: return launch_conv_v<CIN, HIN, STRIDE, 2, 2>(a, st);        // 2 tiles per wave share each weight read
    case 2: return launch_conv_v<CIN, HIN, STRIDE, 1, 2>(a, st);
    case 4: return launch_conv_v<CIN, HIN, STRIDE, 1, 4>(a, st);        // 1 workgroup/CU, 4 accumulator chains/wave
    case 6: return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 1>(a, st);     // no input loads
    case 7: return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 2>(a, st);     // no LDS weight reads
    case 8: return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 3>(a, st);     // neither: MFMA + epilogue only
    case 10: case 11: case 12: {                                        // + per-wave time stamps
      ConvArgs b = a;
      b.stamps = g_conv_stamps;
      if (g_conv_variant == 10) return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 4>(b, st);        // start/end only
      if (g_conv_variant == 11) return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 12>(b, st);       // + every tile
      return launch_conv_v<CIN, HIN, STRIDE, 2, 1, 7>(b, st);                                  // MFMA only, start/end
    }
    default: break;
  }
#endif
  return launch_conv_v<CIN, HIN, STRIDE, 2, 1>(a, st);
}

// 1 = rolling-tile kernel (conv2..4), the product path.  The development build can select 2 (block-shared input
// rows, measured slower) through DRQ_WGRAD_VARIANT.
inline int wgrad_variant() {
#ifdef DRQ_DEV
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DRQ_WGRAD_VARIANT");
    v = e ? atoi(e) : 1;
  }
  return v;
#else
  return 1;
#endif
}

// defer: run only the partial-sum kernel (records in ws, their count in *nblocks_out); the caller reduces several
// layers with one launch of conv3x3_wgrad_reduce_multi_kernel
template <int CIN, int HIN, int STRIDE>
int launch_wgrad(const WgradArgs& a0, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t st,
                 bool defer = false, int* nblocks_out = nullptr) {
  using G = WgradGeom<CIN, HIN, STRIDE>;
  using G2 = Wgrad2Geom<CIN, HIN, STRIDE>;
  static_assert(4 * G::WAVE_LDS * 4 <= 160 * 1024, "LDS tile too large");
  static_assert(4 * G::PART * 4 <= 160 * 1024, "reduction tile too large");
  static_assert(G2::LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU need <= 80 KB each");
  static_assert(G::PART % 64 == 0, "record size");
  const bool v2 = wgrad_variant() == 2;
  constexpr int lds1 = (4 * G::WAVE_LDS > 4 * G::PART) ? 4 * G::WAVE_LDS : 4 * G::PART;
  const int lds_floats = v2 ? G2::LDS_FLOATS : lds1;
  const long units = v2 ? (long)a0.nb * G2::NG : ((long)a0.nb * G::HOUT + 3) / 4;
  // conv1 (SMALL): 63 MFMAs per output row against 38 staging loads + LDS writes that the wave cannot overlap with
  // its own MFMAs; its tile is 15 KB per wave and it needs 180 VGPRs, so two workgroups share a CU and one's staging
  // runs under the other's MFMAs
  long blocks = ((v2 || G::SMALL) ? 2L : 1L) * drq_num_cus();
  if (blocks > units) blocks = units;
  if (blocks < 1) blocks = 1;
  if ((size_t)blocks * G::PART * sizeof(float) > ws_bytes) return DRQ_EWS;
  if (((size_t)ws & 15) != 0) return DRQ_EARG;          // partial records are written 16 bytes at a time
  WgradArgs a = a0;
  a.part = ws;
  static bool attr_set_dev[kMaxDevices] = {};
  bool& attr_set = attr_set_dev[drq_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<CIN, HIN, STRIDE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds1 * 4);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv3x3_wgrad2_kernel<CIN, HIN, STRIDE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, G2::LDS_FLOATS * 4);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if constexpr (STRIDE == 1 && CIN == 32) {
    if (wgrad_variant() == 1) {          // default: rolling-tile kernel
      constexpr int wave3 = 4 * CIN * G::XP + 2 * 32 * G::DP + 128;
      constexpr int lds3 = (4 * wave3 > 4 * G::PART) ? 4 * wave3 : 4 * G::PART;
      static_assert(lds3 * 4 <= 160 * 1024, "rolling tile too large");
      static bool attr3_dev[kMaxDevices] = {};
      bool& attr3 = attr3_dev[drq_device()];
      if (!attr3) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad3_kernel<HIN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds3 * 4);
        if (e != hipSuccess) return (int)e;
        attr3 = true;
      }
      if (HIN == 41 && g_wgrad_stamps) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad3_kernel<41, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds3 * 4);
        WgradArgs as = a;
#ifdef DRQ_DEV
        if (getenv("DRQ_WGRAD_NOLOAD")) as.x_bytes = as.dy_bytes = 0;     // every load out of range: no memory traffic
#endif
        hipLaunchKernelGGL((conv3x3_wgrad3_kernel<41, true>), dim3((unsigned)blocks), dim3(256), lds3 * 4, st, as,
                           g_wgrad_stamps);
      } else
      hipLaunchKernelGGL((conv3x3_wgrad3_kernel<HIN>), dim3((unsigned)blocks), dim3(256), lds3 * 4, st, a,
                         (unsigned long long*)nullptr);
      DRQ_LAUNCH_CHECK();
      if (nblocks_out) *nblocks_out = (int)blocks;
      if (defer) return DRQ_OK;
      hipLaunchKernelGGL((conv3x3_wgrad_reduce_kernel<CIN, G::SMALL>), dim3(G::PART / 64), dim3(1024), 0, st,
                         (const float*)ws, (int)blocks, dw, db);
      DRQ_LAUNCH_CHECK();
      return DRQ_OK;
    }
  }
  if (v2)
    hipLaunchKernelGGL((conv3x3_wgrad2_kernel<CIN, HIN, STRIDE>), dim3((unsigned)blocks), dim3(256), lds_floats * 4,
                       st, a);
  else if (CIN == 9 && g_wgrad1_stamps) {
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<9, 84, 2, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<9, 84, 2, true>), dim3((unsigned)blocks), dim3(256), lds_floats * 4, st, a,
                       g_wgrad1_stamps);
  } else
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<CIN, HIN, STRIDE>), dim3((unsigned)blocks), dim3(256), lds_floats * 4,
                       st, a, (unsigned long long*)nullptr);
  DRQ_LAUNCH_CHECK();
  if (nblocks_out) *nblocks_out = (int)blocks;
  if (defer) return DRQ_OK;
  hipLaunchKernelGGL((conv3x3_wgrad_reduce_kernel<CIN, G::SMALL>), dim3(G::PART / 64), dim3(1024), 0, st,
                     (const float*)ws, (int)blocks, dw, db);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // namespace

// conv_wino_wgrad.hip
int drq_conv3x3_wgrad_partial_wino(const float* x, const float* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                   long dy_off, float* part, size_t part_bytes, int* nblocks, hipStream_t st);

// ---- internal entry points of the step orchestration (step.hip): per-layer partial sums now, one reduction later
int drq_conv3x3_wgrad_partial(const float* x, const float* dy, int nb, int cin, int hin, int stride, long dy_bs,
                              long dy_cs, long dy_rs, long dy_off, float* part, size_t part_bytes, int* nblocks,
                              hipStream_t st) {
  if (!x || !dy || !part || !nblocks || nb <= 0) return DRQ_EARG;
  const size_t xb = (size_t)nb * cin * hin * hin * 4;
  const size_t dyb = (size_t)nb * dy_bs * 4;
  if (xb >= (1ull << 31) || dyb >= (1ull << 31) || dy_off < 0 || dy_bs <= 0) return DRQ_EARG;
  WgradArgs a{x, dy, dy_bs, dy_cs, dy_rs, dy_off, nullptr, (unsigned)xb, (unsigned)dyb, nb};
  if (cin == 9 && hin == 84 && stride == 2)
    return launch_wgrad<9, 84, 2>(a, nullptr, nullptr, part, part_bytes, st, true, nblocks);
  if (cin == 32 && stride == 1) {
    if (hin == 41) return launch_wgrad<32, 41, 1>(a, nullptr, nullptr, part, part_bytes, st, true, nblocks);
    if (hin == 39) return launch_wgrad<32, 39, 1>(a, nullptr, nullptr, part, part_bytes, st, true, nblocks);
    if (hin == 37) return launch_wgrad<32, 37, 1>(a, nullptr, nullptr, part, part_bytes, st, true, nblocks);
  }
  return DRQ_EARG;
}

int drq_conv3x3_wgrad_reduce_multi(int n, const float* const* part, const int* nblocks, const int* cin,
                                   float* const* dw, float* const* db, hipStream_t st) {
  if (n <= 0 || n > 4 || !part || !nblocks || !cin || !dw || !db) return DRQ_EARG;
  ReduceJobs js{};
  int maxpart = 0;
  for (int i = 0; i < n; ++i) {
    if (!part[i] || !dw[i] || !db[i] || nblocks[i] <= 0) return DRQ_EARG;
    const int small = cin[i] * 3 <= 32;
    js.j[i] = ReduceJob{part[i], dw[i], db[i], nblocks[i], cin[i], small};
    const int p = (small ? 3 : 9) * 1024 + 64;
    if (p > maxpart) maxpart = p;
  }
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_multi_kernel, dim3(maxpart / 64, n), dim3(1024), 0, st, js);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI (declared in include/drqv2_hip.h)
// ------------------------------------------------------------------------------------------------
extern "C" {

#ifdef DRQ_DEV
// development hooks (libdrqv2_hip_dev.so only, never in the product library): device buffer of 32 u64 per wave
#pragma GCC visibility push(default)
void drq_dev_conv_stamps(void* p) { g_conv_stamps = (unsigned long long*)p; }
void drq_dev_wgrad_stamps(void* p) { g_wgrad_stamps = (unsigned long long*)p; }
void drq_dev_wgrad1_stamps(void* p) { g_wgrad1_stamps = (unsigned long long*)p; }
void drq_dev_conv_variant(int v) { g_conv_variant = v; }
#pragma GCC visibility pop
#endif

// y = relu?(conv3x3(x, w) + bias); x [nb][cin][hin][hin], y written with the given strides.
DRQ_API int drq_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int nb, int cin, int hin,
                    int stride, int relu, long y_bs, long y_cs, long y_rs, long y_off, hipStream_t st) {
  if (!x || !w || !y || nb <= 0) return DRQ_EARG;
  const size_t xb = (size_t)nb * cin * hin * hin * 4;
  if (xb >= (1ull << 31)) return DRQ_EARG;
  const size_t yb = (size_t)nb * y_bs * 4;
  if (yb >= (1ull << 31) || y_off < 0 || y_bs <= 0) return DRQ_EARG;
  ConvArgs a{x, w, bias, nullptr, y, y_bs, y_cs, y_rs, y_off, (unsigned)xb, (unsigned)yb, 0u, nb, relu, 0, nullptr};
  if (cin == 9 && hin == 84 && stride == 2) return launch_conv<9, 84, 2>(a, st);
  if (cin == 32 && stride == 1) {
    if (hin == 41) return launch_conv<32, 41, 1>(a, st);
    if (hin == 39) return launch_conv<32, 39, 1>(a, st);
    if (hin == 37) return launch_conv<32, 37, 1>(a, st);
  }
  return DRQ_EARG;
}

// dx = conv_transpose(dy, w) * (mask > 0): dy_pad is the pre-activation gradient stored zero-padded by 2
// ([nb][32][hout+4][hout+4]); the result has the layer-input size hin = hout+2 and is written with strides.
DRQ_API int drq_conv3x3_dgrad(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout,
                      long dx_bs, long dx_cs, long dx_rs, long dx_off, hipStream_t st) {
  if (!dy_pad || !w || !dx || nb <= 0) return DRQ_EARG;
  const int hp = hout + 4;
  const size_t xb = (size_t)nb * 32 * hp * hp * 4;
  if (xb >= (1ull << 31)) return DRQ_EARG;
  const size_t yb = (size_t)nb * dx_bs * 4;
  const size_t mb = (size_t)nb * 32 * (hout + 2) * (hout + 2) * 4;
  if (yb >= (1ull << 31) || dx_off < 0 || dx_bs <= 0) return DRQ_EARG;
  ConvArgs a{dy_pad, w, nullptr, mask, dx, dx_bs, dx_cs, dx_rs, dx_off, (unsigned)xb, (unsigned)yb, (unsigned)mb,
             nb, 0, 1, nullptr};
  if (hp == 39) return launch_conv<32, 39, 1>(a, st);
  if (hp == 41) return launch_conv<32, 41, 1>(a, st);
  if (hp == 43) return launch_conv<32, 43, 1>(a, st);
  return DRQ_EARG;
}

// dw[32][cin][3][3], db[32] from the layer input x and the pre-activation gradient dy (strided view).
DRQ_API int drq_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int nb, int cin, int hin,
                      int stride, long dy_bs, long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes,
                      hipStream_t st) {
  if (!x || !dy || !dw || !db || !ws || nb <= 0) return DRQ_EARG;
  const size_t xb = (size_t)nb * cin * hin * hin * 4;
  const size_t dyb = (size_t)nb * dy_bs * 4;
  if (xb >= (1ull << 31) || dyb >= (1ull << 31) || dy_off < 0 || dy_bs <= 0) return DRQ_EARG;
  WgradArgs a{x, dy, dy_bs, dy_cs, dy_rs, dy_off, nullptr, (unsigned)xb, (unsigned)dyb, nb};
  if (cin == 9 && hin == 84 && stride == 2) return launch_wgrad<9, 84, 2>(a, dw, db, ws, ws_bytes, st);
  if (cin == 32 && stride == 1) {
    if (hin == 41) return launch_wgrad<32, 41, 1>(a, dw, db, ws, ws_bytes, st);
    if (hin == 39) return launch_wgrad<32, 39, 1>(a, dw, db, ws, ws_bytes, st);
    if (hin == 37) return launch_wgrad<32, 37, 1>(a, dw, db, ws, ws_bytes, st);
  }
  return DRQ_EARG;
}

// The same gradients of the 32->32 layers in Winograd F(2x2,3x3) form (conv_wino_wgrad.hip): partial records in the
// same format, the same fixed-order reduction.
DRQ_API int drq_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, float* db, int nb, int hin, long dy_bs,
                                   long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes, hipStream_t st) {
  if (!x || !dy || !dw || !db || !ws || nb <= 0) return DRQ_EARG;
  int nblocks = 0;
  const int rc = drq_conv3x3_wgrad_partial_wino(x, dy, nb, hin, dy_bs, dy_cs, dy_rs, dy_off, ws, ws_bytes, &nblocks, st);
  if (rc != 0) return rc;
  constexpr int PART = 9 * 1024 + 64;
  hipLaunchKernelGGL((conv3x3_wgrad_reduce_kernel<32, false>), dim3(PART / 64), dim3(1024), 0, st, (const float*)ws,
                     nblocks, dw, db);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API size_t drq_conv3x3_wgrad_ws_bytes(void) { return (size_t)1024 * (9 * 1024 + 64) * sizeof(float); }

}  // extern "C"
