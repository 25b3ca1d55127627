// 3x3 stride-1 convolution of the encoder's 32->32 layers in Winograd F(2x2,3x3) form on the f32 matrix cores.
//
// Reference op: nn.Conv2d(32,32,3,1) + ReLU (drqv2.py:56-58) and its input gradient.  Same interface and data
// layouts as conv3x3_kernel (conv.hip); the arithmetic is the minimal-filtering form
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray), summed over the 32 input channels,
// i.e. per 2x2 output tile 16 channel-GEMMs of 32x32 instead of 36 tap-GEMMs: 2.25x fewer matrix FLOPs, paid for
// with 32 adds per (tile, input channel) and 24 adds per (tile, output channel) on the VALU.  All of it fp32; the
// result differs from the direct form by rounding only (measured 1.9e-7 vs 1.6e-7 normwise against fp64).
//
// Mapping: v_mfma_f32_16x16x4_f32, D[cout 16][tile 16] += U[cout][cin 4] * V[cin 4][tile].  A wave owns a "unit" of
// 16 consecutive tiles of the flattened (sample, tile row, tile column) index and all 32 output channels:
// 16 positions x 2 cout halves x 4 registers = 128 accumulator registers, two waves per SIMD.  Lane l holds tile
// l&15 and input channel 4c + (l>>4) of k-step c: it loads that channel's 4x4 input patch (four 16-byte loads),
// transforms it in registers and feeds the 16 positions.  U = G g G^T is computed once per workgroup into LDS in
// MFMA-lane order (one ds_read_b64 per position and k-step serves both cout halves).
#include "common.h"
#include "wino_u.h"
#include <type_traits>

namespace {

struct WinoArgs {
  const float* x;      // [NB][32][HIN][HIN]
  const float* w;      // canonical [32][32][3][3]
  const float* pre;    // optional: the U image of (w, wmode) in global memory (wino_u.h), else null
  const float* bias;   // [32] or null
  const float* mask;   // [NB][32][HOUT][HOUT] or null : out *= (mask > 0)
  float* y;
  int y_bs, y_cs, y_rs, y_off;   // output strides (elements)
  unsigned x_bytes, y_bytes, mask_bytes;
  int nb;
  int relu;
  int wmode;           // 0 forward gather, 1 dgrad gather (transposed + flipped)
  int stagger;         // start delay of the second wave of each SIMD, in units of 1024 clocks (see the kernel)
  unsigned long long* stamps;   // development only (ABL & 8): 32 per wave
};

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

#ifdef DRQ_DEV
int g_wino_variant = 0;   // drq_dev_wino_variant: timing ablations (tools/wino_bench.py)
int g_wino_stagger = -1;  // drq_dev_wino_stagger: overrides kStagger
unsigned long long* g_wino_stamps = nullptr;
#endif
constexpr int kStagger = 0;

// ABL (development build only): 1 = no input transform (V = d), 2 = no output transform (position 0 is stored),
// 4 = no patch loads; the results are wrong on purpose; 8 = per-wave s_memtime stamps (start, after the prologue,
// after every unit; [29],[30] = s_memrealtime end / start, [31] = HW_ID)
template <int HIN, bool MASK, bool RELU, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(WinoArgs a) {
#pragma clang fp contract(off)
  constexpr int HOUT = HIN - 2;
  constexpr int TH = (HOUT + 1) / 2;         // tiles per dimension (the last one is half empty: HOUT is odd)
  constexpr int TT = TH * TH;
  constexpr int PLANE = HIN * HIN * 4;       // bytes of one input channel
  __shared__ __attribute__((aligned(16))) float U[16 * 8 * 64 * 2];   // [pos][k-step][lane][cout half]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  unsigned long long* stamp = nullptr;
  int nstamp = 0;
  auto mark = [&]() {
    if constexpr (ABL & 8) {
      if (stamp && nstamp < 28 && lane == 0) stamp[nstamp] = __builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  if constexpr (ABL & 8) {
    if (a.stamps) {
      stamp = a.stamps + ((size_t)blockIdx.x * 4 + wid) * 32;
      if (lane == 0) {
        stamp[30] = __builtin_amdgcn_s_memrealtime();
        stamp[31] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        stamp[28] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
      }
    }
  }
  mark();
  const int tl = lane & 15;     // tile of the unit (B column / D column)
  const int kk = lane >> 4;     // input channel inside the k-step (A/B k index); D rows 4*kk + r

  // ---- U = G g G^T for the 1024 (cout, cin) filters -> LDS: a plain copy of the image a rider of conv1_aug_kernel
  // prepared for this update (a.pre), else computed here (wino_u.h)
  if (a.pre) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.pre);
    f32x4* dst = reinterpret_cast<f32x4*>(U);
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i * 256 + tid] = v[i];
  } else {
    wino_u_image<true>(a.w, a.wmode, U, U, tid);
  }
  // bias of this lane's eight output channels: cout = 16*h + 4*kk + r
  float bv[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[h][r] = a.bias ? a.bias[16 * h + 4 * kk + r] : 0.f;
  __syncthreads();

  // The two waves of a SIMD come from two workgroups that start together and run the same program: left alone they
  // stay in lockstep -- both in their transform / epilogue phases at the same time, the matrix pipe idle meanwhile.
  // The wave in the odd slot of its SIMD (HW_ID bits 3:0) starts late, so that one wave's non-matrix phases fall
  // into the other's MFMA runs.  Speed only.
  if (a.stagger) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    if (hwid & 1u)
      for (int q = 0; q < a.stagger; ++q) __builtin_amdgcn_s_sleep(16);
  }

  mark();
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, 0, a.mask ? a.mask_bytes : 0, 0x00020000);
  constexpr int kDrop = (int)0x80000000u;    // beyond any num_records: the store is discarded

  const int ntile = a.nb * TT;
  const int nunit = (ntile + 15) >> 4;
  const int nwave = (int)gridDim.x * 4;
  const f32x2w* Ul = reinterpret_cast<const f32x2w*>(U) + lane;

  auto patch_voff = [&](int unit) {          // byte offset of this lane's patch (channel kk of k-step 0)
    int t = unit * 16 + tl;
    t = t < ntile ? t : ntile - 1;
    const int b = t / TT, rem = t - b * TT;
    const int ty = rem / TH, tx = rem - ty * TH;
    return (((b * 32 + kk) * HIN + 2 * ty) * HIN + 2 * tx) * 4;
  };
  auto load_patch_row = [&](float (&d)[16], int voff, int c, int i) {
    if constexpr (ABL & 4) {
      d[i * 4 + 0] = 1.0f + (float)(voff + c); d[i * 4 + 1] = 2.0f + (float)(voff + i);
      d[i * 4 + 2] = 1.5f + (float)voff; d[i * 4 + 3] = (float)(c + i);
      return;
    }
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * (4 * PLANE) + i * (HIN * 4), 0);
    const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
    d[i * 4 + 0] = __uint_as_float(e0);
    d[i * 4 + 1] = __uint_as_float(e1);
    d[i * 4 + 2] = __uint_as_float(e2);
    d[i * 4 + 3] = __uint_as_float(e3);
  };
  auto load_patch = [&](float (&d)[16], int voff, int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (ABL & 4) {
        d[i * 4 + 0] = 1.0f + (float)(voff + c); d[i * 4 + 1] = 2.0f + (float)(voff + i);
        d[i * 4 + 2] = 1.5f + (float)voff; d[i * 4 + 3] = (float)(c + i);
        continue;
      }
      const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * (4 * PLANE) + i * (HIN * 4), 0);
      const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
      d[i * 4 + 0] = __uint_as_float(e0);
      d[i * 4 + 1] = __uint_as_float(e1);
      d[i * 4 + 2] = __uint_as_float(e2);
      d[i * 4 + 3] = __uint_as_float(e3);
    }
  };

  // A operands: positions [8*g, 8*g+8) of k-step c -> 16 registers (both cout halves per ds_read_b64)
  auto load_A = [&](f32x2w (&A)[8], int c, int g) {
#pragma unroll
    for (int p = 0; p < 8; ++p) A[p] = Ul[((8 * g + p) * 8 + c) * 64];
  };

  // ---- this wave's work.  Units are dealt round-robin: wave gw takes units gw, gw + nwave, ... of the first q rounds.
  // The r = nunit - q*nwave units left over are cut in HALF-units (16 tiles x one half of the output channels: the
  // MFMAs of one channel half, the whole input transform) wherever that lowers the busiest SIMD's load: with 2r <=
  // nwave every left-over unit goes to two neighbouring waves as two halves (a SIMD then carries 2q + 1/2 or 2q + 1
  // units instead of 2q + 1 or 2q + 2 -- the two workgroups of a CU are blockIdx b and b + 256, so the waves
  // [0, 1024) sit on distinct SIMDs); with 2r > nwave the first nf = 2r - nwave waves take a whole unit, every other
  // wave a half.  Results do not depend on the split (the same MFMAs in the same order per accumulator).
  // Workgroup -> position in a round, XCD-aware: the hardware deals workgroups to the eight XCDs round-robin
  // (blockIdx & 7; a speed assumption only), and neighbouring workgroups of a round read neighbouring tile rows, which
  // share two of their four input rows.  With the plain order those re-reads came from eight different L2s (1.5x the
  // compulsory fabric traffic); here each half of the grid (one workgroup per CU) gives every XCD a contiguous
  // eighth of the round, and the two halves stay as they are (see the left-over rule below).
  const int hg = (int)gridDim.x >> 1;
  const int bh = (int)blockIdx.x >= hg ? 1 : 0, bl = (int)blockIdx.x - bh * hg;
  const int lb = ((int)gridDim.x & 15) ? (int)blockIdx.x : bh * hg + (bl & 7) * (hg >> 3) + (bl >> 3);
  const int gw = lb * 4 + wid;
  const int q = nunit / nwave, r = nunit - q * nwave;
  const int nf = 2 * r > nwave ? 2 * r - nwave : 0;
  const int nfull = q + (gw < nf ? 1 : 0);
  const int hk = gw - nf;                                   // index among the half-units
  const bool has_half = hk >= 0 && hk < 2 * (r - nf);
  const int half_unit = q * nwave + nf + (hk >> 1), half_h = hk & 1;
  auto item_unit = [&](int i) { return i < nfull ? gw + i * nwave : half_unit; };   // i-th work item of this wave
  const int nitem = nfull + (has_half ? 1 : 0);

  float d[16];
  f32x2w A0[8], A1[8];
  int voff = 0;
  if (nitem > 0) {
    voff = patch_voff(item_unit(0));
    load_patch(d, voff, 0);
    load_A(A0, 0, 0);
  }
  // HSEL: 2 = whole unit, 0 / 1 = the half-unit of output channels [0,16) / [16,32)
  auto body = [&](auto hsel_tag, int unit, int next_unit) {
    constexpr int HSEL = decltype(hsel_tag)::value;
    const int nvoff = patch_voff(next_unit);
    // where this unit's outputs (and the mask) live
    const int t0 = unit * 16 + tl;
    const bool valid = t0 < ntile;
    const int t = valid ? t0 : ntile - 1;
    const int b = t / TT, rem = t - b * TT;
    const int ty = rem / TH, tx = rem - ty * TH;
    const bool c1ok = 2 * tx + 1 < HOUT, r1ok = 2 * ty + 1 < HOUT;
    const int ybase = (a.y_off + b * a.y_bs + (4 * kk) * a.y_cs + (2 * ty) * a.y_rs + 2 * tx) * 4;
    const int mbase = (((b * 32 + 4 * kk) * HOUT + 2 * ty) * HOUT + 2 * tx) * 4;
    // per-lane store offsets of the unit: [row][8-byte pair | lone first column of the last tile of a row]
    // (invalid -> beyond num_records: the buffer unit drops the store; no branches in the epilogue)
    int o64[2], o32[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rowok = valid && (i == 0 || r1ok);
      o64[i] = (rowok && c1ok) ? ybase : kDrop;
      o32[i] = (rowok && !c1ok) ? ybase : kDrop;
    }
    f32x4 acc[16][2];
    u32x2w mk[2][4][2];                      // the ReLU mask of the layer below (MASK)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // Every k-step is pinned into the same issue order (sched_barrier): the patch of the NEXT step and the A
      // operands of the next half-step are requested a half-step (>= 512 matrix cycles) before they are used.
      // The memory clobber keeps the loop-invariant LDS reads from being hoisted out of the loops (256 registers).
      asm volatile("" ::: "memory");
      float t[16];                           // B^T d
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ABL & 1) {
          t[0 * 4 + j] = d[0 * 4 + j]; t[1 * 4 + j] = d[1 * 4 + j]; t[2 * 4 + j] = d[2 * 4 + j]; t[3 * 4 + j] = d[3 * 4 + j];
          continue;
        }
        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
      }
      __builtin_amdgcn_sched_barrier(0);
      load_A(A1, c, 1);
      __builtin_amdgcn_sched_barrier(0);
      float V[16];                           // (B^T d) B, all of it ahead of the MFMAs (no VALU->MFMA hazard per use)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ABL & 1) {
          V[i * 4 + 0] = t[i * 4 + 0]; V[i * 4 + 1] = t[i * 4 + 1]; V[i * 4 + 2] = t[i * 4 + 2]; V[i * 4 + 3] = t[i * 4 + 3];
          continue;
        }
        V[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
        V[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
        V[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
        V[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
      }
      auto half_step = [&](const f32x2w (&A)[8], int g) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          if (g == 0 && (p & 1) == 0) {
            // the next patch (next k-step, or the next unit's first): one row ahead of every four MFMAs of the first
            // half-step rather than four loads in a burst
            __builtin_amdgcn_sched_barrier(0);
            load_patch_row(d, c + 1 < 8 ? voff : nvoff, c + 1 < 8 ? c + 1 : 0, p >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          const int pos = 8 * g + p;
          const f32x2w Ap = A[p];
          if (c == 0) {
            // position (1,1) enters all four outputs of the tile with weight +1: the bias rides in its accumulator
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 b0 = {bv[0][0], bv[0][1], bv[0][2], bv[0][3]}, b1 = {bv[1][0], bv[1][1], bv[1][2], bv[1][3]};
            if constexpr (HSEL != 1) acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[0], V[pos], pos == 5 ? b0 : z, 0, 0, 0);
            if constexpr (HSEL != 0) acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[1], V[pos], pos == 5 ? b1 : z, 0, 0, 0);
          } else {
            if constexpr (HSEL != 1) acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[0], V[pos], acc[pos][0], 0, 0, 0);
            if constexpr (HSEL != 0) acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[1], V[pos], acc[pos][1], 0, 0, 0);
          }
        }
      };
      half_step(A0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_A(A0, (c + 1) & 7, 0);            // the next k-step's (or the next unit's first) lower half
      if constexpr (MASK) {
        if (c == 7) {                        // the first half of the mask flies under the unit's last sixteen MFMAs
          constexpr int h0 = HSEL == 1 ? 1 : 0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              mk[h0][r][i] = __builtin_amdgcn_raw_buffer_load_b64(mrs, mbase + i * (HOUT * 4),
                                                                  (16 * h0 + r) * (HOUT * HOUT * 4), 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      half_step(A1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- output transform Y = A^T M A, bias, ReLU / mask, stores
    {
      if constexpr (ABL & 2) {               // every accumulator stays live although only four are stored
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
          if constexpr (HSEL != 1) asm volatile("" ::"v"(acc[pos][0]));
          if constexpr (HSEL != 0) asm volatile("" ::"v"(acc[pos][1]));
        }
      }
      if constexpr (MASK && HSEL == 2) {     // ... the second half under the first half's transform arithmetic
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            mk[1][r][i] = __builtin_amdgcn_raw_buffer_load_b64(mrs, mbase + i * (HOUT * 4), (16 + r) * (HOUT * HOUT * 4), 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int ycs4 = a.y_cs * 4, yrs4 = a.y_rs * 4;
#pragma unroll
      for (int h = (HSEL == 1 ? 1 : 0); h < (HSEL == 0 ? 1 : 2); ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s[2][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float m0 = acc[0 * 4 + j][h][r], m1 = acc[1 * 4 + j][h][r], m2 = acc[2 * 4 + j][h][r],
                        m3 = acc[3 * 4 + j][h][r];
            s[0][j] = (ABL & 2) ? m0 : (m0 + m1) + m2;
            s[1][j] = (ABL & 2) ? m0 + m3 : (m1 - m2) - m3;
          }
          const int co = (16 * h + r);       // + 4*kk rides in the lane's base offset
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float y0 = (ABL & 2) ? s[i][0] : (s[i][0] + s[i][1]) + s[i][2];
            float y1 = (ABL & 2) ? s[i][3] : (s[i][1] - s[i][2]) - s[i][3];
            if constexpr (RELU) {
              y0 = __builtin_fmaxf(y0, 0.f);
              y1 = __builtin_fmaxf(y1, 0.f);
            }
            if constexpr (MASK) {
              const unsigned q0 = mk[h][r][i][0], q1 = mk[h][r][i][1];
              y0 = __uint_as_float(q0) > 0.f ? y0 : 0.f;
              y1 = __uint_as_float(q1) > 0.f ? y1 : 0.f;
            }
            const int soff = co * ycs4 + i * yrs4;                 // wave-uniform: scalar offset
            const u32x2w pk = {__float_as_uint(y0), __float_as_uint(y1)};
            __builtin_amdgcn_raw_buffer_store_b64(pk, yrs, o64[i], soff, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y0), yrs, o32[i], soff, 0);
          }
        }
    }
    voff = nvoff;
    mark();
  };
  for (int i = 0; i < nfull; ++i) body(std::integral_constant<int, 2>{}, item_unit(i), item_unit(i + 1 < nitem ? i + 1 : i));
  if (has_half) {
    if (half_h == 0) body(std::integral_constant<int, 0>{}, half_unit, half_unit);
    else body(std::integral_constant<int, 1>{}, half_unit, half_unit);
  }
  if constexpr (ABL & 8) {
    __builtin_amdgcn_s_waitcnt(0);
    if (stamp && lane == 0) stamp[29] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int HIN>
int launch_wino(const WinoArgs& a0, hipStream_t st) {
  WinoArgs a = a0;
  a.stagger = kStagger;
#ifdef DRQ_DEV
  if (g_wino_stagger >= 0) a.stagger = g_wino_stagger;
#endif
  constexpr int HOUT = HIN - 2, TH = (HOUT + 1) / 2;
  const long nunit = ((long)a.nb * TH * TH + 15) / 16;
  long blocks = (nunit + 3) / 4;
  const long cap = 2L * drq_num_cus();
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
#ifdef DRQ_DEV
  if (!a.mask && a.relu && g_wino_variant) {
    switch (g_wino_variant) {
      case 1: hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 3: hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 3>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 4: hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 8: a.stamps = g_wino_stamps; hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 15: a.stamps = g_wino_stamps; hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 15>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true, 7>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    DRQ_LAUNCH_CHECK();
    return DRQ_OK;
  }
#endif
  const dim3 g((unsigned)blocks), t(256);
  if (a.mask && a.relu) hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, true, true>), g, t, 0, st, a);
  else if (a.mask) hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, true, false>), g, t, 0, st, a);
  else if (a.relu) hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, true>), g, t, 0, st, a);
  else hipLaunchKernelGGL((conv3x3_wino_kernel<HIN, false, false>), g, t, 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // namespace

#ifdef DRQ_DEV
extern "C" DRQ_API void drq_dev_wino_variant(int v) { g_wino_variant = v; }
extern "C" DRQ_API void drq_dev_wino_stagger(int v) { g_wino_stagger = v; }
extern "C" DRQ_API void drq_dev_wino_stamps(void* p) { g_wino_stamps = (unsigned long long*)p; }
#endif

// internal (step.hip): u_image = the layer's U image prepared by conv1_aug_kernel's rider for this update, or null
int drq_conv3x3_fwd_wino_pre(const float* x, const float* w, const float* u_image, const float* bias, float* y, int nb,
                             int hin, int relu, long y_bs, long y_cs, long y_rs, long y_off, hipStream_t st) {
  if (!x || !w || !y || nb <= 0) return DRQ_EARG;
  const size_t xb = (size_t)nb * 32 * hin * hin * 4;
  const size_t yb = (size_t)nb * y_bs * 4;
  if (xb >= (1ull << 31) || yb >= (1ull << 31) || y_off < 0 || y_bs <= 0) return DRQ_EARG;
  WinoArgs a{x, w, u_image, bias, nullptr, y, (int)y_bs, (int)y_cs, (int)y_rs, (int)y_off, (unsigned)xb, (unsigned)yb, 0u, nb, relu, 0, 0, nullptr};
  if (hin == 41) return launch_wino<41>(a, st);
  if (hin == 39) return launch_wino<39>(a, st);
  if (hin == 37) return launch_wino<37>(a, st);
  return DRQ_EARG;
}

int drq_conv3x3_dgrad_wino_pre(const float* dy_pad, const float* w, const float* u_image, const float* mask, float* dx,
                               int nb, int hout, long dx_bs, long dx_cs, long dx_rs, long dx_off, hipStream_t st) {
  if (!dy_pad || !w || !dx || nb <= 0) return DRQ_EARG;
  const int hp = hout + 4;
  const size_t xb = (size_t)nb * 32 * hp * hp * 4;
  const size_t yb = (size_t)nb * dx_bs * 4;
  const size_t mb = (size_t)nb * 32 * (hout + 2) * (hout + 2) * 4;
  if (xb >= (1ull << 31) || yb >= (1ull << 31) || dx_off < 0 || dx_bs <= 0) return DRQ_EARG;
  WinoArgs a{dy_pad, w, u_image, nullptr, mask, dx, (int)dx_bs, (int)dx_cs, (int)dx_rs, (int)dx_off, (unsigned)xb, (unsigned)yb,
             (unsigned)mb, nb, 0, 1, 0, nullptr};
  if (hp == 39) return launch_wino<39>(a, st);
  if (hp == 41) return launch_wino<41>(a, st);
  if (hp == 43) return launch_wino<43>(a, st);
  return DRQ_EARG;
}

extern "C" {

// Same contract as drq_conv3x3_fwd (conv.hip) for cin = 32, stride 1, in Winograd form (rounding differs).
DRQ_API int drq_conv3x3_fwd_wino(const float* x, const float* w, const float* bias, float* y, int nb, int hin, int relu,
                                 long y_bs, long y_cs, long y_rs, long y_off, hipStream_t st) {
  return drq_conv3x3_fwd_wino_pre(x, w, nullptr, bias, y, nb, hin, relu, y_bs, y_cs, y_rs, y_off, st);
}

// Same contract as drq_conv3x3_dgrad (conv.hip) in Winograd form.
DRQ_API int drq_conv3x3_dgrad_wino(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout,
                                   long dx_bs, long dx_cs, long dx_rs, long dx_off, hipStream_t st) {
  return drq_conv3x3_dgrad_wino_pre(dy_pad, w, nullptr, mask, dx, nb, hout, dx_bs, dx_cs, dx_rs, dx_off, st);
}

}  // extern "C"
