// First encoder layer fused with the augmentation: RandomShiftsAug (drqv2.py:19-45) + obs/255-0.5 (drqv2.py:64)
// + Conv2d(9,32,3,stride 2) + ReLU (drqv2.py:55) in ONE kernel that reads the uint8 frames once.
//
// Before: aug_rows_kernel wrote both augmented views as fp32 (130 MB at B=256) and conv1 read them back through
// 12-byte tap loads (47.6 + 72 us, 40 % MFMA / 42 % HBM).  Here a workgroup owns one (frame, band of output rows):
//   stage 1  the uint8 source rows of the band (<= 21 rows x 9 channels) -> LDS with dword loads;
//   stage 2  the augmented, normalised fp32 rows of the band (<= 19 rows x 9 channels) -> LDS, computed with the
//            statements of aug_rows_kernel (bit-identical results); frames of the obs view are also stored to
//            global memory, because conv1's weight gradient reads them in the backward pass;
//   stage 3  implicit GEMM on v_mfma_f32_32x32x2_f32 with the B operand read from that LDS tile (columns stored
//            de-interleaved by parity: the stride-2 taps of 32 neighbouring pixels are conflict-free reads) and the
//            A operand (this lane's 45 weight values) held in registers for the whole kernel.
// Two workgroups share a CU: one's stage 1 (memory) runs under the other's compute.  Stage 2 and stage 3 do NOT
// overlap, although they come from different waves: on gfx950 the f32 MFMA and the f32 VALU share a SIMD's
// datapath (ablations in tools/conv1aug_ab.py: augmentation alone 43 us, tiles alone 54 us, both 89 us; per-CU stamps
// in tools/conv1aug_stamps.py show both workgroups resident all the time; staggering them changes nothing).
// k order and accumulation order are those of conv3x3_kernel<9,84,2>: the results are bit-identical to the
// unfused path (tests/test_hip_ops.py).
#include "common.h"
#include "wino_u.h"
#include <stdio.h>

namespace {

#ifdef DRQ_DEV
int g_conv1aug_variant = 0;
int g_conv1aug_stagger = 1;
unsigned long long* g_conv1aug_stamps = nullptr;
#endif

constexpr int H = 84, C = 9, PAD = 4, S = 92, HW = H * H;
constexpr int HO = 41, PO = HO * HO;
constexpr int NBAND = 5;                 // output rows [0,9) [9,17) [17,25) [25,33) [33,41)
constexpr int MAXR = 9;
constexpr int NROWS = 2 * MAXR + 1;      // augmented rows a band needs (stride 2, 3 taps)
constexpr int SROWS = NROWS + 2;         // source rows those touch (floor may land one lower, +1 for the second tap)
constexpr int XPL = 42;                  // columns per parity plane
constexpr int XPITCH = 2 * XPL;          // floats per augmented row: [even columns | odd columns]
constexpr int HQ = H / 4;                // dwords per source row
constexpr int CP = 5, NS = CP * 9;       // channel pairs, MFMA steps per tile (k = 90: channel 9 is a zero weight)
constexpr int XS_FLOATS = C * NROWS * XPITCH;
constexpr int U8_DWORDS = C * SROWS * HQ;                 // source tile [9][SROWS][HQ], filled by LDS-DMA
constexpr int NDMA = (U8_DWORDS + 63) / 64;               // wave-instructions of 64 dwords per tile
constexpr int U8_ALLOC = NDMA * 64;                       // the last instruction's tail lands in padding
constexpr int NTHR = 256;                                 // 4 waves (512 with waves 4..7 augmenting only: stage 2 40 instead
                                                          // of 50 us, but the stage times add up either way: 90 vs 86 us)
constexpr int NWAVE = NTHR / 64;
constexpr int MAXU = 64;                                  // units per workgroup (shift table); the grid grows beyond it
constexpr int LDS_BYTES = (XS_FLOATS + U8_ALLOC + H + 3 * MAXU) * 4;   // + base grid, shift table, frame table

struct Conv1AugArgs {
  const uint8_t* obs[2];     // view 0 = obs, view 1 = next_obs: [n][9][84][84] -- or, with fidx, a store of frames
  const long* fidx[2];       // optional: row b of the view is frame fidx[view][b] of obs[view] (device replay: the batch is
                             // never materialised, the kernel gathers its source rows straight from the store)
  const float* shift[2];     // [n][2] (x, y)
  const float* base;         // [84]
  const float* w;            // [32][9][3][3]
  const float* bias;         // [32]
  float* xaug;               // [2n][9][84][84]: frames < n_store are written (obs view: needed by conv1's wgrad)
  float* y;                  // [2n][32][41][41]
  int n, n_store;
  unsigned y_bytes;
  int stagger;               // start delay of the second half of the grid, in units of 127*64 clocks
  // riders: the first n_rider (0 or 6) workgroups do not touch frames; workgroup 2*l + m writes the Winograd image of
  // layer l+2's weights (m = 0 forward, 1 input-gradient form) to wino_u + (2*l + m)*16384 (wino_u.h).  The three
  // 32->32 layers of THIS update read them instead of transforming the weights in every workgroup's prologue.
  const float* wino_w[3];
  float* wino_u;
  int n_rider;
  unsigned long long* stamps;   // development build: [grid][4] = start, end (s_memrealtime), HW_ID, XCC_ID
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_c1 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_c1 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_c1 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16_c1(float lo, float hi) {      // RNE, lo in bits 15:0
  typedef __bf16 bf16x2_c1 __attribute__((ext_vector_type(2)));
  const bf16x2_c1 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ float div255(float v) {   // correctly rounded v / 255 (see elementwise.hip)
  const float r = 1.0f / 255.0f;
  const float q = __fmul_rn(v, r);
  const float e = __fmaf_rn(-q, 255.0f, v);
  return __fmaf_rn(e, r, q);
}

// ABL (development build only, tools/conv1aug_ab.py): timing ablations -- 1 skips stage 3 (tiles), 2 skips stage 2
// (augmentation), 4 skips the stage-2 global stores, 8 skips the LDS-DMA of the source rows
// BF (the bf16 update path, DrqStep.bf16): stage 3 runs on v_mfma_f32_32x32x16_bf16 -- one MFMA per tap with k = 16
// channel slots (9 real channels; lanes 0-31 supply channels 0-7, lanes 32-63 channel 8 and seven zero weights), the
// operands rounded to bf16 as they are read from the fp32 LDS tile.  Stages 1 and 2 (and the stored encoder input)
// are unchanged.  The bf16 matrix unit is separate from the f32 VALU datapath: here the two workgroups of a CU DO
// overlap (one augments while the other multiplies).
// YN (with BF): y is written as bf16 [frame][41][41][32 channels] (conv_bf16.hip's activation layout) instead of fp32 NCHW
template <int ABL, bool BF = false, bool YN = false>
__global__ __launch_bounds__(NTHR, 2 * NTHR / 256) void conv1_aug_kernel(Conv1AugArgs a) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  float* xs = smem;                                               // [9][NROWS][XPITCH]
  unsigned* u8w = reinterpret_cast<unsigned*>(smem + XS_FLOATS);   // [9][SROWS][HQ] dwords (+ padding)
  const uint8_t* u8b = reinterpret_cast<const uint8_t*>(u8w);
  float* bg = smem + XS_FLOATS + U8_ALLOC;                         // [84] base grid
  float* shs = bg + H;                                             // [MAXU][2] shifts (x, y) of this workgroup's units
  int* fof = reinterpret_cast<int*>(shs + 2 * MAXU);               // [MAXU] frame number of the unit's source in obs[view]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  if ((int)blockIdx.x < a.n_rider) {         // wave-uniform: the whole workgroup takes the rider's path
    const int job = blockIdx.x;
    wino_u_image<false>(a.wino_w[job >> 1], job & 1, smem, a.wino_u + (size_t)job * 16384, tid);
    return;
  }
  const int bid = (int)blockIdx.x - a.n_rider;
  const int units = 2 * a.n * NBAND;
  const int G = (int)gridDim.x - a.n_rider;

#ifdef DRQ_DEV
  if (a.stamps && tid == 0) {
    a.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
    a.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    a.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
  }
#endif
  if (tid < H) bg[tid] = a.base[tid];
  if (tid >= 128 && tid < 128 + MAXU) {                            // the host sizes the grid so that MAXU covers them
    const int k = tid - 128, uu = bid + k * G;
    if (uu < units) {
      const int f = uu / NBAND;
      const int view = f >= a.n ? 1 : 0, fb = f - view * a.n;
      shs[2 * k + 0] = a.shift[view][2 * fb + 0];
      shs[2 * k + 1] = a.shift[view][2 * fb + 1];
      fof[k] = a.fidx[view] ? (int)a.fidx[view][fb] : fb;
    }
  }
  // A operand: this lane's weights, step s = c*9 + t -> w[cout = col][cin = 2c + half][t]; cin 9 does not exist
  float wreg[NS];
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cin = 2 * c + half;
      wreg[c * 9 + t] = cin < C ? a.w[(col * C + cin) * 9 + t] : 0.f;
    }
  // BF: fragment t (tap) holds w[cout = col][channel 8*half + j][t], j = 0..7 (channels >= 9: zero)
  bf16x8_c1 wfb[9];
  if constexpr (BF) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cin = 8 * half + j;
        wv[j] = cin < C ? a.w[(col * C + (cin < C ? cin : 0)) * 9 + t] : 0.f;
      }
      const u32x4_c1 pk = {pack_bf16_c1(wv[0], wv[1]), pack_bf16_c1(wv[2], wv[3]), pack_bf16_c1(wv[4], wv[5]),
                           pack_bf16_c1(wv[6], wv[7])};
      wfb[t] = __builtin_bit_cast(bf16x8_c1, pk);
    }
  }
  float breg[16];     // accumulator row (cout) of register r: (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int r = 0; r < 16; ++r) breg[r] = a.bias[(r & 3) + 8 * (r >> 2) + 4 * half];
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  __syncthreads();                                                 // bg, shs, fof

  const float sc = (float)(2.0 / (double)S);
  auto cl = [&](int v) { v -= PAD; return v < 0 ? 0 : (v > H - 1 ? H - 1 : v); };
  auto coordv = [&](int k, float sh, float& fl) {            // unnormalised sample coordinate of grid index k
    const float g = bg[k] + sh;
    const float v = ((g + 1.f) * (float)S - 1.f) / 2.f;
    fl = floorf(v);
    return v;
  };

  // unit = (frame of the stacked [2n] batch, band of output rows); everything here is wave-uniform and comes from
  // LDS tables: no vector-memory load (and so no vmcnt wait) between the DMA issue and the tiles
  struct Unit {
    int f, r0, R, i0, nr, sy_lo, nsrc, fo;
    float shx, shy;
  };
  auto make_unit = [&](int k) {
    const int u = bid + k * G;
    Unit q;
    q.f = u / NBAND;
    const int band = u - q.f * NBAND;
    q.r0 = band == 0 ? 0 : 9 + 8 * (band - 1);
    q.R = band == 0 ? 9 : 8;
    // augmented rows [i0, i0 + nr).  The last band also produces row 83: no stride-2 window reaches it, but the
    // stored encoder input is then complete (it is compared with the reference's aug output element by element)
    q.i0 = 2 * q.r0;
    q.nr = 2 * q.R + 1 + (band == NBAND - 1 ? 1 : 0);
    q.shx = shs[2 * k + 0] * sc;
    q.shy = shs[2 * k + 1] * sc;
    q.fo = fof[k];
    float fa, fbb;
    coordv(q.i0, q.shy, fa);
    coordv(q.i0 + q.nr - 1, q.shy, fbb);
    q.sy_lo = cl((int)fa);                                   // the grid is increasing: rows in between lie inside
    q.nsrc = cl((int)fbb + 1) - q.sy_lo + 1;                 // 1 .. SROWS
    return q;
  };
  // stage 1: the source rows of a unit -> LDS by LDS-DMA (no register destination: nothing downstream waits for
  // them until the explicit vmcnt(0) at the top of the unit's iteration).  One wave-instruction fills 64 consecutive
  // dwords of the [channel][row][dword] tile; the per-lane SOURCE address does the row gather.  Rows past the
  // band's last source row re-load that row (harmless).
  auto dma_src = [&](const Unit& q) {
    const int view = q.f >= a.n ? 1 : 0;
    const uint8_t* src = a.obs[view] + (long)q.fo * C * HW;
    for (int k = wid; k < NDMA; k += NWAVE) {
      int e = k * 64 + lane;
      e = e < U8_DWORDS ? e : U8_DWORDS - 1;
      const int rowid = e / HQ, qd = e - rowid * HQ;
      const int ch = rowid / SROWS, r = rowid - ch * SROWS;
      const int rr = r < q.nsrc ? r : q.nsrc - 1;
      const unsigned* g = reinterpret_cast<const unsigned*>(src + (long)ch * HW + (long)(q.sy_lo + rr) * H) + qd;
      if constexpr (!(ABL & 8)) __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(u8w + k * 64), 4, 0, 0);
    }
  };

  const int nmine = bid < units ? (units - 1 - bid) / G + 1 : 0;
  Unit cur{};
  if (nmine > 0) {
    cur = make_unit(0);
    dma_src(cur);
  }
  // Two workgroups share a CU and run the same program: left alone they stay in lockstep (both in the VALU-bound
  // stage 2, then both in the matrix-bound stage 3: the stage times ADD).  The second half of the grid starts one
  // stage late, so that one workgroup's augmentation runs beside the other's tiles (MI355X_MICROARCH.md, "Two waves
  // that run the SAME program": stagger).  Speed only: nothing depends on it.
  // Which workgroups share a CU is the dispatcher's business: the wave slot number (HW_ID bits 3:0) tells the second
  // workgroup of a CU (slot 1 of each SIMD) from the first (slot 0).
  if (a.stagger) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    if (hwid & 1u) {
      for (int q = 0; q < a.stagger; ++q) __builtin_amdgcn_s_sleep(127);
    }
  }
  for (int k = 0; k < nmine; ++k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this unit's source rows have landed
    __syncthreads();                                           // ... for every wave; previous stage 3 has read xs

    // ---- stage 2: augmented + normalised rows -> LDS (and to global memory for the obs view).
    // On gfx950 the f32 MFMA and the f32 VALU share one datapath per SIMD (measured: with the two workgroups of a CU
    // staggered so that one's stage 2 runs beside the other's stage 3 the stage times still ADD), so every VALU
    // instruction here is matrix time lost.  Hence: a thread keeps ONE column (everything that depends on the
    // column and the x shift is computed once per unit), channels go two at a time through the packed-f32
    // instructions (v_pk_mul/add/fma_f32: two IEEE operations per lane and issue slot, same roundings), and the
    // stores use a buffer descriptor with scalar channel offsets (no 64-bit address arithmetic per element).
    if constexpr (!(ABL & 2)) {
      const bool store = cur.f < a.n_store && !(ABL & 4);
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.xaug + (long)cur.f * C * HW), 0, store ? (unsigned)(C * HW * 4) : 0u, 0x00020000);
      constexpr int RPP = NTHR / H;                            // rows per pass (the last NTHR - RPP*84 threads idle)
      const int j = tid % H, r3 = tid / H;
      if (r3 < RPP) {
        float fx;
        const float ix = coordv(j, cur.shx, fx);
        const int x0 = (int)fx;
        const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
        const bool okx0 = x0 >= 0 && x0 < S, okx1 = x0 + 1 >= 0 && x0 + 1 < S;
        const int sx0 = cl(x0), sx1 = cl(x0 + 1);
        const int xcol = (j & 1) * XPL + (j >> 1);
        for (int il = r3; il < cur.nr; il += RPP) {
          const int i = cur.i0 + il;
          float fy;
          const float iy = coordv(i, cur.shy, fy);
          const int y0 = (int)fy;
          const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
          const bool oky0 = y0 >= 0 && y0 < S, oky1 = y0 + 1 >= 0 && y0 + 1 < S;
          const int sy0 = cl(y0), sy1 = cl(y0 + 1);
          // a tap outside the padded frame gets weight +0 (t >= 0: the sum is unchanged bit for bit), see aug_rows_kernel
          const float w00 = okx0 && oky0 ? wx0 * wy0 : 0.f, w01 = okx1 && oky0 ? wx1 * wy0 : 0.f;     // nw, ne
          const float w10 = okx0 && oky1 ? wx0 * wy1 : 0.f, w11 = okx1 && oky1 ? wx1 * wy1 : 0.f;     // sw, se
          const int o00 = (sy0 - cur.sy_lo) * H + sx0, o01 = (sy0 - cur.sy_lo) * H + sx1;
          const int o10 = (sy1 - cur.sy_lo) * H + sx0, o11 = (sy1 - cur.sy_lo) * H + sx1;
          const int xo = il * XPITCH + xcol;
          const int go = (i * H + j) * 4;
          uint8_t tap[C][4];
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {                     // all 36 LDS byte reads in flight together
            const uint8_t* s8 = u8b + ch * SROWS * H;
            tap[ch][0] = s8[o00]; tap[ch][1] = s8[o01]; tap[ch][2] = s8[o10]; tap[ch][3] = s8[o11];
          }
          const f32x2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
          const f32x2 R255 = {1.0f / 255.0f, 1.0f / 255.0f}, N255 = {-255.0f, -255.0f}, MH = {-0.5f, -0.5f};
#pragma unroll
          for (int cp = 0; cp < C / 2; ++cp) {
            const int c0 = 2 * cp, c1 = 2 * cp + 1;
            // each product is rounded on its own: the empty asm keeps hipcc from contracting it into the next add
            f32x2 p0 = (f32x2){(float)tap[c0][0], (float)tap[c1][0]} * W00;
            f32x2 p1 = (f32x2){(float)tap[c0][1], (float)tap[c1][1]} * W01;
            f32x2 p2 = (f32x2){(float)tap[c0][2], (float)tap[c1][2]} * W10;
            f32x2 p3 = (f32x2){(float)tap[c0][3], (float)tap[c1][3]} * W11;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            f32x2 v = p0 + p1;
            asm volatile("" : "+v"(v));
            v = v + p2;
            asm volatile("" : "+v"(v));
            v = v + p3;
            asm volatile("" : "+v"(v));
            // v / 255 correctly rounded (div255), then - 0.5
            f32x2 q = v * R255;
            asm volatile("" : "+v"(q));
            const f32x2 e = __builtin_elementwise_fma(q, N255, v);
            q = __builtin_elementwise_fma(e, R255, q);
            asm volatile("" : "+v"(q));
            q = q + MH;
            xs[c0 * NROWS * XPITCH + xo] = q[0];
            xs[c1 * NROWS * XPITCH + xo] = q[1];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(q[0]), xrs, go, c0 * HW * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(q[1]), xrs, go, c1 * HW * 4, 0);
          }
          {
            constexpr int ch = C - 1;
            float p0 = (float)tap[ch][0] * w00, p1 = (float)tap[ch][1] * w01, p2 = (float)tap[ch][2] * w10,
                  p3 = (float)tap[ch][3] * w11;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            float v = p0 + p1;
            v = v + p2;
            v = v + p3;
            v = __fsub_rn(div255(v), 0.5f);
            xs[ch * NROWS * XPITCH + xo] = v;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), xrs, go, ch * HW * 4, 0);
          }
        }
      }
    }
    __syncthreads();                                           // xs complete; the source tile is free again

    // the next unit's source rows start their trip now and land while this unit's tiles are computed
    const Unit done = cur;
    if (k + 1 < nmine) {
      cur = make_unit(k + 1);
      dma_src(cur);
    }

    // ---- stage 3: 32-pixel tiles of the band's flattened (row, column) index, round-robin over the 4 waves
    const int npix = done.R * HO;
    const int ntiles = (npix + 31) >> 5;
    // B operands of channel pair c: element (channel 2c+half, row 2*oyl+ky, column 2*ox+kx); channel 9 (c = 4,
    // half = 1) reads channel 8 (its weight is 0).  The reads of pair c+1 are issued BEFORE the MFMAs of pair c
    // (the sched_barrier keeps hipcc from sinking them next to their use, where every group would wait out a full
    // LDS round trip -- longer still while the other workgroup's stage 2 keeps the LDS queue busy).
    auto load_group = [&](float (&dst)[9], const float* xb, int c) {
      const int ch = (c == CP - 1) ? (C - 1) : 2 * c + half;
      const float* xc = xb + ch * (NROWS * XPITCH);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        dst[ky * 3 + 0] = xc[ky * XPITCH];
        dst[ky * 3 + 1] = xc[ky * XPITCH + XPL];
        dst[ky * 3 + 2] = xc[ky * XPITCH + 1];
      }
    };
    for (int tile = wid; tile < ((ABL & 1) || wid >= 4 ? 0 : ntiles); tile += 4) {
      const int p0 = tile * 32 + col;
      const int p = p0 < npix ? p0 : npix - 1;
      const int oyl = p / HO, ox = p - oyl * HO;
      const float* xb = xs + (2 * oyl) * XPITCH + ox;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = breg[r];
      if constexpr (BF) {
        // slot j of this lane: channel 8*half + j; the zero-weight slots of the upper half re-read channel 8
        const float* xj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xj[j] = xb + (half ? C - 1 : j) * (NROWS * XPITCH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t - 3 * ky;
          const int off = ky * XPITCH + (kx == 1 ? XPL : 0) + (kx == 2 ? 1 : 0);
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = xj[j][off];
          const u32x4_c1 pk = {pack_bf16_c1(v[0], v[1]), pack_bf16_c1(v[2], v[3]), pack_bf16_c1(v[4], v[5]),
                               pack_bf16_c1(v[6], v[7])};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfb[t], __builtin_bit_cast(bf16x8_c1, pk), acc, 0, 0, 0);
        }
      } else {
      float X[2][9];
      load_group(X[0], xb, 0);
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        // the next pair's reads go out right behind this pair's first MFMA: they are then the youngest LDS
        // operations when the next pair needs them (an exact lgkmcnt(0)) and have had eight MFMAs to land
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[c * 9], X[c & 1][0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < CP) load_group(X[(c + 1) & 1], xb, c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < 9; ++t)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[c * 9 + t], X[c & 1][t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      const int oy = done.r0 + oyl;
      if constexpr (YN) {      // channels 8g + 4*half + 0..3 of the pixel: 8 bytes at 16g + 8*half
        const int yo = p0 < npix ? (((done.f * HO + oy) * HO + ox) * 16 + half * 2) * 4 : (int)0x80000000u;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * g4 + e] > 0.f ? acc[4 * g4 + e] : 0.f;
          const u32x2_c1 o = {pack_bf16_c1(v[0], v[1]), pack_bf16_c1(v[2], v[3])};
          __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, yo, g4 * 16, 0);
        }
        continue;
      }
      const int yoff = p0 < npix ? (((done.f * 32 + 4 * half) * HO + oy) * HO + ox) * 4 : (int)0x80000000u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r] > 0.f ? acc[r] : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, yoff, ((r & 3) + 8 * (r >> 2)) * PO * 4, 0);
      }
    }
  }
#ifdef DRQ_DEV
  if (a.stamps && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    a.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace

#ifdef DRQ_DEV
extern "C" DRQ_API void drq_dev_conv1aug_variant(int v) { g_conv1aug_variant = v; }
extern "C" DRQ_API void drq_dev_conv1aug_stagger(int v) { g_conv1aug_stagger = v; }
extern "C" DRQ_API void drq_dev_conv1aug_stamps(void* p) { g_conv1aug_stamps = (unsigned long long*)p; }
#endif

// C ABI (include/drqv2_hip.h): both views of the update through aug + conv1 in one launch.
//   y [2n][32][41][41] = relu(conv1(aug(view)/255 - 0.5));  xaug [2n][9][84][84]: frames [0, n_store) are written.
// bf_mma != 0: the layer's products on the bf16 MFMA (the bf16 update path; internal and drq_conv1_aug_fwd_bf16);
// bf_mma == 2: y is bf16 [2n][41][41][32] (drq_conv1_aug_fwd_bf16_nhwc)
int drq_conv1_aug_fwd_any(int bf_mma, const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                          const float* base_grid, const float* w, const float* bias, float* xaug, float* y, int n,
                          int n_store, hipStream_t st, const float* const* wino_w, float* wino_u, const long* fidx0,
                          const long* fidx1) {
  if (!obs || !shift || !obs1 || !shift1 || !base_grid || !w || !bias || !y || n <= 0 || n_store < 0 || n_store > 2 * n)
    return DRQ_EARG;
  if (n_store > 0 && !xaug) return DRQ_EARG;
  if (((uintptr_t)obs & 3) || ((uintptr_t)obs1 & 3)) return DRQ_EARG;     // rows are read as dwords
  const size_t yb = (size_t)2 * n * 32 * PO * (bf_mma == 2 ? 2 : 4);      // 2: bf16 [2n][41][41][32]
  if (yb >= (1ull << 31) || (bf_mma == 2 && ((uintptr_t)y & 15))) return DRQ_EARG;
  Conv1AugArgs a{};
  a.obs[0] = obs; a.obs[1] = obs1;
  a.fidx[0] = fidx0; a.fidx[1] = fidx1;
  a.shift[0] = shift; a.shift[1] = shift1;
  a.base = base_grid; a.w = w; a.bias = bias; a.xaug = xaug; a.y = y;
  a.n = n; a.n_store = n_store; a.y_bytes = (unsigned)yb; a.stagger = 1; a.stamps = nullptr;
  if (wino_w && wino_u) {
    for (int l = 0; l < 3; ++l) a.wino_w[l] = wino_w[l];
    a.wino_u = wino_u;
    a.n_rider = 6;
  }
#ifdef DRQ_DEV
  a.stagger = g_conv1aug_stagger;
  a.stamps = g_conv1aug_stamps;
#endif
  static bool attr_set_dev[kMaxDevices] = {};
  bool& attr_set = attr_set_dev[drq_device()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv1_aug_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv1_aug_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv1_aug_kernel<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
#ifdef DRQ_DEV
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1_aug_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1_aug_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1_aug_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1_aug_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv1_aug_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#endif
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long units = (long)2 * n * NBAND;
  long blocks = units;
  const long cap = 2L * drq_num_cus();
  if (blocks > cap) blocks = cap;
  if (blocks * MAXU < units) blocks = (units + MAXU - 1) / MAXU;     // a workgroup's shift table holds MAXU units
  blocks += a.n_rider;
#ifdef DRQ_DEV
  {
    static bool said = false;
    if (!said) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)conv1_aug_kernel<0>, NTHR, LDS_BYTES);
      fprintf(stderr, "conv1_aug_kernel: %d workgroups per CU by the occupancy query (LDS %d bytes), grid %ld\n", nb,
              LDS_BYTES, blocks);
      said = true;
    }
  }
  if (bf_mma) {
    if (bf_mma == 2) hipLaunchKernelGGL((conv1_aug_kernel<0, true, true>), dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv1_aug_kernel<0, true>), dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
    DRQ_LAUNCH_CHECK();
    return DRQ_OK;
  }
  switch (g_conv1aug_variant) {
    case 1: hipLaunchKernelGGL(conv1_aug_kernel<1>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a); break;
    case 2: hipLaunchKernelGGL(conv1_aug_kernel<2>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a); break;
    case 3: hipLaunchKernelGGL(conv1_aug_kernel<3>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a); break;
    case 4: hipLaunchKernelGGL(conv1_aug_kernel<4>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a); break;
    case 10: hipLaunchKernelGGL(conv1_aug_kernel<10>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a); break;
    default: hipLaunchKernelGGL(conv1_aug_kernel<0>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
  }
#else
  if (bf_mma == 2) hipLaunchKernelGGL((conv1_aug_kernel<0, true, true>), dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
  else if (bf_mma) hipLaunchKernelGGL((conv1_aug_kernel<0, true>), dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(conv1_aug_kernel<0>, dim3((unsigned)blocks), dim3(NTHR), LDS_BYTES, st, a);
#endif
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

extern "C" {

DRQ_API int drq_conv1_aug_fwd(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                              const float* base_grid, const float* w, const float* bias, float* xaug, float* y, int n,
                              int n_store, hipStream_t st) {
  return drq_conv1_aug_fwd_any(0, obs, shift, obs1, shift1, base_grid, w, bias, xaug, y, n, n_store, st, nullptr, nullptr,
                               nullptr, nullptr);
}

DRQ_API int drq_conv1_aug_fwd_indexed(const uint8_t* frames, const long* idx, const float* shift, const uint8_t* frames1,
                                      const long* idx1, const float* shift1, const float* base_grid, const float* w,
                                      const float* bias, float* xaug, float* y, int n, int n_store, hipStream_t st) {
  if (!idx || !idx1) return DRQ_EARG;
  return drq_conv1_aug_fwd_any(0, frames, shift, frames1, shift1, base_grid, w, bias, xaug, y, n, n_store, st, nullptr,
                               nullptr, idx, idx1);
}

DRQ_API int drq_conv1_aug_fwd_bf16(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                                   const float* base_grid, const float* w, const float* bias, float* xaug, float* y,
                                   int n, int n_store, hipStream_t st) {
  return drq_conv1_aug_fwd_any(1, obs, shift, obs1, shift1, base_grid, w, bias, xaug, y, n, n_store, st, nullptr, nullptr,
                               nullptr, nullptr);
}

DRQ_API int drq_conv1_aug_fwd_bf16_nhwc(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                                        const float* base_grid, const float* w, const float* bias, float* xaug,
                                        void* y_nhwc, int n, int n_store, hipStream_t st) {
  return drq_conv1_aug_fwd_any(2, obs, shift, obs1, shift1, base_grid, w, bias, xaug, (float*)y_nhwc, n, n_store, st,
                               nullptr, nullptr, nullptr, nullptr);
}

}  // extern "C"
