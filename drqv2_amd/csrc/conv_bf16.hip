// bf16-MFMA variants of the encoder's 32->32 channel 3x3 convolutions (BASELINE configs[4]: humanoid_run, batch 2048,
// "bf16"): conv2..4 forward, their dgrad and their wgrad on v_mfma_f32_32x32x16_bf16.
//
// New functionality: the reference is fp32 only.  Semantics chosen here ("fp32 math,
// rounded"): every MFMA operand (activation, gradient, weight) is rounded to bf16 (round-to-nearest-even) when it is
// staged, products are exact, accumulation is fp32, and everything that is STORED stays fp32 in the layouts of the
// fp32 path (NCHW activations, zero-padded gradient buffers, fp32 master weights and Adam).  So a kernel's result
// equals an fp32-accumulated convolution of the bf16-rounded operands -- that is what tests/test_hip_bf16.py checks
// (against fp64 on rounded operands: ~1e-6), next to the distance from the unrounded fp32 result (~3e-3).
//
// Forward / dgrad (conv3x3_bf16_kernel): a workgroup owns one (sample, third or quarter of the output rows).  It stages the
// input rows of that third for all 32 channels into LDS as bf16 in [pixel][channel] order (80-byte pixel pitch: the
// 16-byte operand reads of 16 neighbouring pixels fall on 64 distinct banks), so that the B operand of one MFMA --
// 8 consecutive channels of one tap of one pixel -- is ONE aligned ds_read_b128.  The A operands (18 fragments: 9 taps
// x 2 channel groups) stay in registers for the whole kernel.  Each input element is read from global memory once
// per workgroup (plus the two halo rows), not once per tap: with the matrix pipe 16x faster than in fp32 the kernel
// is bound by HBM (fp32 activations in and out), not by MFMA.  dgrad is the same kernel on the zero-padded output
// gradient with the weights transposed and flipped, epilogue = ReLU mask.
//
// Wgrad (conv3x3_wgrad_bf16_kernel): D[cout][cin] (x 9 taps) += dY[cout][pixels] * X[cin][pixels + tap], k = 16
// consecutive pixels of an output row per MFMA.  A wave stages one (sample, output row) at a time into wave-private
// LDS as bf16 in [channel][x] order; a lane's A fragment is 8 consecutive pixels of its cout (one ds_read_b128), and
// the three kx taps of an input row come from ONE aligned 5-dword read (v_alignbit for the odd shift).
//
// Activation layouts.  LAY = 0: fp32 NCHW in and out (the C ABI entries; operands rounded as they are staged).  The
// update's bf16 path keeps the activations BETWEEN the encoder layers (conv1..conv3 outputs) as bf16 in
// [frame][y][x][32 channels] order, 64 bytes per pixel -- the order of the LDS image below.  The stage of the next layer
// is then a straight copy of 16-byte pieces, a lane's sixteen outputs (channels 8g + 4*half + 0..3, g = 0..3, in the
// 32x32 MFMA's C layout) are four 8-byte stores, and the ReLU mask of an input gradient four 8-byte loads.  Rounding at
// the store instead of at the stage gives the same operands in the same k order: results are identical bit for bit
// (round-to-nearest is monotone, so it commutes with ReLU and keeps the sign the mask tests); tests compare with LAY = 0.
// LAY bits: 1 = input in that layout, 2 = output in that layout, 4 = mask in that layout.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // RNE, lo in bits 15:0
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

struct ConvBfArgs {
  const float* x;      // [NB][32][HIN][HIN]
  const float* w;      // canonical [32][32][3][3]
  const float* bias;   // [32] or null
  const float* mask;   // [NB][32][HOUT][HOUT] or null : out *= (mask > 0)
  float* y;
  long y_bs, y_cs, y_rs, y_off;   // output strides (elements)
  unsigned y_bytes, mask_bytes;
  int nb;
  int relu;
  int wmode;           // 0 forward, 1 dgrad (transposed + flipped weights)
  int y_pad;           // LAY & 2: the output is the interior of a [NB][HOUT+2*y_pad]^2[32] buffer (zero border kept by the caller)
};

// A workgroup's unit is load -> barrier -> MFMAs -> stores, one after the other; the workgroups of a CU hide each
// other's phases.  With the bf16 channel-contiguous input (ten 16-byte loads per thread) three per CU measure 12 %
// faster than two; with fp32 input (77 dword loads per thread in flight, 162 registers) three are 9 % slower.
constexpr int conv_bf_wgs(int lay) { return (lay & 1) ? 3 : 2; }
constexpr int PIXP = 20;   // dwords per pixel in LDS: 16 (32 bf16 channels) + 4 pad

template <int HIN, int LAY = 0>
struct ConvBfGeom {
  static constexpr int HOUT = HIN - 2;
  static constexpr int WGS = conv_bf_wgs(LAY);
  static constexpr int NPART = (WGS == 3 && HIN >= 43) ? 4 : 3;   // parts per sample: WGS images fit a CU's 160 KB of LDS
  static constexpr int RP = (HOUT + NPART - 1) / NPART;           // output rows per part
  static constexpr int NIN = (RP + 2) * HIN;                      // staged pixels per part
  static constexpr int LDS_DWORDS = NIN * PIXP > 18 * 64 * 4 ? NIN * PIXP : 18 * 64 * 4;
};

template <int HIN, bool MASK, int LAY = 0>
__global__ __launch_bounds__(256, conv_bf_wgs(LAY)) void conv3x3_bf16_kernel(ConvBfArgs a) {
  constexpr bool IN_NHWC = (LAY & 1) != 0, OUT_NHWC = (LAY & 2) != 0, MASK_NHWC = (LAY & 4) != 0;
  using G = ConvBfGeom<HIN, LAY>;
  constexpr int HOUT = G::HOUT, RP = G::RP, P = HOUT * HOUT, NPART = G::NPART;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, half = lane >> 5;

  // ---- A operands: 18 fragments (step s = tap*2 + channel group) -> registers, through LDS once
  {
    unsigned short* wl = reinterpret_cast<unsigned short*>(smem_u);      // [18][64][8] bf16
    for (int idx = tid; idx < 32 * 32 * 9; idx += 256) {
      const float v = a.w[idx];
      const int t = idx % 9, q = idx / 9;
      const int co = q >> 5, ci = q & 31;
      int row, kc, tt;
      if (a.wmode == 0) { row = co; kc = ci; tt = t; }        // A[cout][k = cin] at tap t
      else              { row = ci; kc = co; tt = 8 - t; }    // A[cin][k = cout] at the flipped tap
      const int s = tt * 2 + (kc >> 4), ln = ((kc >> 3) & 1) * 32 + row;
      const __bf16 bv = (__bf16)v;
      wl[(s * 64 + ln) * 8 + (kc & 7)] = __builtin_bit_cast(unsigned short, bv);
    }
  }
  __syncthreads();
  bf16x8 wf[18];
#pragma unroll
  for (int s = 0; s < 18; ++s)
    wf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem_u + (s * 64 + lane) * 4));
  float breg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) breg[r] = a.bias ? a.bias[(r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, 0, MASK ? a.mask_bytes : 0u, 0x00020000);
  const int ycs4 = (int)a.y_cs * 4;

  const int units = a.nb * NPART;
  // XCD-aware order (workgroups go to the XCDs round-robin by blockIdx; speed only): each XCD takes a contiguous eighth
  // of every round of units, so the thirds of one sample -- which share their two halo rows -- meet in one L2
  const int G8 = (int)gridDim.x;
  const int lb = (G8 & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (G8 >> 3) + ((int)blockIdx.x >> 3);
  for (int u = lb; u < units; u += gridDim.x) {
    const int b = u / NPART, part = u - b * NPART;
    const int r0 = part * RP;
    const int R = (r0 + RP <= HOUT ? RP : HOUT - r0);
    const int nin = (R + 2) * HIN;
    const float* src = a.x + ((long)b * 32 * HIN + r0) * HIN;      // channel 0, row r0
    __syncthreads();                                               // weights read / previous unit's tiles done
    if constexpr (IN_NHWC) {
      // ---- stage: the part's rows are nin * 64 contiguous bytes in exactly the image's order: 16-byte pieces
      const u32x4* s4 = reinterpret_cast<const u32x4*>(a.x) + ((long)b * HIN + r0) * HIN * 4;
      for (int i = tid; i < nin * 4; i += 256)
        *reinterpret_cast<u32x4*>(smem_u + (i >> 2) * PIXP + (i & 3) * 4) = s4[i];
    } else
    // ---- stage: pixels [0, nin) of every channel (contiguous per channel) -> bf16 [pixel][channel]
    for (int q0 = 0; q0 < nin; q0 += 256) {
      const int q = q0 + tid;
      const int qc = q < nin ? q : nin - 1;
      float v[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = src[(long)c * HIN * HIN + qc];
      if (q < nin) {
        unsigned* d = smem_u + q * PIXP;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = pack_bf16(v[c4 * 8 + 2 * e], v[c4 * 8 + 2 * e + 1]);
          *reinterpret_cast<u32x4*>(d + c4 * 4) = pk;
        }
      }
    }
    __syncthreads();

    // ---- tiles of 32 consecutive pixels of the part's flattened (row, column) index
    const int npix = R * HOUT;
    const int ntiles = (npix + 31) >> 5;
    for (int tile = wid; tile < ntiles; tile += 4) {
      const int p0 = tile * 32 + col;
      const int p = p0 < npix ? p0 : npix - 1;
      const int oyl = p / HOUT, ox = p - oyl * HOUT;
      const unsigned* xb = smem_u + (oyl * HIN + ox) * PIXP + half * 4;      // channels 8*half.. of group 0
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = breg[r];
      float mv[16];
      const int oy = r0 + oyl;
      if constexpr (MASK && MASK_NHWC) {       // the lane's channels 8g + 4*half + 0..3: 8 bytes at 16g + 8*half of the pixel
        const int moff = ((b * P + oy * HOUT + ox) * 16 + half * 2) * 4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const u32x2b m = __builtin_amdgcn_raw_buffer_load_b64(mrsrc, moff, g4 * 16, 0);
          const unsigned w0 = m[0], w1 = m[1];
          mv[4 * g4 + 0] = __uint_as_float(w0 << 16);
          mv[4 * g4 + 1] = __uint_as_float(w0 & 0xffff0000u);
          mv[4 * g4 + 2] = __uint_as_float(w1 << 16);
          mv[4 * g4 + 3] = __uint_as_float(w1 & 0xffff0000u);
        }
      } else if constexpr (MASK) {
        const int moff = ((b * 32 + 4 * half) * P + oy * HOUT + ox) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrsrc, moff, ((r & 3) + 8 * (r >> 2)) * P * 4, 0));
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const bf16x8 xv =
              __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + (ky * HIN + kx) * PIXP + g * 8));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t * 2 + g], xv, acc, 0, 0, 0);
        }
      }
      if constexpr (OUT_NHWC) {                // channels 8g + 4*half + 0..3: 8 bytes at 16g + 8*half of the pixel
        const int hp = HOUT + 2 * a.y_pad;
        const int yoff = p0 < npix ? (((b * hp + oy + a.y_pad) * hp + ox + a.y_pad) * 16 + half * 2) * 4 : (int)0x80000000u;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[4 * g4 + e];
            if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
            if constexpr (MASK) v[e] = mv[4 * g4 + e] > 0.f ? v[e] : 0.f;
          }
          const u32x2b o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, yoff, g4 * 16, 0);
        }
        continue;
      }
      const int yoff = p0 < npix
                           ? ((int)a.y_off + b * (int)a.y_bs + oy * (int)a.y_rs + ox) * 4 + half * 4 * ycs4
                           : (int)0x80000000u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        if constexpr (MASK) v = mv[r] > 0.f ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, yoff, ((r & 3) + 8 * (r >> 2)) * ycs4, 0);
      }
    }
  }
}

template <int HIN, bool MASK, int LAY = 0>
int launch_conv_bf16(const ConvBfArgs& a, hipStream_t st) {
  using G = ConvBfGeom<HIN, LAY>;
  constexpr int lds = G::LDS_DWORDS * 4;
  static_assert(G::WGS * lds <= 160 * 1024, "workgroups per CU");
  static bool attr_dev[kMaxDevices] = {};
  bool& attr = attr_dev[drq_device()];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<HIN, MASK, LAY>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  long blocks = (long)a.nb * G::NPART;
  const long cap = (long)G::WGS * drq_num_cus();
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((conv3x3_bf16_kernel<HIN, MASK, LAY>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
struct WgradBfArgs {
  const float* x;     // layer input  [NB][32][HIN][HIN]
  const float* dy;    // grad of the pre-activation, addressed with strides (zero-padded buffer of the fp32 path)
  long dy_bs, dy_cs, dy_rs, dy_off;
  float* part;        // [nblocks][PART]
  int nb;
};

constexpr int WG_PART = 9 * 1024 + 64;   // floats per partial record: same layout as the fp32 kernels'

constexpr int pitch4(int need) {          // smallest dword pitch >= need with pitch/4 odd: 16-byte aligned rows whose
  int p = (need + 3) / 4;                 // b128 reads by 16 neighbouring rows fall on 64 distinct banks
  if (p % 2 == 0) ++p;
  return 4 * p;
}

template <int HIN>
struct WgradBfGeom {
  static constexpr int HOUT = HIN - 2;
  static constexpr int KB = (HOUT + 15) / 16;           // k-blocks of 16 pixels per output row
  static constexpr int XROW = pitch4(8 * KB + 4);       // dwords per staged input row (pixels 0 .. 16KB+7, zero tail)
  static constexpr int DROW = pitch4(8 * KB);           // dwords per staged dY row (pixels 0 .. 16KB-1, zero tail)
  static constexpr int XPAIRS = (HIN + 1) / 2, DPAIRS = (HOUT + 1) / 2;
  static constexpr int XIT = (32 * XPAIRS + 63) / 64;   // staging instructions (pairs of pixels) per input row set
  static constexpr int DIT = (32 * DPAIRS + 63) / 64;
  static constexpr int WAVE_DWORDS = 3 * 32 * XROW + 32 * DROW;
};

// XNHWC: the layer input is bf16 [frame][y][x][32 channels] (see the head of the file); DYNHWC: dy is the bf16
// [frame][HOUT+4][HOUT+4][32] buffer zero-padded by 2 that the input-gradient launches write (a.dy = its base; the
// strides are not used).  The bias gradient then sums the bf16 values (with fp32 dy it sums the unrounded ones).
template <int HIN, bool XNHWC = false, bool DYNHWC = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_bf16_kernel(WgradBfArgs a) {
  using G = WgradBfGeom<HIN>;
  constexpr int HOUT = G::HOUT, KB = G::KB, XROW = G::XROW, DROW = G::DROW;
  constexpr int XPAIRS = G::XPAIRS, DPAIRS = G::DPAIRS, XIT = G::XIT, DIT = G::DIT;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  unsigned* xs = smem_u + wid * G::WAVE_DWORDS;       // ring of 3 input rows: [slot][32 cin][XROW]
  unsigned* ds = xs + 3 * 32 * XROW;                  // [32 cout][DROW]
  for (int i = lane; i < G::WAVE_DWORDS; i += 64) xs[i] = 0u;      // zero tails: pad pixels of X meet dY == 0

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bs[DIT];                                       // bias gradient: this lane's pieces of the dY rows it staged
#pragma unroll
  for (int k = 0; k < DIT; ++k) bs[k] = 0.f;
  constexpr int DITN = DYNHWC ? (4 * DPAIRS + 63) / 64 : 1;
  float bsn[DITN][8];                                  // DYNHWC: item (pixel pair, piece c4) -> channels 8*c4 + j
#pragma unroll
  for (int k = 0; k < DITN; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) bsn[k][j] = 0.f;

  // (sample, output row) units split evenly over the waves of the grid; a wave walks its run in order, so
  // consecutive units of one sample share two of their three input rows: only row oy+2 is new
  const int units = a.nb * HOUT;
  const int nw = (int)gridDim.x * 4;
  const int gw = (int)blockIdx.x * 4 + wid;
  const int per = units / nw, rem = units - per * nw;
  const int u0 = gw * per + (gw < rem ? gw : rem);
  const int u1 = u0 + per + (gw < rem ? 1 : 0);

  // one input row (all 32 channels) -> ring slot: item = channel * XPAIRS + pixel pair
  auto stage_x = [&](const float* xrow, int slot) {      // xrow: channel 0 of that row
    float v0[XIT], v1[XIT];
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      int it = k * 64 + lane;
      it = it < 32 * XPAIRS ? it : 32 * XPAIRS - 1;
      const int c = it / XPAIRS, pr = it - c * XPAIRS;
      const float* p = xrow + (long)c * HIN * HIN + 2 * pr;
      v0[k] = p[0];
      v1[k] = p[2 * pr + 1 < HIN ? 1 : 0];
    }
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int it = k * 64 + lane;
      if (it < 32 * XPAIRS) {
        const int c = it / XPAIRS, pr = it - c * XPAIRS;
        xs[(slot * 32 + c) * XROW + pr] = pack_bf16(v0[k], 2 * pr + 1 < HIN ? v1[k] : 0.f);
      }
    }
  };

  // the same from the channel-contiguous layout: item = (pixel pair, 16-byte piece of eight channels); the two
  // pixels' values of a channel are packed into the dword the [channel][pixel pair] image wants (v_perm)
  constexpr int XIT2 = (4 * XPAIRS + 63) / 64;
  auto stage_x_nhwc = [&](const u32x4* xrow, int slot) {  // xrow: pixel 0 of that row (4 pieces per pixel)
    u32x4 p0[XIT2], p1[XIT2];
#pragma unroll
    for (int k = 0; k < XIT2; ++k) {
      int it = k * 64 + lane;
      it = it < 4 * XPAIRS ? it : 4 * XPAIRS - 1;
      const int pr = it >> 2, c4 = it & 3;
      p0[k] = xrow[(2 * pr) * 4 + c4];
      p1[k] = xrow[(2 * pr + 1 < HIN ? 2 * pr + 1 : 2 * pr) * 4 + c4];
    }
#pragma unroll
    for (int k = 0; k < XIT2; ++k) {
      const int it = k * 64 + lane;
      if (it < 4 * XPAIRS) {
        const int pr = it >> 2, c4 = it & 3;
        const bool odd_ok = 2 * pr + 1 < HIN;
        unsigned* dst = xs + (slot * 32 + 8 * c4) * XROW + pr;          // channels 8*c4 + j, j = 0..7
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned a0 = p0[k][e], a1 = odd_ok ? p1[k][e] : 0u;
          const unsigned lo = __builtin_amdgcn_perm(a1, a0, 0x05040100u);   // (a0.lo16, a1.lo16)
          const unsigned hi = __builtin_amdgcn_perm(a1, a0, 0x07060302u);   // (a0.hi16, a1.hi16)
          dst[(2 * e) * XROW] = lo;
          dst[(2 * e + 1) * XROW] = hi;
        }
      }
    }
  };

  int prev_b = -1, prev_oy = -2;
  for (int u = u0; u < u1; ++u) {
    const int b = u / HOUT, oy = u - b * HOUT;
    const float* xsrc = a.x + (long)b * 32 * HIN * HIN;
    const u32x4* xsrc4 = reinterpret_cast<const u32x4*>(a.x) + (long)b * HIN * HIN * 4;
    // (single wave: its LDS operations complete in order, no barrier needed)
    if (b == prev_b && oy == prev_oy + 1) {
      if constexpr (XNHWC) stage_x_nhwc(xsrc4 + (long)(oy + 2) * HIN * 4, (oy + 2) % 3);
      else stage_x(xsrc + (long)(oy + 2) * HIN, (oy + 2) % 3);
    } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        if constexpr (XNHWC) stage_x_nhwc(xsrc4 + (long)(oy + ky) * HIN * 4, (oy + ky) % 3);
        else stage_x(xsrc + (long)(oy + ky) * HIN, (oy + ky) % 3);
      }
    }
    prev_b = b; prev_oy = oy;
    if constexpr (DYNHWC) {
      // item = (pixel pair, 16-byte piece of eight channels), as for x; DIT2 * 64 >= 4 * DPAIRS items
      constexpr int HP = HOUT + 4;
      constexpr int DIT2 = (4 * DPAIRS + 63) / 64;
      static_assert(DIT2 <= DIT, "bias pieces");
      const u32x4* drow = reinterpret_cast<const u32x4*>(a.dy) + (((long)b * HP + oy + 2) * HP + 2) * 4;
      u32x4 p0[DIT2], p1[DIT2];
#pragma unroll
      for (int k = 0; k < DIT2; ++k) {
        int it = k * 64 + lane;
        it = it < 4 * DPAIRS ? it : 4 * DPAIRS - 1;
        const int pr = it >> 2, c4 = it & 3;
        p0[k] = drow[(2 * pr) * 4 + c4];
        p1[k] = drow[(2 * pr + 1) * 4 + c4];          // pixel HOUT of the last pair is the zero border
      }
#pragma unroll
      for (int k = 0; k < DIT2; ++k) {
        const int it = k * 64 + lane;
        if (it < 4 * DPAIRS) {
          const int pr = it >> 2, c4 = it & 3;
          unsigned* dst = ds + (8 * c4) * DROW + pr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned a0 = p0[k][e], a1 = p1[k][e];
            dst[(2 * e) * DROW] = __builtin_amdgcn_perm(a1, a0, 0x05040100u);
            dst[(2 * e + 1) * DROW] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
            bsn[k][2 * e] += __uint_as_float(a0 << 16) + __uint_as_float(a1 << 16);
            bsn[k][2 * e + 1] += __uint_as_float(a0 & 0xffff0000u) + __uint_as_float(a1 & 0xffff0000u);
          }
        }
      }
    } else {
      const float* dsrc = a.dy + a.dy_off + (long)b * a.dy_bs + (long)oy * a.dy_rs;
      float v0[DIT], v1[DIT];
#pragma unroll
      for (int k = 0; k < DIT; ++k) {
        int it = k * 64 + lane;
        it = it < 32 * DPAIRS ? it : 32 * DPAIRS - 1;
        const int c = it / DPAIRS, pr = it - c * DPAIRS;
        const float* p = dsrc + (long)c * a.dy_cs + 2 * pr;
        v0[k] = p[0];
        v1[k] = p[2 * pr + 1 < HOUT ? 1 : 0];
      }
#pragma unroll
      for (int k = 0; k < DIT; ++k) {
        const int it = k * 64 + lane;
        if (it < 32 * DPAIRS) {
          const int c = it / DPAIRS, pr = it - c * DPAIRS;
          const float w1 = 2 * pr + 1 < HOUT ? v1[k] : 0.f;
          ds[c * DROW + pr] = pack_bf16(v0[k], w1);
          bs[k] += v0[k] + w1;                          // exact fp32 values (not the bf16-rounded ones)
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      // A: dY[cout = col][pixels 16kb + 8half .. +7]
      const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ds + col * DROW + kb * 8 + half * 4));
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // B: X[cin = col][row oy+ky][pixels 16kb + 8half + kx .. +7]: dwords d0..d4 hold pixels 16kb+8half .. +9
        const unsigned* xp = xs + (((oy + ky) % 3) * 32 + col) * XROW + kb * 8 + half * 4;
        const u32x4 d = *reinterpret_cast<const u32x4*>(xp);
        const unsigned d4 = xp[4];
        const u32x4 s1 = {__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                          __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d4, d[3], 16)};
        const u32x4 s2 = {d[1], d[2], d[3], d4};
        acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, d), acc[ky * 3 + 0], 0, 0, 0);
        acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s1), acc[ky * 3 + 1], 0, 0, 0);
        acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s2), acc[ky * 3 + 2], 0, 0, 0);
      }
    }
  }

  // ---- bias gradient of this wave: lane pieces -> per-channel sums through the (now idle) wave-private LDS.
  // Item k*64+lane belongs to channel (k*64+lane)/DPAIRS; fixed order: deterministic.
  float* fl = reinterpret_cast<float*>(xs);
  float bsum = 0.f;
  if constexpr (DYNHWC) {
#pragma unroll
    for (int k = 0; k < DITN; ++k)
      if (k * 64 + lane < 4 * DPAIRS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) fl[(k * 64 + lane) * 8 + j] = bsn[k][j];
      }
    if (lane < 32) {
      for (int pr = 0; pr < DPAIRS; ++pr) bsum += fl[(pr * 4 + (lane >> 3)) * 8 + (lane & 7)];
    }
  } else {
#pragma unroll
    for (int k = 0; k < DIT; ++k) fl[k * 64 + lane] = bs[k];
    if (lane < 32) {
      for (int pr = 0; pr < DPAIRS; ++pr) bsum += fl[lane * DPAIRS + pr];
    }
  }

  // ---- reduce the 4 waves of the block through LDS, one partial record per block (layout of the fp32 kernels:
  // [tap][reg][lane] then 64 bias pieces, db[c] = piece[c] + piece[c+32]; conv3x3_wgrad_reduce_multi_kernel turns
  // the records of all blocks into dW / db)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem_u);      // [4][WG_PART]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid * WG_PART + t * 1024 + r * 64 + lane] = acc[t][r];
  red[wid * WG_PART + 9 * 1024 + lane] = bsum;
  __syncthreads();
  float* out = a.part + (long)blockIdx.x * WG_PART;
  for (int i = threadIdx.x; i < WG_PART; i += 256)
    out[i] = (red[i] + red[WG_PART + i]) + (red[2 * WG_PART + i] + red[3 * WG_PART + i]);
}

template <int HIN, bool XNHWC = false, bool DYNHWC = false>
int launch_wgrad_bf16(const WgradBfArgs& a0, float* ws, size_t ws_bytes, int* nblocks_out, hipStream_t st) {
  using G = WgradBfGeom<HIN>;
  constexpr int lds_dwords = 4 * G::WAVE_DWORDS > 4 * WG_PART ? 4 * G::WAVE_DWORDS : 4 * WG_PART;
  static_assert(lds_dwords * 4 <= 160 * 1024, "LDS");
  const long units = (long)a0.nb * G::HOUT;
  long blocks = drq_num_cus();
  if (blocks * 4 > units) blocks = (units + 3) / 4;
  if (blocks < 1) blocks = 1;
  if ((size_t)blocks * WG_PART * sizeof(float) > ws_bytes) return DRQ_EWS;
  WgradBfArgs a = a0;
  a.part = ws;
  static bool attr_dev[kMaxDevices] = {};
  bool& attr = attr_dev[drq_device()];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel<HIN, XNHWC, DYNHWC>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_dwords * 4);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<HIN, XNHWC, DYNHWC>), dim3((unsigned)blocks), dim3(256), lds_dwords * 4, st, a);
  DRQ_LAUNCH_CHECK();
  if (nblocks_out) *nblocks_out = (int)blocks;
  return DRQ_OK;
}

}  // namespace

// conv.hip (internal): fixed-order reduction of the partial records of up to four layers
int drq_conv3x3_wgrad_reduce_multi(int n, const float* const* part, const int* nblocks, const int* cin,
                                   float* const* dw, float* const* db, hipStream_t st);

// ---- internal (step.hip) and C ABI: the same launches with the bf16 channel-contiguous layout on some operands -----
// x: bf16 [nb][hin][hin][32] when lay & 1, else fp32 NCHW; y: bf16 [nb][hout][hout][32] when lay & 2 (the strides are
// ignored), else fp32 with the given strides
int drq_conv3x3_fwd_bf16_lay(const void* x, const float* w, const float* bias, void* y, int nb, int hin, int relu,
                             long y_bs, long y_cs, long y_rs, long y_off, int lay, hipStream_t st) {
  if (!x || !w || !y || nb <= 0 || (lay & ~3)) return DRQ_EARG;
  const int hout = hin - 2;
  if (lay & 2) { y_bs = 16L * hout * hout; y_cs = 0; y_rs = 0; y_off = 0; }   // 64 bytes per pixel, in floats
  const size_t yb = (size_t)nb * y_bs * 4;
  if (yb >= (1ull << 31) || y_off < 0 || y_bs <= 0 || ((lay & 1) && ((uintptr_t)x & 15)) || ((lay & 2) && ((uintptr_t)y & 15)))
    return DRQ_EARG;
  ConvBfArgs a{(const float*)x, w, bias, nullptr, (float*)y, y_bs, y_cs, y_rs, y_off, (unsigned)yb, 0u, nb, relu, 0};
#define DRQ_BF_FWD(H)                                            \
  if (hin == H) {                                                \
    if (lay == 0) return launch_conv_bf16<H, false, 0>(a, st);   \
    if (lay == 1) return launch_conv_bf16<H, false, 1>(a, st);   \
    if (lay == 2) return launch_conv_bf16<H, false, 2>(a, st);   \
    return launch_conv_bf16<H, false, 3>(a, st);                 \
  }
  DRQ_BF_FWD(41) DRQ_BF_FWD(39) DRQ_BF_FWD(37)
#undef DRQ_BF_FWD
  return DRQ_EARG;
}

// lay bits: 1 = dy_pad is bf16 [nb][hout+4][hout+4][32] (zero border), 2 = dx is the interior of a bf16
// [nb][hout+6][hout+6][32] buffer padded by 2 (the strides are ignored; the border is the caller's), 4 = the mask is bf16
// [nb][hout+2][hout+2][32]; a clear bit: fp32 NCHW as in drq_conv3x3_dgrad_bf16
int drq_conv3x3_dgrad_bf16_lay(const void* dy_pad, const float* w, const void* mask, void* dx, int nb, int hout,
                               long dx_bs, long dx_cs, long dx_rs, long dx_off, int lay, hipStream_t st) {
  if (!dy_pad || !w || !dx || !mask || nb <= 0 || (lay & ~7)) return DRQ_EARG;
  const int hp = hout + 4, hin = hout + 2;
  if (lay & 2) { dx_bs = 16L * (hin + 4) * (hin + 4); dx_cs = 0; dx_rs = 0; dx_off = 0; }
  const size_t yb = (size_t)nb * dx_bs * 4;
  const size_t mb = (size_t)nb * 32 * hin * hin * ((lay & 4) ? 2 : 4);
  if (yb >= (1ull << 31) || mb >= (1ull << 31) || dx_off < 0 || dx_bs <= 0) return DRQ_EARG;
  if (((lay & 4) && ((uintptr_t)mask & 15)) || ((lay & 1) && ((uintptr_t)dy_pad & 15)) || ((lay & 2) && ((uintptr_t)dx & 15)))
    return DRQ_EARG;
  ConvBfArgs a{(const float*)dy_pad, w, nullptr, (const float*)mask, (float*)dx, dx_bs, dx_cs, dx_rs, dx_off, (unsigned)yb,
               (unsigned)mb, nb, 0, 1, (lay & 2) ? 2 : 0};
#define DRQ_BF_DGRAD(H)                                         \
  if (hp == H) {                                                \
    switch (lay) {                                              \
      case 0: return launch_conv_bf16<H, true, 0>(a, st);       \
      case 4: return launch_conv_bf16<H, true, 4>(a, st);       \
      case 5: return launch_conv_bf16<H, true, 5>(a, st);       \
      case 6: return launch_conv_bf16<H, true, 6>(a, st);       \
      case 7: return launch_conv_bf16<H, true, 7>(a, st);       \
      default: return DRQ_EARG;                                 \
    }                                                           \
  }
  DRQ_BF_DGRAD(39) DRQ_BF_DGRAD(41) DRQ_BF_DGRAD(43)
#undef DRQ_BF_DGRAD
  return DRQ_EARG;
}

// ---- C ABI (include/drqv2_hip.h), argument meaning as the fp32 entries of conv.hip ------------------------------
extern "C" {

DRQ_API int drq_conv3x3_fwd_bf16_nhwc(const void* x, const float* w, const float* bias, void* y, int nb, int hin, int relu,
                                      int x_nhwc, int y_nhwc, hipStream_t st) {
  const int hout = hin - 2;
  return drq_conv3x3_fwd_bf16_lay(x, w, bias, y, nb, hin, relu, 32L * hout * hout, (long)hout * hout, hout, 0,
                                  (x_nhwc ? 1 : 0) | (y_nhwc ? 2 : 0), st);
}

DRQ_API int drq_conv3x3_dgrad_bf16_nhwc(const void* dy_pad, const float* w, const void* mask_nhwc, void* dx, int nb, int hout,
                                        int dy_nhwc, int dx_nhwc, long dx_bs, long dx_cs, long dx_rs, long dx_off,
                                        hipStream_t st) {
  return drq_conv3x3_dgrad_bf16_lay(dy_pad, w, mask_nhwc, dx, nb, hout, dx_bs, dx_cs, dx_rs, dx_off,
                                    4 | (dy_nhwc ? 1 : 0) | (dx_nhwc ? 2 : 0), st);
}

DRQ_API int drq_conv3x3_fwd_bf16(const float* x, const float* w, const float* bias, float* y, int nb, int hin, int relu, long y_bs,
                         long y_cs, long y_rs, long y_off, hipStream_t st) {
  return drq_conv3x3_fwd_bf16_lay(x, w, bias, y, nb, hin, relu, y_bs, y_cs, y_rs, y_off, 0, st);
}

DRQ_API int drq_conv3x3_dgrad_bf16(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout, long dx_bs,
                           long dx_cs, long dx_rs, long dx_off, hipStream_t st) {
  return drq_conv3x3_dgrad_bf16_lay(dy_pad, w, mask, dx, nb, hout, dx_bs, dx_cs, dx_rs, dx_off, 0, st);
}

}  // extern "C"

// internal (step.hip): partial records only; one reduction launch serves all layers
// lay bits: 1 = x is bf16 [nb][hin][hin][32] instead of fp32 NCHW; 2 = dy is the bf16 [nb][hin+2][hin+2][32] buffer
// zero-padded by 2 (its base; the strides are ignored) instead of an fp32 strided view
int drq_conv3x3_wgrad_partial_bf16_lay(const void* x, const void* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                       long dy_off, float* part, size_t part_bytes, int* nblocks, int lay,
                                       hipStream_t st) {
  if (!x || !dy || !part || !nblocks || nb <= 0 || (lay & ~3)) return DRQ_EARG;
  if (!(lay & 2) && (dy_off < 0 || dy_bs <= 0)) return DRQ_EARG;
  if (((size_t)part & 15) != 0 || ((lay & 1) && ((uintptr_t)x & 15)) || ((lay & 2) && ((uintptr_t)dy & 15))) return DRQ_EARG;
  if (lay == 2) return DRQ_EARG;                        // not instantiated: the update never has fp32 x beside bf16 dy
  WgradBfArgs a{(const float*)x, (const float*)dy, dy_bs, dy_cs, dy_rs, dy_off, nullptr, nb};
#define DRQ_BF_WGRAD(H)                                                                  \
  if (hin == H) {                                                                        \
    if (lay == 3) return launch_wgrad_bf16<H, true, true>(a, part, part_bytes, nblocks, st);  \
    if (lay == 1) return launch_wgrad_bf16<H, true, false>(a, part, part_bytes, nblocks, st); \
    return launch_wgrad_bf16<H>(a, part, part_bytes, nblocks, st);                       \
  }
  DRQ_BF_WGRAD(41) DRQ_BF_WGRAD(39) DRQ_BF_WGRAD(37)
#undef DRQ_BF_WGRAD
  return DRQ_EARG;
}

int drq_conv3x3_wgrad_partial_bf16(const float* x, const float* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                   long dy_off, float* part, size_t part_bytes, int* nblocks, hipStream_t st) {
  return drq_conv3x3_wgrad_partial_bf16_lay(x, dy, nb, hin, dy_bs, dy_cs, dy_rs, dy_off, part, part_bytes, nblocks, 0, st);
}

extern "C" DRQ_API int drq_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw, float* db, int nb, int hin,
                                              long dy_bs, long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes,
                                              hipStream_t st) {
  if (!dw || !db || !ws) return DRQ_EARG;
  int nblk = 0;
  const int rc = drq_conv3x3_wgrad_partial_bf16(x, dy, nb, hin, dy_bs, dy_cs, dy_rs, dy_off, ws, ws_bytes, &nblk, st);
  if (rc != DRQ_OK) return rc;
  const float* parts[1] = {ws};
  const int cins[1] = {32};
  float *dws[1] = {dw}, *dbs[1] = {db};
  return drq_conv3x3_wgrad_reduce_multi(1, parts, &nblk, cins, dws, dbs, st);
}

extern "C" DRQ_API int drq_conv3x3_wgrad_bf16_nhwc(const void* x_nhwc, const void* dy, float* dw, float* db, int nb, int hin,
                                                   int dy_nhwc, long dy_bs, long dy_cs, long dy_rs, long dy_off, float* ws,
                                                   size_t ws_bytes, hipStream_t st) {
  if (!dw || !db || !ws) return DRQ_EARG;
  int nblk = 0;
  const int rc = drq_conv3x3_wgrad_partial_bf16_lay(x_nhwc, dy, nb, hin, dy_bs, dy_cs, dy_rs, dy_off, ws, ws_bytes, &nblk,
                                                    dy_nhwc ? 3 : 1, st);
  if (rc != DRQ_OK) return rc;
  const float* parts[1] = {ws};
  const int cins[1] = {32};
  float *dws[1] = {dw}, *dbs[1] = {db};
  return drq_conv3x3_wgrad_reduce_multi(1, parts, &nblk, cins, dws, dbs, st);
}
