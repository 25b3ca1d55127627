// Weight gradient of the encoder's 32->32 layers in Winograd F(2x2,3x3) form on the f32 matrix cores.
//
// Reference op: the autograd weight / bias gradient of nn.Conv2d(32,32,3,1) (drqv2.py:56-58).  With
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 2x2 output tile (conv_wino.hip),
// the gradient of the loss with respect to g is
//   dg = G^T [ sum over samples and tiles of (A dY A^T) .* (B^T d B) ] G      per (cout, cin):
// 16 positions x a 32x32 (cout x cin) matrix, reduced over all tiles -- 16 tile-GEMMs instead of the direct form's
// 9 pixel-GEMMs over four times as many pixels: 2.25x fewer matrix FLOPs.
//
// Mapping: v_mfma_f32_16x16x4_f32 with K = 4 tiles: D[cout 16][cin 16] += dM[cout][tile] * V[tile][cin].  Lane l holds
// channel l&15 (+16 for the second half) and tile 4s + (l>>4) of step s of a tile row -- in BOTH roles: it loads that
// channel's 2x2 dY patch (A operand, after A dY A^T) and that channel's 4x4 input patch (B operand, after B^T d B)
// straight from global memory (the patches of the four tiles of a step are 40 contiguous bytes of a row: L2 serves the
// re-touched lines).  A wave takes steps of four consecutive tiles round-robin and keeps all 16 positions x 2x2 channel-half quadrants x 4 registers =
// 256 accumulators (one wave per SIMD: the unified register file holds them as AGPRs).  At the end every wave maps its
// sums back to 3x3 taps (G^T . G), the four waves of a workgroup are added through LDS in wave order and the workgroup
// writes ONE partial record in the format of conv3x3_wgrad3_kernel (conv.hip): the same fixed-order reduction kernel
// finishes the job (deterministic, no float atomics).
#include "common.h"

namespace {

struct WWArgs {
  const float* x;     // layer input  [NB][32][HIN][HIN]
  const float* dy;    // grad of the pre-activation, zero-padded by 2, addressed with strides; dy_off = element (0,0)
  int dy_bs, dy_cs, dy_rs, dy_off;
  float* part;        // [nblocks][9*1024 + 64]
  unsigned x_bytes, dy_bytes;
  int nb;
};

typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned u32x2q __attribute__((ext_vector_type(2)));

#ifdef DRQ_DEV
int g_ww_variant = 0;   // drq_dev_wgrad_wino_variant: timing ablations (tools/wino_ab.py)
// drq_dev_wgrad_wino_stamps: 8 u64 per wave (s_memtime at start / first patches issued / loop done / image in LDS /
// after the barrier / record written, then s_memrealtime start and end); tools/wwino_stamps.py
__device__ unsigned long long* d_ww_stamps = nullptr;
#define WW_MARK(k)                                                                                                     \
  do {                                                                                                                 \
    if (d_ww_stamps && lane == 0) d_ww_stamps[((size_t)bx * 4 + wid) * 8 + (k)] =                                       \
        (k) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();                                    \
  } while (0)
#else
#define WW_MARK(k)
#endif

// ABL (development build only): 1 = no patch loads, 2 = no transforms (the raw patches are multiplied); wrong results
// bx / nblk: the workgroup's index among the nblk workgroups that share this layer (a launch of its own: blockIdx.x /
// gridDim.x; the three-layer launch below gives every layer a range of its grid)
template <int HIN, int ABL = 0>
__device__ __forceinline__ void ww_body(const WWArgs& a, int bx, int nblk) {
#pragma clang fp contract(off)
  constexpr int HOUT = HIN - 2;
  constexpr int TH = (HOUT + 1) / 2;          // tiles per row / tile rows per sample
  constexpr int PLANE = HIN * HIN * 4;        // bytes of one input channel
  constexpr int PART = 9 * 1024 + 64;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][PART]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = lane & 15, tq = lane >> 4;
  WW_MARK(6);
  WW_MARK(0);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  constexpr int kDrop = (int)0x80000000u;     // beyond num_records: the load returns 0
  const int dcs4 = a.dy_cs * 4, drs4 = a.dy_rs * 4;

  // work: steps of 4 consecutive tiles of the flattened (sample, tile row, tile column) index, dealt round-robin to
  // the waves of the grid (a step may straddle two tile rows: every lane places its own tile)
  constexpr int TT = TH * TH;
  const int ntile = a.nb * TT;
  const int nstep = (ntile + 3) >> 2;
  const int nw = nblk * 4, gw = bx * 4 + wid;
  const int total = gw < nstep ? (nstep - 1 - gw) / nw + 1 : 0;

  // per-lane byte offsets of this wave's it-th step: the x patch and the dY patch
  auto item_off = [&](int it, int& xo, int& dyo) {
    const int t = 4 * (gw + it * nw) + tq;
    const int tc = t < ntile ? t : ntile - 1;
    const int b = tc / TT, rem = tc - b * TT;
    const int ty = rem / TH, tx = rem - ty * TH;
    xo = (((b * 32 + ch) * HIN + 2 * ty) * HIN + 2 * tx) * 4;
    dyo = t < ntile ? (a.dy_off + b * a.dy_bs + ch * a.dy_cs + 2 * ty * a.dy_rs + 2 * tx) * 4 : kDrop;
  };
  // two register sets of patches: the loads of step it+2 are issued while step it multiplies (one wave per SIMD has
  // nobody else to hide an HBM miss behind: one step = 64 MFMAs = 2048 matrix cycles is not always enough)
  auto load_item = [&](float (&Pdx)[2][16], float (&Pdy)[2][4], int xo, int dyo) {
    if constexpr (ABL & 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Pdx[h][e] = 1.0f + (float)(xo + e);
#pragma unroll
        for (int e = 0; e < 4; ++e) Pdy[h][e] = 0.5f + (float)(dyo + e);
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, h * (16 * PLANE) + i * (HIN * 4), 0);
        const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
        Pdx[h][i * 4 + 0] = __uint_as_float(e0);
        Pdx[h][i * 4 + 1] = __uint_as_float(e1);
        Pdx[h][i * 4 + 2] = __uint_as_float(e2);
        Pdx[h][i * 4 + 3] = __uint_as_float(e3);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x2q v = __builtin_amdgcn_raw_buffer_load_b64(drs, dyo, h * 16 * dcs4 + i * drs4, 0);
        const unsigned e0 = v[0], e1 = v[1];
        Pdy[h][i * 2 + 0] = __uint_as_float(e0);
        Pdy[h][i * 2 + 1] = __uint_as_float(e1);
      }
    }
  };

  // the k-th of the twelve load instructions of an item: 0..7 = x rows (h = k>>2, i = k&3), 8..11 = dY rows
  auto load_one = [&](float (&Pdx)[2][16], float (&Pdy)[2][4], int xo, int dyo, int k) {
    if constexpr (ABL & 1) {
      if (k < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Pdx[k >> 2][(k & 3) * 4 + e] = 1.0f + (float)(xo + e + k);
      } else {
        Pdy[(k - 8) >> 1][((k - 8) & 1) * 2 + 0] = 0.5f + (float)(dyo + k);
        Pdy[(k - 8) >> 1][((k - 8) & 1) * 2 + 1] = 0.25f + (float)(dyo + k);
      }
      return;
    }
    if (k < 8) {
      const int h = k >> 2, i = k & 3;
      const u32x4q v = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, h * (16 * PLANE) + i * (HIN * 4), 0);
      const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
      Pdx[h][i * 4 + 0] = __uint_as_float(e0);
      Pdx[h][i * 4 + 1] = __uint_as_float(e1);
      Pdx[h][i * 4 + 2] = __uint_as_float(e2);
      Pdx[h][i * 4 + 3] = __uint_as_float(e3);
    } else {
      const int h = (k - 8) >> 1, i = (k - 8) & 1;
      const u32x2q v = __builtin_amdgcn_raw_buffer_load_b64(drs, dyo, h * 16 * dcs4 + i * drs4, 0);
      const unsigned e0 = v[0], e1 = v[1];
      Pdy[h][i * 2 + 0] = __uint_as_float(e0);
      Pdy[h][i * 2 + 1] = __uint_as_float(e1);
    }
  };

  f32x4 acc[16][2][2];                        // [position][cout half][cin half]
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) acc[p][qa][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};                 // bias gradient: sum of this lane's dY values per channel half

  // one step: transforms of the patches in P, then P is refilled with item `nit`, then the 64 MFMAs
  auto step = [&](float (&Pdx)[2][16], float (&Pdy)[2][4], int nit) {
    int nxo, ndyo;
    item_off(nit < total ? nit : total - 1, nxo, ndyo);
    float V[2][16], M[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // V = B^T d B
      float t[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0 * 4 + j] = Pdx[h][0 * 4 + j] - Pdx[h][2 * 4 + j];
        t[1 * 4 + j] = Pdx[h][1 * 4 + j] + Pdx[h][2 * 4 + j];
        t[2 * 4 + j] = Pdx[h][2 * 4 + j] - Pdx[h][1 * 4 + j];
        t[3 * 4 + j] = Pdx[h][1 * 4 + j] - Pdx[h][3 * 4 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ABL & 2) {
          V[h][i * 4 + 0] = Pdx[h][i * 4 + 0]; V[h][i * 4 + 1] = Pdx[h][i * 4 + 1];
          V[h][i * 4 + 2] = Pdx[h][i * 4 + 2]; V[h][i * 4 + 3] = Pdx[h][i * 4 + 3];
          continue;
        }
        V[h][i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
        V[h][i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
        V[h][i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
        V[h][i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
      }
      // dM' = A' dY A'^T with A' = [[1,0],[1,1],[1,-1],[0,1]]: the last row of A = [0,-1] with its sign flipped (no
      // negations here); dM = S dM' S with S = diag(1,1,1,-1), and S moves into the final transform: dg = (SG)^T dU' (SG)
      const float y00 = Pdy[h][0], y01 = Pdy[h][1], y10 = Pdy[h][2], y11 = Pdy[h][3];
      bsum[h] += (y00 + y01) + (y10 + y11);
      const float r[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {y10, y11}};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ABL & 2) {
          M[h][i * 4 + 0] = Pdy[h][0]; M[h][i * 4 + 1] = Pdy[h][1]; M[h][i * 4 + 2] = Pdy[h][2]; M[h][i * 4 + 3] = Pdy[h][3];
          continue;
        }
        M[h][i * 4 + 0] = r[i][0];
        M[h][i * 4 + 1] = r[i][0] + r[i][1];
        M[h][i * 4 + 2] = r[i][0] - r[i][1];
        M[h][i * 4 + 3] = r[i][1];
      }
    }
    // The refill's twelve loads are spread over the step's 64 MFMAs, three ahead of every sixteen: a wave issues in
    // order, so a burst of loads that backs up in the texture addresser's queue would hold its MFMAs too -- and this
    // kernel has one wave per SIMD.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
      for (int k = 3 * k4; k < 3 * k4 + 3; ++k) load_one(Pdx, Pdy, nxo, ndyo, k);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 4 * k4; p < 4 * k4 + 4; ++p)
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            acc[p][qa][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(M[qa][p], V[qb][p], acc[p][qa][qb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  float dx0[2][16], dy0[2][4];
  if (total > 0) {
    int xo, dyo;
    item_off(0, xo, dyo);
    load_item(dx0, dy0, xo, dyo);
  }
  WW_MARK(1);
  for (int it = 0; it < total; ++it) step(dx0, dy0, it + 1);
  WW_MARK(2);

  // ---- dg = G^T dU G per (cout, cin) -> this wave's image of the partial record in LDS
  // D register r of lane l: cout = 16*qa + 4*(l>>4) + r, cin = 16*qb + (l&15)
  // record element of (cout, cin, tap): tap*1024 + rr*64 + half*32 + cin with cout = (rr&3) + 8*(rr>>2) + 4*half
  float* mine = red + wid * PART;
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) u[i][j] = acc[i * 4 + j][qa][qb][r];
        float tm[3][4];                        // (SG)^T u: G with the sign of its last row flipped (see dM' above)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float hs = 0.5f * (u[1][j] + u[2][j]), hd = 0.5f * (u[1][j] - u[2][j]);
          tm[0][j] = u[0][j] + hs;
          tm[1][j] = hd;
          tm[2][j] = hs - u[3][j];
        }
        const int co = 16 * qa + 4 * tq + r, ci = 16 * qb + ch;
        const int half = (co >> 2) & 1, rr = (co & 3) + 4 * (co >> 3);
        float* dst = mine + rr * 64 + half * 32 + ci;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float hs = 0.5f * (tm[ky][1] + tm[ky][2]), hd = 0.5f * (tm[ky][1] - tm[ky][2]);
          dst[(ky * 3 + 0) * 1024] = tm[ky][0] + hs;
          dst[(ky * 3 + 1) * 1024] = hd;
          dst[(ky * 3 + 2) * 1024] = hs - tm[ky][3];
        }
      }
  // bias: lanes with the same channel (4 tile quarters) are added by the record reduction's two slots and here
  {
    float b0 = bsum[0], b1 = bsum[1];
    b0 += __shfl_xor(b0, 16); b0 += __shfl_xor(b0, 32);
    b1 += __shfl_xor(b1, 16); b1 += __shfl_xor(b1, 32);
    if (lane < 16) {
      mine[9 * 1024 + ch] = b0;
      mine[9 * 1024 + 16 + ch] = b1;
      mine[9 * 1024 + 32 + ch] = 0.f;
      mine[9 * 1024 + 48 + ch] = 0.f;
    }
  }
  WW_MARK(3);
  __syncthreads();
  WW_MARK(4);
  float4* out = reinterpret_cast<float4*>(a.part + (long)bx * PART);
  const float4* r4 = reinterpret_cast<const float4*>(red);
  for (int i = threadIdx.x; i < PART / 4; i += 256) {
    const float4 p = r4[i], q = r4[PART / 4 + i], v = r4[2 * (PART / 4) + i], w = r4[3 * (PART / 4) + i];
    out[i] = make_float4((p.x + q.x) + (v.x + w.x), (p.y + q.y) + (v.y + w.y), (p.z + q.z) + (v.z + w.z),
                         (p.w + q.w) + (v.w + w.w));
  }
#ifdef DRQ_DEV
  if (d_ww_stamps) __builtin_amdgcn_s_waitcnt(0);
#endif
  WW_MARK(5);
  WW_MARK(7);
}

template <int HIN, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_wino_kernel(WWArgs a) {
  const int G = (int)gridDim.x;   // XCD-aware order, see conv3x3_wgrad_wino3_kernel
  ww_body<HIN, ABL>(a, (G & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3), G);
}

// The three 32->32 layers of the encoder backward in ONE launch: layer l owns workgroups [beg[l], beg[l+1]) of the
// grid (shares proportional to its steps), every wave still accumulates one layer only and every workgroup writes one
// record of its layer.  Two launches' worth of cold start, epilogue and tail less than three launches.
struct WW3Args {
  WWArgs l[3];        // HIN = 41, 39, 37 (conv2, conv3, conv4)
  int beg[4];
};
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_wino3_kernel(WW3Args g) {
  // XCD-aware order (workgroups go to the eight XCDs round-robin by blockIdx; speed only): every XCD takes a contiguous
  // eighth of the grid, so that the workgroups which read neighbouring steps of a round -- 40 of 128 bytes of a line
  // horizontally, two of four input rows vertically -- share an L2 instead of pulling each line through the fabric once
  // per XCD
  const int G = (int)gridDim.x;
  const int b = (G & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  if (b < g.beg[1]) ww_body<41>(g.l[0], b, g.beg[1]);
  else if (b < g.beg[2]) ww_body<39>(g.l[1], b - g.beg[1], g.beg[2] - g.beg[1]);
  else ww_body<37>(g.l[2], b - g.beg[2], g.beg[3] - g.beg[2]);
}

template <int HIN>
int launch_ww(const WWArgs& a, int* nblocks, hipStream_t st) {
  constexpr int PART = 9 * 1024 + 64;
  constexpr int lds = 4 * PART * 4;
  static bool attr_dev[kMaxDevices] = {};
  bool& attr = attr_dev[drq_device()];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_wino_kernel<HIN>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  constexpr int TH = (HIN - 1) / 2;
  long blocks = drq_num_cus();
  const long steps = ((long)a.nb * TH * TH + 3) / 4;
  if (blocks * 4 > steps) blocks = (steps + 3) / 4;
  if (blocks < 1) blocks = 1;
#ifdef DRQ_DEV
  if (g_ww_variant) {
    const void* fn = g_ww_variant == 1 ? (const void*)conv3x3_wgrad_wino_kernel<HIN, 1>
                   : g_ww_variant == 2 ? (const void*)conv3x3_wgrad_wino_kernel<HIN, 2> : (const void*)conv3x3_wgrad_wino_kernel<HIN, 3>;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (g_ww_variant == 1) hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<HIN, 1>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    else if (g_ww_variant == 2) hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<HIN, 2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<HIN, 3>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    DRQ_LAUNCH_CHECK();
    *nblocks = (int)blocks;
    return DRQ_OK;
  }
#endif
  hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<HIN>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  DRQ_LAUNCH_CHECK();
  *nblocks = (int)blocks;
  return DRQ_OK;
}

}  // namespace

#ifdef DRQ_DEV
extern "C" DRQ_API void drq_dev_wgrad_wino_variant(int v) { g_ww_variant = v; }
extern "C" DRQ_API void drq_dev_wgrad_wino_stamps(unsigned long long* p) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(d_ww_stamps), &p, sizeof(p));
}
#endif

// internal (step.hip): conv2, conv3, conv4 (hin 41, 39, 37) in one launch; x[l], dy[l], part[l] and the dY strides per
// layer; nblocks[l] = records written for layer l
int drq_conv3x3_wgrad_partial_wino3(const float* const* x, const float* const* dy, int nb, const long* dy_bs,
                                    const long* dy_cs, const long* dy_rs, const long* dy_off, float* const* part,
                                    size_t part_bytes, int* nblocks, hipStream_t st) {
  if (!x || !dy || !dy_bs || !dy_cs || !dy_rs || !dy_off || !part || !nblocks || nb <= 0) return DRQ_EARG;
  constexpr int PART = 9 * 1024 + 64;
  const int hins[3] = {41, 39, 37};
  WW3Args g{};
  long steps[3], tot = 0;
  for (int l = 0; l < 3; ++l) {
    const int hin = hins[l], th = (hin - 1) / 2;
    if (!x[l] || !dy[l] || !part[l] || ((size_t)part[l] & 15)) return DRQ_EARG;
    const size_t xb = (size_t)nb * 32 * hin * hin * 4, dyb = (size_t)nb * dy_bs[l] * 4;
    if (xb >= (1ull << 31) || dyb >= (1ull << 31) || dy_off[l] < 0 || dy_bs[l] <= 0) return DRQ_EARG;
    g.l[l] = WWArgs{x[l], dy[l], (int)dy_bs[l], (int)dy_cs[l], (int)dy_rs[l], (int)dy_off[l], part[l], (unsigned)xb,
                    (unsigned)dyb, nb};
    steps[l] = ((long)nb * th * th + 3) / 4;
    tot += steps[l];
  }
  const int cus = drq_num_cus();
  if (cus < 3 || tot < 3L * 4) return DRQ_EARG;          // tiny problems: the caller launches the layers one by one
  // workgroups per layer proportional to its steps (at least one each), all of them resident at once
  int nb_l[3], used = 0;
  for (int l = 0; l < 3; ++l) {
    long n = (steps[l] * cus + tot / 2) / tot;
    if (n < 1) n = 1;
    if (n * 4 > steps[l]) n = (steps[l] + 3) / 4;
    nb_l[l] = (int)n;
    used += nb_l[l];
  }
  while (used > cus) {                                   // rounding overshoot: take from the largest share
    int m = 0;
    for (int l = 1; l < 3; ++l) if (nb_l[l] > nb_l[m]) m = l;
    --nb_l[m]; --used;
  }
  g.beg[0] = 0;
  for (int l = 0; l < 3; ++l) {
    g.beg[l + 1] = g.beg[l] + nb_l[l];
    nblocks[l] = nb_l[l];
    if ((size_t)nb_l[l] * PART * sizeof(float) > part_bytes) return DRQ_EWS;
  }
  constexpr int lds = 4 * PART * 4;
  static bool attr_dev[kMaxDevices] = {};
  bool& attr = attr_dev[drq_device()];
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_wino3_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(conv3x3_wgrad_wino3_kernel, dim3((unsigned)g.beg[3]), dim3(256), lds, st, g);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// internal (step.hip, conv.hip's public entry): partial records of the 32->32 weight gradient in Winograd form, same
// record format and reduction as drq_conv3x3_wgrad_partial
int drq_conv3x3_wgrad_partial_wino(const float* x, const float* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                   long dy_off, float* part, size_t part_bytes, int* nblocks, hipStream_t st) {
  if (!x || !dy || !part || !nblocks || nb <= 0) return DRQ_EARG;
  const size_t xb = (size_t)nb * 32 * hin * hin * 4;
  const size_t dyb = (size_t)nb * dy_bs * 4;
  if (xb >= (1ull << 31) || dyb >= (1ull << 31) || dy_off < 0 || dy_bs <= 0) return DRQ_EARG;
  // The kernel reads row and column `hout` of every dy plane as the empty half of the last 2x2 tile: dy must be the
  // interior of a zero-padded buffer (one more zero row and column at least), as the update stores its gradients.  A
  // contiguous [nb][32][hout][hout] dy would silently wrap into the next row / plane: refused.
  {
    const int hout = hin - 2;
    if (dy_rs < hout + 1 || dy_cs < (long)(hout + 1) * dy_rs) return DRQ_EARG;
  }
  if ((size_t)drq_num_cus() * (9 * 1024 + 64) * sizeof(float) > part_bytes) return DRQ_EWS;
  if (((size_t)part & 15) != 0) return DRQ_EARG;
  WWArgs a{x, dy, (int)dy_bs, (int)dy_cs, (int)dy_rs, (int)dy_off, part, (unsigned)xb, (unsigned)dyb, nb};
  if (hin == 41) return launch_ww<41>(a, nblocks, st);
  if (hin == 39) return launch_ww<39>(a, nblocks, st);
  if (hin == 37) return launch_ww<37>(a, nblocks, st);
  return DRQ_EARG;
}
