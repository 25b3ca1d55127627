// HBM-bound and tiny kernels of the DrQ-v2 update step:
//   RandomShiftsAug (drqv2.py:19-45) fused with the encoder's /255-0.5 (drqv2.py:64),
//   LayerNorm+Tanh forward/backward (drqv2.py:74-75), TruncatedNormal sample (utils.py:112-126),
//   TD target + twin MSE (drqv2.py:185-189), actor loss (drqv2.py:212-216, 225-226),
//   bias-gradient column sums, Adam (torch/optim/adam.py) with optional fused Polyak (utils.py:42-45).
// All reductions are fixed-order (no float atomics): results are run-to-run bit-stable.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// RandomShiftsAug: replicate-pad by `pad`, bilinear grid_sample (zeros, align_corners=False) at
// base[j] + shift*2/S.  One thread per output pixel, looping over channels (coordinates and the
// four tap weights are per (sample,y,x)).  Arithmetic follows ATen's GridSampler scalar formulas.
// ------------------------------------------------------------------------------------------------
// obs1/shift1 (optional): a second set of n frames handled by blockIdx.y == 1, written behind the first n frames of
// out (the update augments obs and next_obs with independent shifts: one launch for both)
// v / 255 correctly rounded (v >= 0, normal range) in three instructions instead of the IEEE division sequence:
// q = v*r, e = v - 255 q (exact), q' = q + e*r with r = RN(1/255).  tools/check_div255.py checks every float32
// mantissa against the exactly rounded quotient (the identity is scale invariant).
__device__ __forceinline__ float div255(float v) {
  const float r = 1.0f / 255.0f;
  const float q = __fmul_rn(v, r);
  const float e = __fmaf_rn(-q, 255.0f, v);
  return __fmaf_rn(e, r, q);
}

template <typename T>
__global__ void aug_kernel(const T* __restrict__ obs, const float* __restrict__ shift,
                           const float* __restrict__ base, float* __restrict__ out, int n, int c, int h,
                           int pad, int fuse_norm, const T* __restrict__ obs1 = nullptr,
                           const float* __restrict__ shift1 = nullptr) {
  // one rounding per operation, in the order of the CPU restatement (oracle/drq_oracle.py
  // random_shifts_aug): the tap weights are differences of nearly equal numbers, so a fused
  // multiply-add anywhere in the coordinate chain changes them by O(1) relative.
#pragma clang fp contract(off)
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int hw = h * h;
  if (idx >= (long)n * hw) return;
  if (blockIdx.y == 1) {
    obs = obs1;
    shift = shift1;
    out += (long)n * c * hw;
  }
  const int b = (int)(idx / hw);
  const int r = (int)(idx - (long)b * hw);
  const int i = r / h, j = r - i * h;
  const int S = h + 2 * pad;
  const float sc = (float)(2.0 / (double)S);
  const float gx = base[j] + shift[2 * b + 0] * sc;
  const float gy = base[i] + shift[2 * b + 1] * sc;
  const float ix = ((gx + 1.f) * (float)S - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * (float)S - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;   // nw, ne, sw, se
  const bool okx0 = x0 >= 0 && x0 < S, okx1 = x0 + 1 >= 0 && x0 + 1 < S;
  const bool oky0 = y0 >= 0 && y0 < S, oky1 = y0 + 1 >= 0 && y0 + 1 < S;
  auto cl = [&](int v) { v -= pad; return v < 0 ? 0 : (v > h - 1 ? h - 1 : v); };
  const int sx0 = cl(x0), sx1 = cl(x0 + 1), sy0 = cl(y0), sy1 = cl(y0 + 1);
  const T* src = obs + (long)b * c * hw;
  float* dst = out + (long)b * c * hw + r;
  const int o00 = sy0 * h + sx0, o01 = sy0 * h + sx1, o10 = sy1 * h + sx0, o11 = sy1 * h + sx1;
  const bool k00 = okx0 && oky0, k01 = okx1 && oky0, k10 = okx0 && oky1, k11 = okx1 && oky1;
  auto blend = [&](float t00, float t01, float t10, float t11) {
    // each product is rounded on its own (torch's grid_sample does not fuse them): the empty asm keeps hipcc from
    // contracting a product into the following add, which the contract(off) pragma does not prevent in a lambda
    float p0 = t00 * w00, p1 = t01 * w01, p2 = t10 * w10, p3 = t11 * w11;
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    float v = 0.f;
    if (k00) v = v + p0;
    if (k01) v = v + p1;
    if (k10) v = v + p2;
    if (k11) v = v + p3;
    if (fuse_norm) v = __fsub_rn(div255(v), 0.5f);
    return v;
  };
  if (c == 9) {          // frame_stack 3 (cfgs/config.yaml:7): all 36 tap loads in flight together
    T t[9][4];
#pragma unroll
    for (int ch = 0; ch < 9; ++ch) {
      const T* s = src + (long)ch * hw;
      t[ch][0] = s[o00]; t[ch][1] = s[o01]; t[ch][2] = s[o10]; t[ch][3] = s[o11];
    }
#pragma unroll
    for (int ch = 0; ch < 9; ++ch)
      dst[(long)ch * hw] = blend((float)t[ch][0], (float)t[ch][1], (float)t[ch][2], (float)t[ch][3]);
  } else {
    for (int ch = 0; ch < c; ++ch) {
      const T* s = src + (long)ch * hw;
      dst[(long)ch * hw] = blend((float)s[o00], (float)s[o01], (float)s[o10], (float)s[o11]);
    }
  }
}

// The same augmentation for uint8 frames with the source rows staged through LDS.  aug_kernel issues 4 byte
// loads per channel and pixel (36 global load instructions per thread, 64 bytes each per wave): at 9x84x84 it is
// bound by the texture addresser, not by memory (63 us for 163 MB).  Here a workgroup owns R output rows of one
// sample: it copies the source rows those touch (all channels, usually R+1 rows of h bytes) into LDS with dword
// loads, and the four taps become LDS byte reads.  Coordinates, weights, tap selection and the blend are the
// statements of aug_kernel, so results are bit-identical; a row range that does not fit NR rows (not expected:
// the grid is monotone with unit steps) reads its taps from global memory like aug_kernel.
__global__ __launch_bounds__(256) void aug_rows_kernel(const uint8_t* __restrict__ obs, const float* __restrict__ shift,
                                                       const float* __restrict__ base, float* __restrict__ out,
                                                       int n, int c, int h, int pad, int fuse_norm,
                                                       const uint8_t* __restrict__ obs1,
                                                       const float* __restrict__ shift1, int R, int NR) {
#pragma clang fp contract(off)
  extern __shared__ unsigned aug_lds[];           // [c][NR][h/4] dwords
  const int hw = h * h, hq = h >> 2;
  const int b = blockIdx.y;
  if (blockIdx.z == 1) {
    obs = obs1;
    shift = shift1;
    out += (long)n * c * hw;
  }
  const int S = h + 2 * pad;
  const float sc = (float)(2.0 / (double)S);
  auto cl = [&](int v) { v -= pad; return v < 0 ? 0 : (v > h - 1 ? h - 1 : v); };
  const float shx = shift[2 * b + 0] * sc, shy = shift[2 * b + 1] * sc;
  auto coord = [&](int k, float sh, float& f) {   // unnormalised sample coordinate of grid index k
    const float g = base[k] + sh;
    const float v = ((g + 1.f) * (float)S - 1.f) / 2.f;
    f = floorf(v);
    return v;
  };
  const int i0 = blockIdx.x * R;
  const int ilast = i0 + R - 1 < h - 1 ? i0 + R - 1 : h - 1;
  float fa, fb;
  coord(i0, shy, fa);
  coord(ilast, shy, fb);
  const int sy_lo = cl((int)fa), sy_hi = cl((int)fb + 1);       // the grid is increasing: rows in between lie inside
  const int nrows = sy_hi - sy_lo + 1;
  const bool staged = nrows >= 1 && nrows <= NR;
  const uint8_t* src = obs + (long)b * c * hw;
  if (staged) {
    // thread -> (row slot, dword of the row); row slots walk the (channel, row) pairs without a division per step
    const int rs = threadIdx.x / hq, q = threadIdx.x - rs * hq;
    const int nslots = 256 / hq;
    if (rs < nslots) {
      const int step_ch = nslots / nrows, step_r = nslots - step_ch * nrows;
      int ch = rs / nrows, r = rs - ch * nrows;
      while (ch < c) {
        aug_lds[(ch * NR + r) * hq + q] =
            reinterpret_cast<const unsigned*>(src + (long)ch * hw + (long)(sy_lo + r) * h)[q];
        r += step_r;
        ch += step_ch;
        if (r >= nrows) {
          r -= nrows;
          ++ch;
        }
      }
    }
  }
  __syncthreads();
  const int il = threadIdx.x / h, j = threadIdx.x - il * h;
  const int i = i0 + il;
  if (il >= R || i >= h) return;
  float fx, fy;
  const float ix = coord(j, shx, fx), iy = coord(i, shy, fy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  const bool okx0 = x0 >= 0 && x0 < S, okx1 = x0 + 1 >= 0 && x0 + 1 < S;
  const bool oky0 = y0 >= 0 && y0 < S, oky1 = y0 + 1 >= 0 && y0 + 1 < S;
  const int sx0 = cl(x0), sx1 = cl(x0 + 1), sy0 = cl(y0), sy1 = cl(y0 + 1);
  // a tap outside the padded frame is skipped by aug_kernel; here it gets weight +0 (t >= 0, so the sum is
  // unchanged bit for bit), and the first add of aug_kernel is 0 + x = x: no selects and one add less per channel
  const float w00 = okx0 && oky0 ? wx0 * wy0 : 0.f, w01 = okx1 && oky0 ? wx1 * wy0 : 0.f;     // nw, ne
  const float w10 = okx0 && oky1 ? wx0 * wy1 : 0.f, w11 = okx1 && oky1 ? wx1 * wy1 : 0.f;     // sw, se
  auto blend = [&](float t00, float t01, float t10, float t11) {
    // each product is rounded on its own: the empty asm keeps hipcc from contracting it into the following add
    // (the contract(off) pragma is not honoured inside this lambda once the selects around the adds are gone)
    float p0 = t00 * w00, p1 = t01 * w01, p2 = t10 * w10, p3 = t11 * w11;
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    float v = p0 + p1;
    v = v + p2;
    v = v + p3;
    if (fuse_norm) v = __fsub_rn(div255(v), 0.5f);
    return v;
  };
  float* dst = out + (long)b * c * hw + i * h + j;
  if (staged) {
    const uint8_t* l = reinterpret_cast<const uint8_t*>(aug_lds);
    const int o00 = (sy0 - sy_lo) * h + sx0, o01 = (sy0 - sy_lo) * h + sx1;
    const int o10 = (sy1 - sy_lo) * h + sx0, o11 = (sy1 - sy_lo) * h + sx1;
    for (int ch = 0; ch < c; ++ch) {
      const uint8_t* s = l + ch * NR * h;
      dst[(long)ch * hw] = blend((float)s[o00], (float)s[o01], (float)s[o10], (float)s[o11]);
    }
  } else {
    const int o00 = sy0 * h + sx0, o01 = sy0 * h + sx1, o10 = sy1 * h + sx0, o11 = sy1 * h + sx1;
    for (int ch = 0; ch < c; ++ch) {
      const uint8_t* s = src + (long)ch * hw;
      dst[(long)ch * hw] = blend((float)s[o00], (float)s[o01], (float)s[o10], (float)s[o11]);
    }
  }
}

// rows-in-LDS launch when the frame layout allows it (dword rows); returns false when it does not
bool launch_aug_rows(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                     const float* base, float* out, int n, int c, int h, int pad, int fuse_norm, hipStream_t st) {
  if (h % 4 != 0 || h > 256 || n > 65535 || ((uintptr_t)obs & 3) || (obs1 && ((uintptr_t)obs1 & 3))) return false;
  const int R = 256 / h, NR = R + 2;
  const size_t lds = (size_t)c * NR * h;
  if (R < 1 || lds > 64 * 1024) return false;
  hipLaunchKernelGGL(aug_rows_kernel, dim3((unsigned)((h + R - 1) / R), (unsigned)n, obs1 ? 2 : 1), dim3(256), lds, st,
                     obs, shift, base, out, n, c, h, pad, fuse_norm, obs1, shift1, R, NR);
  return true;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm + tanh.  One wave per row, F <= 256.
// ------------------------------------------------------------------------------------------------
struct LnArgs {
  const float* z[4];      // [rows][ldz] pre-norm (bias already added)
  const float* gamma[4];
  const float* beta[4];
  float* out[4];          // [rows][ldo] tanh(LN(z))
  float* xhat[4];         // [rows][F] (may be null)
  float* rstd[4];         // [rows]   (may be null)
  int ldz[4], ldo[4];
  int rows, F;
  const float* tail[4];   // optional [rows][tail_ld]: copied into out columns [F, F+tail_n) (the [h, action] concat)
  int tail_ld[4];
  int tail_n;
  const float* part;      // optional: z is not given but split-K partials [n*splitk][rows][F] of the trunk GEMM
  const float* bias[4];   //           (gemm.hip layout) plus the layer bias; summed in splitk_reduce_kernel's order
  int splitk;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sum of the split-K partial slabs of one GEMM output element, in splitk_reduce_kernel's association
// (gemm.hip): p points at the element in slab 0, consecutive slabs are `slab` floats apart
__device__ __forceinline__ float sum_partials(const float* p, long slab, int splitk) {
  // four interleaved chains (slab k goes to chain k & 3), then a tree: splitk_reduce_kernel's order.  16 slabs are
  // loaded per batch so that a thread pays one memory round trip per 16 slabs, not one per 4.
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k0 = 0; k0 < splitk; k0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(long)min(k0 + u, splitk - 1) * slab];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      if (k0 + u < splitk) s0 += v[u];
      if (k0 + u + 1 < splitk) s1 += v[u + 1];
      if (k0 + u + 2 < splitk) s2 += v[u + 2];
      if (k0 + u + 3 < splitk) s3 += v[u + 3];
    }
  }
  return (s0 + s1) + (s2 + s3);
}

__global__ void ln_tanh_fwd_kernel(LnArgs a) {
  const int g = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  float v[4];
  float s = 0.f;
  if (a.part) {
    const long mn = (long)a.rows * a.F;
    const float* p = a.part + (long)g * a.splitk * mn + (long)row * a.F;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = lane + 64 * q;
      if (f < a.F) {
        v[q] = sum_partials(p + f, mn, a.splitk) + (a.bias[g] ? a.bias[g][f] : 0.f);
      } else {
        v[q] = 0.f;
      }
      s += v[q];
    }
  } else {
    const float* z = a.z[g] + (long)row * a.ldz[g];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = lane + 64 * q;
      v[q] = f < a.F ? z[f] : 0.f;
      s += v[q];
    }
  }
  const float mean = wave_sum(s) / (float)a.F;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    const float d = f < a.F ? v[q] - mean : 0.f;
    ss += d * d;
  }
  const float var = wave_sum(ss) / (float)a.F;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    if (f < a.F) {
      const float xh = (v[q] - mean) * rstd;
      const float y = xh * a.gamma[g][f] + a.beta[g][f];
      a.out[g][(long)row * a.ldo[g] + f] = tanhf(y);
      if (a.xhat[g]) a.xhat[g][(long)row * a.F + f] = xh;
    }
  }
  if (lane == 0 && a.rstd[g]) a.rstd[g][row] = rstd;
  if (a.tail[g] && lane < a.tail_n) a.out[g][(long)row * a.ldo[g] + a.F + lane] = a.tail[g][(long)row * a.tail_ld[g] + lane];
}

struct LnBwdArgs {
  const float* dh0;     // [rows][ld0] gradient w.r.t. tanh output (source 0)
  const float* dh1;     // optional second source, summed
  const float* h;       // [rows][ldh] saved tanh output
  const float* xhat;    // [rows][F]
  const float* rstd;    // [rows]
  const float* gamma;
  float* dz;            // [rows][F]
  float* dln;           // [rows][F] scratch: gradient w.r.t. the LN output (for dgamma/dbeta)
  int ld0, ld1, ldh;
  int rows, F;
  // optional: the incoming gradient is given as split-K partials of `nprob` dgrad GEMMs ([nprob*splitk][rows][ldp],
  // gemm.hip layout) whose results are added (dh0/dh1 are then unused)
  const float* part;
  int splitk, nprob, ldp;
};

__global__ void ln_tanh_bwd_kernel(LnBwdArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  float dxh[4], xh[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    dxh[q] = 0.f;
    xh[q] = 0.f;
    if (f < a.F) {
      float d;
      if (a.part) {
        const long slab = (long)a.rows * a.ldp;
        const float* p = a.part + (long)row * a.ldp + f;
        d = sum_partials(p, slab, a.splitk);
        if (a.nprob > 1) d += sum_partials(p + (long)a.splitk * slab, slab, a.splitk);
      } else {
        d = a.dh0[(long)row * a.ld0 + f];
        if (a.dh1) d += a.dh1[(long)row * a.ld1 + f];
      }
      const float hv = a.h[(long)row * a.ldh + f];
      const float dl = d * (1.f - hv * hv);
      a.dln[(long)row * a.F + f] = dl;
      xh[q] = a.xhat[(long)row * a.F + f];
      dxh[q] = dl * a.gamma[f];
      s1 += dxh[q];
      s2 += dxh[q] * xh[q];
    }
  }
  const float m1 = wave_sum(s1) / (float)a.F;
  const float m2 = wave_sum(s2) / (float)a.F;
  const float rs = a.rstd[row];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    if (f < a.F) a.dz[(long)row * a.F + f] = rs * (dxh[q] - m1 - xh[q] * m2);
  }
}

// dgamma[f] = sum_rows dln*xhat, dbeta[f] = sum_rows dln.  One block (256 threads) per column.
__global__ void ln_param_grad_kernel(const float* dln, const float* xhat, float* dgamma, float* dbeta, int rows,
                                     int F) {
  __shared__ float sg[256], sb[256];
  const int f = blockIdx.x;
  float g = 0.f, b = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float d = dln[(long)r * F + f];
    g += d * xhat[(long)r * F + f];
    b += d;
  }
  sg[threadIdx.x] = g;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sg[threadIdx.x] += sg[threadIdx.x + o];
      sb[threadIdx.x] += sb[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dgamma[f] = sg[0];
    dbeta[f] = sb[0];
  }
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[batch][n] = sum_m dy[batch][m][n].  Block = 64 columns x 4 row groups.
// ------------------------------------------------------------------------------------------------
// CW columns x 1024/CW row groups per workgroup: 64 x 16 for the batches of the reference's configs; from 1,024 rows on
// 16 x 64 (four times the workgroups and a quarter of the rows per thread: 21 -> 7 us for [2048][1024], five launches per
// bf16 update).  Fixed order either way.
template <int CW>
__global__ __launch_bounds__(1024) void colsum_kernel(const float* dy, long ld, long dy_bs, float* out, long out_bs,
                                                      int M, int N) {
  constexpr int NRG = 1024 / CW;
  __shared__ float s[NRG][CW];
  const int batch = blockIdx.y;
  const int c = threadIdx.x % CW;
  const int n = blockIdx.x * CW + c;
  const int rg = threadIdx.x / CW;
  const float* p = dy + batch * dy_bs;
  float a0 = 0.f, a1 = 0.f;
  if (n < N) {
    int m = rg;
    for (; m + NRG < M; m += 2 * NRG) {     // two independent chains, fixed order
      a0 += p[(long)m * ld + n];
      a1 += p[(long)(m + NRG) * ld + n];
    }
    if (m < M) a0 += p[(long)m * ld + n];
  }
  s[rg][c] = a0 + a1;
  __syncthreads();
  if (rg == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < NRG; ++g2) t += s[g2][c];
    out[batch * out_bs + n] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// policy head output: mu = tanh(pre); a = clampST(mu + clamp(noise*std, +-clip))   (utils.py:117-126)
// ------------------------------------------------------------------------------------------------
__global__ void sample_action_kernel(const float* pre, const float* noise, float std, float clip, int use_clip,
                                     float* mu_out, float* a_out, long lda_out, int B, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const int b = i / A, j = i - b * A;
  const float mu = tanhf(pre[i]);
  float eps = noise[i] * std;
  if (use_clip) eps = fminf(fmaxf(eps, -clip), clip);
  const float x = mu + eps;
  const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
  const float a = fminf(fmaxf(x, lo), hi);
  if (mu_out) mu_out[i] = mu;
  a_out[(long)b * lda_out + j] = a;
}

__global__ void copy_cols_kernel(const float* src, int ld_src, float* dst, long ld_dst, int B, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const int b = i / A, j = i - b * A;
  dst[(long)b * ld_dst + j] = src[(long)b * ld_src + j];
}

// block-wide fixed-order sum of up to 4 values (256 threads)
__device__ __forceinline__ void block_sum4(float (&v)[4], float (*sm)[256]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] = v[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] += sm[q][threadIdx.x + o];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = sm[q][0];
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// TD target + twin MSE (drqv2.py:185-189).  Single block.  invB = 1/global batch.
// sums[0..4] = sum reward, sum target_q, sum q1, sum q2, sum (q1-y)^2+(q2-y)^2  (un-normalised: the
// data-parallel path adds them across ranks before dividing).
// ------------------------------------------------------------------------------------------------
__global__ void td_loss_kernel(const float* tq1, const float* tq2, const float* q1, const float* q2,
                               const float* reward, const float* discount, float* dq1, float* dq2, float* sums,
                               int B, float invB) {
  __shared__ float sm[4][256];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  float sr = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float y = reward[b] + discount[b] * fminf(tq1[b], tq2[b]);
    const float e1 = q1[b] - y, e2 = q2[b] - y;
    dq1[b] = 2.f * e1 * invB;
    dq2[b] = 2.f * e2 * invB;
    s[0] += y;
    s[1] += q1[b];
    s[2] += q2[b];
    s[3] += e1 * e1 + e2 * e2;
    sr += reward[b];
  }
  block_sum4(s, sm);
  float r4[4] = {sr, 0.f, 0.f, 0.f};
  block_sum4(r4, sm);
  if (threadIdx.x == 0) {
    sums[0] = r4[0];
    sums[1] = s[0];
    sums[2] = s[1];
    sums[3] = s[2];
    sums[4] = s[3];
  }
}

// ------------------------------------------------------------------------------------------------
// actor loss (drqv2.py:212-216): L = -mean(min(q1,q2)); gradient routed to the smaller head, ties split.
// sums[5] = sum -min(q1,q2), sums[6] = sum_b sum_A log N(a|mu,std)
// ------------------------------------------------------------------------------------------------
__global__ void actor_loss_kernel(const float* q1, const float* q2, const float* a, long lda, const float* mu,
                                  float std, float* dq1, float* dq2, float* sums, int B, int A, float invB,
                                  float* sums_host, unsigned seq) {
  __shared__ float sm[4][256];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const float log_std = logf(std);
  const float c = 0.91893853320467274178f;   // log(sqrt(2*pi))
  const float var2 = 2.f * std * std;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float x = q1[b], y = q2[b];
    s[0] += -fminf(x, y);
    const float g = -invB;
    dq1[b] = x < y ? g : (x == y ? 0.5f * g : 0.f);
    dq2[b] = y < x ? g : (x == y ? 0.5f * g : 0.f);
  }
  // log-probabilities element by element over the flattened [B][A] array (consecutive lanes, consecutive addresses),
  // thread t adding elements t, t + 256, ... in that order: the scheme of the fused form in qout_bwd_kernel
  const int nel = B * A;
  for (int e0 = threadIdx.x; e0 < nel; e0 += 4 * 256) {
    float av[4], mv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256 < nel ? e0 + u * 256 : nel - 1;
      const int m = e / A, jj = e - m * A;
      av[u] = a[(long)m * lda + jj];
      mv[u] = mu[e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * 256 < nel) {
        const float d = av[u] - mv[u];
        s[1] += -(d * d) / var2 - log_std - c;
      }
  }
  block_sum4(s, sm);
  if (threadIdx.x == 0) {
    sums[5] = s[0];
    sums[6] = s[1];
    if (sums_host) {   // metrics mirror (DrqStep.sums_host): all eight sums, then the sequence word
      float v8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = i == 5 ? s[0] : (i == 6 ? s[1] : sums[i]);
      drq_publish_mirror(sums_host, v8, seq);
    }
  }
}

// d(pre-tanh policy output) = (da1 + da2) * (1 - mu^2): straight-through clamp (utils.py:112-115),
// da_k = action columns of head k's layer-1 input gradient.
__global__ void actor_dmu_kernel(const float* dha1, const float* dha2, long ld, int col0, const float* mu,
                                 float* dpre, int B, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const int b = i / A, j = i - b * A;
  const float da = dha1[(long)b * ld + col0 + j] + dha2[(long)b * ld + col0 + j];
  const float m = mu[i];
  dpre[i] = da * (1.f - m * m);
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults) over a flat arena, rounding sequence of torch's CPU kernels
// (oracle/drq_oracle.py:adam_step), optional fused Polyak update of a target arena.
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float neg_step_size, float sqrt_bc2, float gscale,
                            float* __restrict__ tgt, float tau, float one_minus_tau) {
  // one rounding per operation, in torch's order (oracle/drq_oracle.py:adam_step); sqrtf and / are the
  // IEEE-correct forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is NOT.
#pragma clang fp contract(off)
  const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (gscale != 1.0f) gi *= gscale;
    const float mi = __fmaf_rn(w1, __fsub_rn(gi, m[i]), m[i]);
    const float vi = __fmaf_rn(__fmul_rn(gi, w2), gi, __fmul_rn(v[i], b2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), sqrt_bc2), eps);
    const float pi = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (tgt) tgt[i] = __fadd_rn(__fmul_rn(tau, pi), __fmul_rn(one_minus_tau, tgt[i]));
  }
}

struct AdamSeg2 {
  float* p[2];
  const float* g[2];
  float* m[2];
  float* v[2];
  long n[2];
  float neg_step_size[2], sqrt_bc2[2];
  float gscale;
};

__global__ void adam2_kernel(AdamSeg2 a) {
#pragma clang fp contract(off)
  const int z = blockIdx.y;
  float* __restrict__ p = a.p[z];
  const float* __restrict__ g = a.g[z];
  float* __restrict__ m = a.m[z];
  float* __restrict__ v = a.v[z];
  const long n = a.n[z];
  const float neg_step_size = a.neg_step_size[z], sqrt_bc2 = a.sqrt_bc2[z];
  const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (a.gscale != 1.0f) gi *= a.gscale;
    const float mi = __fmaf_rn(w1, __fsub_rn(gi, m[i]), m[i]);
    const float vi = __fmaf_rn(__fmul_rn(gi, w2), gi, __fmul_rn(v[i], b2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), sqrt_bc2), eps);
    const float pi = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// Data parallel, sharded optimiser (ZeRO-1): this rank owns n parameters of a segment; recv holds the `world` ranks'
// copies of its gradient slice (recv[r*stride + i], the result of an all-to-all) -- summed here in RANK ORDER (the same
// order on every rank and in drq_sum_slices: bit-identical to the replicated path) and fed straight into the Adam
// arithmetic of adam_kernel; the gradient sum never goes back to memory.
__global__ void adam_reduce_kernel(float* __restrict__ p, const float* __restrict__ recv, long stride, int world,
                                   float* __restrict__ m, float* __restrict__ v, long n, float neg_step_size,
                                   float sqrt_bc2, float gscale) {
#pragma clang fp contract(off)
  const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = recv[i];
    for (int r = 1; r < world; ++r) gi = __fadd_rn(gi, recv[(long)r * stride + i]);
    if (gscale != 1.0f) gi *= gscale;
    const float mi = __fmaf_rn(w1, __fsub_rn(gi, m[i]), m[i]);
    const float vi = __fmaf_rn(__fmul_rn(gi, w2), gi, __fmul_rn(v[i], b2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), sqrt_bc2), eps);
    p[i] = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));
    m[i] = mi;
    v[i] = vi;
  }
}

// out[i] = ((in[i] + in[stride + i]) + in[2*stride + i]) + ... : the `world` copies of a gradient slice in rank order
__global__ void sum_slices_kernel(const float* __restrict__ in, long stride, int world, float* __restrict__ out, long n) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = in[i];
    for (int r = 1; r < world; ++r) s = __fadd_rn(s, in[(long)r * stride + i]);
    out[i] = s;
  }
}

__global__ void ema_kernel(const float* __restrict__ p, float* __restrict__ t, long n, float tau,
                           float one_minus_tau) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    t[i] = __fadd_rn(__fmul_rn(tau, p[i]), __fmul_rn(one_minus_tau, t[i]));
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void u8_normalize_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (float)x[i] / 255.0f - 0.5f;
}

__global__ void tanh_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = tanhf(x[i]);
}

// ------------------------------------------------------------------------------------------------
// Output layer of a Q head (nn.Linear(hidden,1), drqv2.py:106,111): a GEMM with N = 1 is a row dot
// product; forward = one wave per row, backward = dgrad (outer product, ReLU-masked), wgrad and bias
// gradient in ONE pass over the hidden activations.  Up to 8 (net, head) problems per launch.
// ------------------------------------------------------------------------------------------------
struct QOutArgs {
  const float* h[8];    // [B][H] hidden activations (post-ReLU)
  const float* w[8];    // [H]
  const float* b[8];    // [1]
  float* q[8];          // [B]
  int B, H;
};

__global__ void qout_fwd_kernel(QOutArgs a) {
  const int z = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.B) return;
  const float* h = a.h[z] + (long)row * a.H;
  const float* w = a.w[z];
  float s0 = 0.f, s1 = 0.f;
  if (a.H <= 1024) {
    // all (up to) 32 loads of the row in flight at once; same summation order as the loop below
    float hv[16], wv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = min(lane + 64 * q, a.H - 1);
      hv[q] = h[i];
      wv[q] = (lane + 64 * q < a.H) ? w[i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      s0 = __fmaf_rn(hv[q], wv[q], s0);
      s1 = __fmaf_rn(hv[q + 1], wv[q + 1], s1);
    }
  } else {
    int i = lane;
    for (; i + 64 < a.H; i += 128) {
      s0 = __fmaf_rn(h[i], w[i], s0);
      s1 = __fmaf_rn(h[i + 64], w[i + 64], s1);
    }
    if (i < a.H) s0 = __fmaf_rn(h[i], w[i], s0);
  }
  const float s = wave_sum(s0 + s1);
  if (lane == 0) a.q[z][row] = s + a.b[z][0];
}

struct QOutBwdArgs {
  const float* dq[8];   // [B]
  const float* h[8];    // [B][H]
  const float* w[8];    // [H]
  float* dh[8];         // [B][H] = dq*w masked by h > 0
  float* dw[8];         // [H] or null
  float* db[8];         // [1] or null
  int B, H;
  // optional fused TD loss (drqv2.py:185-189, two heads: z = 0, 1): dq is not given but computed per row from the
  // Q values; workgroup (0, 0) also leaves sums[0..4] like td_loss_kernel
  // td == 2: the actor loss instead (drqv2.py:212-216, actor_loss_kernel's arithmetic): dq routed to the smaller head,
  // ties split; workgroup (0, 0) leaves sums[5..6] and publishes all eight sums to the host mirror when one is given
  int td;
  const float *tq1, *tq2, *q1, *q2, *reward, *discount;
  float invB;
  float* sums;
  const float *act, *mu;   // td == 2: the sampled action [B][lda] and its mean [B][A]
  long lda;
  int A;
  float std;
  float* sums_host;
  unsigned seq;
};

// Workgroup = 64 columns x 16 row groups (1024 threads).  The per-row scalars dq[B] go to LDS once; the only global
// stream of the row loop is h, 16 rows of it in flight per thread.
// CW columns per workgroup, NRG = 1024 / CW row groups.  64 x 16 is the shape for batches up to a few hundred rows
// (16 workgroups per head at hidden_dim 1024); at batch 2,048 that left 32 workgroups walking 2,048 rows each (60 us per
// launch): 16 columns x 64 row groups gives four times as many (drq_qout_bwd_cw).  The row-group sums are added in
// the same fixed order either way (deterministic); the two shapes group them differently (rounding-level).
template <int CW>
__global__ __launch_bounds__(1024) void qout_bwd_kernel(QOutBwdArgs a) {
  constexpr int NRG = 1024 / CW;
  extern __shared__ float dql[];            // [B] then NRG x CW + 64 floats of reduction scratch
  float* s = dql + a.B;
  float* sb = s + (a.td ? 5 * 1024 : 1024);
  const int z = blockIdx.y;
  const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int n = blockIdx.x * CW + c;
  const bool nok = n < a.H;
  const int nc = nok ? n : a.H - 1;
  const float* h = a.h[z];
  float* dh = a.dh[z];
  const float wn = a.w[z][nc];
  // the first 16 rows of h of this thread do not depend on dq: requested now, so that their round trip overlaps the
  // dq stage's (a kernel of this size is a chain of dependent memory round trips; every load issued late is one more)
  float hv0[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) hv0[u] = h[(long)min(rg + NRG * u, a.B - 1) * a.H + nc];
  if (a.td == 2) {
    const float *qz = z == 0 ? a.q1 : a.q2, *qo = z == 0 ? a.q2 : a.q1;
    const float g = -a.invB;
    for (int m = threadIdx.x; m < a.B; m += 1024) {
      const float x = qz[m], y = qo[m];
      dql[m] = x < y ? g : (x == y ? 0.5f * g : 0.f);
    }
  } else if (a.td) {
    const float* qz = z == 0 ? a.q1 : a.q2;
    for (int m = threadIdx.x; m < a.B; m += 1024) {
      const float y = a.reward[m] + a.discount[m] * fminf(a.tq1[m], a.tq2[m]);
      dql[m] = 2.f * (qz[m] - y) * a.invB;
    }
  } else {
    const float* dq = a.dq[z];
    for (int m = threadIdx.x; m < a.B; m += 1024) dql[m] = dq[m];
  }
  __syncthreads();
  float acc = 0.f, bacc = 0.f;
  for (int m0 = rg; m0 < a.B; m0 += NRG * 16) {
    float hv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) hv[u] = m0 == rg ? hv0[u] : h[(long)min(m0 + NRG * u, a.B - 1) * a.H + nc];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int m = m0 + NRG * u;
      if (m < a.B) {
        const float d = dql[m];
        bacc += d;
        if (nok) dh[(long)m * a.H + n] = hv[u] > 0.f ? d * wn : 0.f;
        acc = __fmaf_rn(d, hv[u], acc);
      }
    }
  }
  s[rg * CW + c] = acc;
  if (c == 0) sb[rg] = bacc;
  __syncthreads();
  if (rg == 0) {
    if (a.dw[z] && nok) {
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < NRG; ++g2) t += s[g2 * CW + c];
      a.dw[z][n] = t;
    }
    if (a.db[z] && blockIdx.x == 0 && c == 0) {
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < NRG; ++g2) t += sb[g2];
      a.db[z][0] = t;
    }
  }
  // metric sums of the actor loss: workgroup (0, 0), the same fixed tree; then the host mirror
  if (a.td == 2 && blockIdx.x == 0 && z == 0) {
    __syncthreads();
    float v[2] = {0.f, 0.f};
    const float log_std = logf(a.std);
    const float c0 = 0.91893853320467274178f;   // log(sqrt(2*pi))
    const float var2 = 2.f * a.std * a.std;
    for (int m = threadIdx.x; m < a.B; m += 1024) v[0] += -fminf(a.q1[m], a.q2[m]);
    // log-probabilities: element by element over the flattened [B][A] array, so that a wave reads consecutive addresses
    // (one row per lane reads 64 different cache lines per instruction, and this is ONE workgroup: 65 us at 2,048 rows
    // x 21 actions, measured).  Thread t adds elements t, t + 1024, ... in that order, then the fixed tree: deterministic;
    // the grouping differs from a row-by-row sum at rounding level.
    const int nel = a.B * a.A;
    for (int e0 = threadIdx.x; e0 < nel; e0 += 4 * 1024) {
      float av[4], mv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 1024 < nel ? e0 + u * 1024 : nel - 1;
        const int m = e / a.A, jj = e - m * a.A;
        av[u] = a.act[(long)m * a.lda + jj];
        mv[u] = a.mu[e];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * 1024 < nel) {
          const float d = av[u] - mv[u];
          v[1] += -(d * d) / var2 - log_std - c0;
        }
    }
    s[threadIdx.x] = v[0];
    s[1024 + threadIdx.x] = v[1];
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        s[threadIdx.x] += s[threadIdx.x + o];
        s[1024 + threadIdx.x] += s[1024 + threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float s5 = s[0], s6 = s[1024];
      a.sums[5] = s5;
      a.sums[6] = s6;
      if (a.sums_host) {   // metrics mirror (DrqStep.sums_host): all eight sums, then the sequence word
        float v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = i == 5 ? s5 : (i == 6 ? s6 : a.sums[i]);
        drq_publish_mirror(a.sums_host, v8, a.seq);
      }
    }
  }
  // metric sums of the TD loss: workgroup (0, 0), per-thread partials over rows tid, tid+1024, ... then a fixed tree
  if (a.td == 1 && blockIdx.x == 0 && z == 0) {
    __syncthreads();
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int m = threadIdx.x; m < a.B; m += 1024) {
      const float r = a.reward[m];
      const float y = r + a.discount[m] * fminf(a.tq1[m], a.tq2[m]);
      const float e1 = a.q1[m] - y, e2 = a.q2[m] - y;
      v[0] += r; v[1] += y; v[2] += a.q1[m]; v[3] += a.q2[m]; v[4] += e1 * e1 + e2 * e2;
    }
    // one tree for the five sums (scratch: 5 x 1024 floats behind dq)
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q * 1024 + threadIdx.x] = v[q];
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q * 1024 + threadIdx.x] += s[q * 1024 + threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x < 5) a.sums[threadIdx.x] = s[threadIdx.x * 1024];
  }
}

// ------------------------------------------------------------------------------------------------
// Replay sampling on the device (replay_buffer.py:142-160): batch row b is transition pos[b] of a flat store of
// steps (episodes contiguous): obs = frame[pos-1], next_obs = frame[pos+nstep-1], action = action[pos], and the
// n-step return in the reference's float32 order:  reward += discount*r[pos+i];  discount *= d[pos+i]*gamma.
// blockIdx.y = 0 / 1: obs / next_obs frame copy (16-byte pieces); blockIdx.y = 2: the scalars of 256 rows.
// ------------------------------------------------------------------------------------------------
struct NstepArgs {
  const uint8_t* frames;
  const float* action;
  const float* reward;
  const float* discount;
  const long* pos;
  uint8_t* obs;
  uint8_t* next_obs;
  float* act_out;
  float* rew_out;
  float* disc_out;
  long frame_bytes;   // multiple of 16
  int B, A, nstep;
  float gamma;
};

__global__ void nstep_gather_kernel(NstepArgs a) {
#pragma clang fp contract(off)
  const int which = a.obs ? blockIdx.y : 2;         // no output frames: the scalars only (grid.y == 1)
  if (which < 2) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const long p = a.pos[b] + (which == 0 ? -1 : (long)a.nstep - 1);
    const uint4* src = reinterpret_cast<const uint4*>(a.frames + p * a.frame_bytes);
    uint4* dst = reinterpret_cast<uint4*>((which == 0 ? a.obs : a.next_obs) + (long)b * a.frame_bytes);
    const long n16 = a.frame_bytes >> 4;
    for (long i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    return;
  }
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const long p = a.pos[b];
  for (int j = 0; j < a.A; ++j) a.act_out[(long)b * a.A + j] = a.action[p * a.A + j];
  float r = 0.f, d = 1.f;
  for (int i = 0; i < a.nstep; ++i) {
    // one rounding per operation: the product is pinned in a register before the add (the packed-math
    // vectoriser otherwise emits v_pk_fma_f32 here, contract(off) notwithstanding)
    float t = d * a.reward[p + i];
    asm volatile("" : "+v"(t));
    r = r + t;
    float gd = a.discount[p + i] * a.gamma;
    asm volatile("" : "+v"(gd));
    d = d * gd;
  }
  a.rew_out[b] = r;
  a.disc_out[b] = d;
}

// ------------------------------------------------------------------------------------------------
// Policy output layer Linear(H, A) (drqv2.py:81,86-90), A <= 64.  Forward: one wave per row computes the A dot
// products (row in registers, W3 rows from L1), adds the bias and -- for rows >= srow0 when a noise table is
// given -- samples the action (tanh, clipped noise, straight-through clamp: utils.py:117-126) in the same pass.
// Backward: dpre = (da1+da2)(1-mu^2), then dp2 = (dpre W3)*(p2>0), dW3 = dpre^T p2, db3 = sum dpre in one kernel.
// Both replace a GEMM with N or M = action_dim (plus split-K reduce and elementwise launches).
// ------------------------------------------------------------------------------------------------
struct PolOutArgs {
  const float* h2;      // [rows][H]
  const float* w;       // [A][H]
  const float* b;       // [A]
  float* p3;            // [rows][A] pre-tanh output
  const float* noise;   // [rows - srow0][A] or null
  float* mu_out;        // [rows - srow0][A] or null
  float* a_out;         // action columns of the critic input, row stride lda_out
  long lda_out;
  int rows, H, A, srow0, use_clip;
  float std, clip;
  // optional second sampling job for rows [0, srow0): its own noise table and outputs (the actor update's action
  // for the obs rows, drqv2.py:210-211, drawn from the same policy output)
  const float* noise0;
  float* mu_out0;
  float* a_out0;
  long lda_out0;
};

__global__ void policy_out_kernel(PolOutArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  const float* h = a.h2 + (long)row * a.H;
  float mine = 0.f;
  if (a.H <= 1024) {
    // the row in 16 registers; the weights of 2 outputs (32 loads) in flight per round trip
    float hv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) hv[q] = (lane + 64 * q < a.H) ? h[lane + 64 * q] : 0.f;
    for (int j = 0; j < a.A; j += 2) {
      const int j1 = j + 1 < a.A ? j + 1 : j;
      const float* w0 = a.w + (long)j * a.H;
      const float* w1 = a.w + (long)j1 * a.H;
      float x0[16], x1[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = min(lane + 64 * q, a.H - 1);      // hv is 0 past H: the clamped weight does not matter
        x0[q] = w0[i];
        x1[q] = w1[i];
      }
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        s0 = __fmaf_rn(hv[q], x0[q], s0);
        s1 = __fmaf_rn(hv[q], x1[q], s1);
      }
      s0 = wave_sum(s0) + a.b[j];
      s1 = wave_sum(s1) + a.b[j1];
      if (lane == j) mine = s0;
      if (lane == j + 1) mine = s1;
    }
  } else {
    for (int j = 0; j < a.A; ++j) {
      const float* w = a.w + (long)j * a.H;
      float s0 = 0.f;
      for (int i = lane; i < a.H; i += 64) s0 = __fmaf_rn(h[i], w[i], s0);
      const float s = wave_sum(s0) + a.b[j];
      if (lane == j) mine = s;
    }
  }
  if (lane >= a.A) return;
  a.p3[(long)row * a.A + lane] = mine;
  const bool hi_job = a.noise && row >= a.srow0, lo_job = a.noise0 && row < a.srow0;
  if (hi_job || lo_job) {
    const long r = hi_job ? row - a.srow0 : row;
    const float* nz = hi_job ? a.noise : a.noise0;
    float* mo = hi_job ? a.mu_out : a.mu_out0;
    float* ao = hi_job ? a.a_out : a.a_out0;
    const long ld = hi_job ? a.lda_out : a.lda_out0;
    const float mu = tanhf(mine);
    float eps = nz[r * a.A + lane] * a.std;
    if (a.use_clip) eps = fminf(fmaxf(eps, -a.clip), a.clip);
    const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
    if (mo) mo[r * a.A + lane] = mu;
    ao[r * ld + lane] = fminf(fmaxf(mu + eps, lo), hi);
  }
}

struct PolBwdArgs {
  const float* da1;     // [B][ld]: action-column gradient from Q head 1 (columns col0..col0+A)
  const float* da2;
  long ld;
  int col0;
  const float* mu;      // [B][A]
  const float* p2;      // [B][H] post-ReLU input of the layer
  const float* w;       // [A][H]
  float* dp2;           // [B][H]
  float* dw;            // [A][H]
  float* db;            // [A]
  int B, H, A;
  const float* part;    // optional: da1/da2 as split-K partials [2*splitk][B][A] of the two dgrad GEMMs
  int splitk;
};

constexpr int kPolMaxA = 32;

// Workgroup = CW columns of H x NRG = 1024 / CW row groups.  dpre[B][A] (tiny) is computed once per workgroup into LDS;
// the only global stream in the row loop is p2, up to 16 rows of it in flight per thread.  CW = 16 (64 workgroups at
// hidden_dim 1024) is what the update uses: with 64 columns the launch kept 16 CUs busy with 2*A FMAs per row and
// thread -- 14.6 us at A = 6, 40 us at A = 21 (batch 256), 47 us at A = 12 (batch 512); CW = 64 remains for tiny H.
template <int AMAX, int CW>    // AMAX: action_dim rounded up to 8 / 16 / 32 (the per-output loops are unrolled over it)
__global__ __launch_bounds__(1024) void policy_out_bwd_kernel(PolBwdArgs a) {
  constexpr int NRG = 1024 / CW;
  extern __shared__ float dpre[];           // [B][A], then 4 x NRG x CW = 4096 floats of reduction scratch
  float* sm = dpre + a.B * a.A;
  const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int n = blockIdx.x * CW + c;
  const bool nok = n < a.H;
  const int nc = nok ? n : a.H - 1;
  // first batch of p2 rows: independent of dpre, requested before the dpre stage (see qout_bwd_kernel)
  float hv0[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) hv0[u] = a.p2[(long)min(rg + NRG * u, a.B - 1) * a.H + nc];
  for (int i = threadIdx.x; i < a.B * a.A; i += 1024) {
    const int m = i / a.A, j = i - m * a.A;
    const float mv = a.mu[i];
    float d1, d2;
    if (a.part) {
      const long slab = (long)a.B * a.A;
      d1 = sum_partials(a.part + i, slab, a.splitk);
      d2 = sum_partials(a.part + (long)a.splitk * slab + i, slab, a.splitk);
    } else {
      d1 = a.da1[(long)m * a.ld + a.col0 + j];
      d2 = a.da2[(long)m * a.ld + a.col0 + j];
    }
    dpre[i] = (d1 + d2) * (1.f - mv * mv);
  }
  float wn[AMAX], acc[AMAX];
#pragma unroll
  for (int j = 0; j < AMAX; ++j) {
    wn[j] = j < a.A ? a.w[(long)j * a.H + nc] : 0.f;
    acc[j] = 0.f;
  }
  __syncthreads();
  for (int m0 = rg; m0 < a.B; m0 += NRG * 16) {
    float hv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) hv[u] = m0 == rg ? hv0[u] : a.p2[(long)min(m0 + NRG * u, a.B - 1) * a.H + nc];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int m = m0 + NRG * u;
      if (m < a.B) {
        const float* dp = dpre + m * a.A;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < AMAX; ++j) {
          if (j < a.A) {
            const float dv = dp[j];
            d = __fmaf_rn(dv, wn[j], d);
            acc[j] = __fmaf_rn(dv, hv[u], acc[j]);
          }
        }
        if (nok) a.dp2[(long)m * a.H + n] = hv[u] > 0.f ? d : 0.f;
      }
    }
  }
  // dW3: the row groups of a column are summed in order through LDS, FOUR output rows per barrier pair (one at a
  // time cost two workgroup barriers per action dimension: 21 rounds for humanoid_run)
  for (int j0 = 0; j0 < a.A; j0 += 4) {
#pragma unroll
    for (int q = 0; q < AMAX; ++q)
      if (q >= j0 && q < j0 + 4 && q < a.A) sm[((q - j0) * NRG + rg) * CW + c] = acc[q];
    __syncthreads();
    if (threadIdx.x < 4 * CW) {
      const int jj = threadIdx.x / CW;                 // c = threadIdx.x % CW here as well
      if (j0 + jj < a.A && nok) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < NRG; ++g) t += sm[(jj * NRG + g) * CW + c];
        a.dw[(long)(j0 + jj) * a.H + n] = t;
      }
    }
    __syncthreads();
  }
  // db3: workgroup 0; thread t sums rows t/32, t/32 + 32, ... of dpre[:, t%32], the 32 partials are added in order
  if (blockIdx.x == 0) {
    const int jb = threadIdx.x & 31, gb = threadIdx.x >> 5;
    float t = 0.f;
    if (jb < a.A)
      for (int m = gb; m < a.B; m += 32) t += dpre[m * a.A + jb];
    sm[gb * 32 + jb] = t;
    __syncthreads();
    if (gb == 0 && jb < a.A) {
      float tot = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) tot += sm[g * 32 + jb];
      a.db[jb] = tot;
    }
  }
}

inline unsigned grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  const long cap = 8L * drq_num_cus();
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int drq_ln_tanh_fwd_multi_ex(int n, const float* const* z, int ldz, const float* const* gamma,
                             const float* const* beta, float* const* out, const int* ldo, float* const* xhat,
                             float* const* rstd, int rows, int F, const float* const* tail, const int* tail_ld,
                             int tail_n, hipStream_t st);
int drq_actor_loss_ex(const float* q1, const float* q2, const float* a, long lda, const float* mu, float std,
                      float* dq1, float* dq2, float* sums, int B, int A, float inv_global_B, float* sums_host,
                      unsigned seq, hipStream_t st);
int drq_ln_tanh_fwd_multi_part(int n, const float* const* z, int ldz, const float* const* gamma,
                               const float* const* beta, float* const* out, const int* ldo, float* const* xhat,
                               float* const* rstd, int rows, int F, const float* const* tail, const int* tail_ld,
                               int tail_n, const float* part, const float* const* bias, int splitk, hipStream_t st);
int drq_ln_tanh_bwd_part(const float* dh0, int ld0, const float* dh1, int ld1, const float* h, int ldh,
                         const float* xhat, const float* rstd, const float* gamma, float* dz, float* dln,
                         float* dgamma, float* dbeta, int rows, int F, const float* part, int splitk, int nprob,
                         int ldp, hipStream_t st);

DRQ_API int drq_aug_fwd(const uint8_t* obs, const float* shift_xy, const float* base_grid, float* out, int n, int c,
                int hw, int pad, int fuse_norm, hipStream_t st) {
  if (!obs || !shift_xy || !base_grid || !out || n <= 0 || c <= 0 || hw <= 0 || pad < 0) return DRQ_EARG;
  if (launch_aug_rows(obs, shift_xy, nullptr, nullptr, base_grid, out, n, c, hw, pad, fuse_norm, st)) {
    DRQ_LAUNCH_CHECK();
    return DRQ_OK;
  }
  const long total = (long)n * hw * hw;
  hipLaunchKernelGGL(aug_kernel<uint8_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, obs, shift_xy,
                     base_grid, out, n, c, hw, pad, fuse_norm);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// same op on a float frame (RandomShiftsAug.forward is handed obs.float(), drqv2.py:241)
DRQ_API int drq_aug_fwd_f32(const float* x, const float* shift_xy, const float* base_grid, float* out, int n, int c, int hw,
                    int pad, hipStream_t st) {
  if (!x || !shift_xy || !base_grid || !out || n <= 0 || c <= 0 || hw <= 0 || pad < 0) return DRQ_EARG;
  const long total = (long)n * hw * hw;
  hipLaunchKernelGGL(aug_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, shift_xy, base_grid,
                     out, n, c, hw, pad, 0);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_ln_tanh_fwd(const float* z, int ldz, const float* gamma, const float* beta, float* out, int ldo,
                    float* xhat, float* rstd, int rows, int F, hipStream_t st) {
  if (!z || !gamma || !beta || !out || rows <= 0 || F <= 0 || F > 256) return DRQ_EARG;
  LnArgs a{};
  a.z[0] = z; a.gamma[0] = gamma; a.beta[0] = beta; a.out[0] = out; a.xhat[0] = xhat; a.rstd[0] = rstd;
  a.ldz[0] = ldz; a.ldo[0] = ldo; a.rows = rows; a.F = F;
  hipLaunchKernelGGL(ln_tanh_fwd_kernel, dim3((rows + 3) / 4, 1), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// two LayerNorm+tanh problems of the same (rows, F) in one launch (actor trunk + critic trunk)
DRQ_API int drq_ln_tanh_fwd2(const float* z0, const float* z1, int ldz, const float* gamma0, const float* beta0,
                     const float* gamma1, const float* beta1, float* out0, int ldo0, float* out1, int ldo1,
                     float* xhat0, float* rstd0, float* xhat1, float* rstd1, int rows, int F, hipStream_t st) {
  if (!z0 || !z1 || !out0 || !out1 || rows <= 0 || F <= 0 || F > 256) return DRQ_EARG;
  LnArgs a{};
  a.z[0] = z0; a.gamma[0] = gamma0; a.beta[0] = beta0; a.out[0] = out0; a.xhat[0] = xhat0; a.rstd[0] = rstd0;
  a.z[1] = z1; a.gamma[1] = gamma1; a.beta[1] = beta1; a.out[1] = out1; a.xhat[1] = xhat1; a.rstd[1] = rstd1;
  a.ldz[0] = a.ldz[1] = ldz; a.ldo[0] = ldo0; a.ldo[1] = ldo1; a.rows = rows; a.F = F;
  hipLaunchKernelGGL(ln_tanh_fwd_kernel, dim3((rows + 3) / 4, 2), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_ln_tanh_bwd(const float* dh0, int ld0, const float* dh1, int ld1, const float* h, int ldh,
                    const float* xhat, const float* rstd, const float* gamma, float* dz, float* dln,
                    float* dgamma, float* dbeta, int rows, int F, hipStream_t st) {
  return drq_ln_tanh_bwd_part(dh0, ld0, dh1, ld1, h, ldh, xhat, rstd, gamma, dz, dln, dgamma, dbeta, rows, F, nullptr, 0,
                              0, 0, st);
}

// internal (step.hip): the incoming gradient may be the split-K partials of 1 or 2 dgrad GEMMs (see LnBwdArgs)
int drq_ln_tanh_bwd_part(const float* dh0, int ld0, const float* dh1, int ld1, const float* h, int ldh,
                         const float* xhat, const float* rstd, const float* gamma, float* dz, float* dln,
                         float* dgamma, float* dbeta, int rows, int F, const float* part, int splitk, int nprob,
                         int ldp, hipStream_t st) {
  // dgamma == dbeta == nullptr: the caller computes the parameter gradients elsewhere (from dln and xhat: the rider
  // workgroup of the trunk weight-gradient launch, or drq_ln_param_grad)
  if ((!dh0 && !part) || !h || !xhat || !rstd || !gamma || !dz || !dln || (!dgamma != !dbeta) || rows <= 0 || F <= 0 ||
      F > 256)
    return DRQ_EARG;
  if (part && (splitk < 1 || nprob < 1 || nprob > 2 || ldp < F)) return DRQ_EARG;
  LnBwdArgs a{dh0, dh1, h, xhat, rstd, gamma, dz, dln, ld0, ld1, ldh, rows, F, part, splitk, nprob, ldp};
  hipLaunchKernelGGL(ln_tanh_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  if (dgamma) {
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3(F), dim3(256), 0, st, (const float*)dln, xhat, dgamma, dbeta, rows, F);
    DRQ_LAUNCH_CHECK();
  }
  return DRQ_OK;
}

// internal (step.hip): the parameter gradients alone
int drq_ln_param_grad(const float* dln, const float* xhat, float* dgamma, float* dbeta, int rows, int F, hipStream_t st) {
  if (!dln || !xhat || !dgamma || !dbeta || rows <= 0 || F <= 0 || F > 256) return DRQ_EARG;
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3(F), dim3(256), 0, st, dln, xhat, dgamma, dbeta, rows, F);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// nz (<= 8) Q-head output layers in one launch; host arrays of device pointers
DRQ_API int drq_qout_fwd(int nz, const float* const* h, const float* const* w, const float* const* b, float* const* q, int B,
                 int H, hipStream_t st) {
  if (nz <= 0 || nz > 8 || !h || !w || !b || !q || B <= 0 || H <= 0) return DRQ_EARG;
  QOutArgs a{};
  for (int z = 0; z < nz; ++z) {
    if (!h[z] || !w[z] || !b[z] || !q[z]) return DRQ_EARG;
    a.h[z] = h[z]; a.w[z] = w[z]; a.b[z] = b[z]; a.q[z] = q[z];
  }
  a.B = B; a.H = H;
  hipLaunchKernelGGL(qout_fwd_kernel, dim3((B + 3) / 4, nz), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // extern "C"
namespace {
// 64 columns per workgroup below 1,024 rows, 16 beyond (see qout_bwd_kernel)
void launch_qout_bwd(const QOutBwdArgs& a, int nz, size_t lds, hipStream_t st) {
  if (a.B >= 1024) hipLaunchKernelGGL(qout_bwd_kernel<16>, dim3((a.H + 15) / 16, nz), dim3(1024), lds, st, a);
  else hipLaunchKernelGGL(qout_bwd_kernel<64>, dim3((a.H + 63) / 64, nz), dim3(1024), lds, st, a);
}
}  // namespace
extern "C" {

// dh = (dq w^T) * (h > 0);  dw = dq^T h, db = sum dq when the dw/db arrays are given
DRQ_API int drq_qout_bwd(int nz, const float* const* dq, const float* const* h, const float* const* w, float* const* dh,
                 float* const* dw, float* const* db, int B, int H, hipStream_t st) {
  if (nz <= 0 || nz > 8 || !dq || !h || !w || !dh || B <= 0 || H <= 0) return DRQ_EARG;
  QOutBwdArgs a{};
  for (int z = 0; z < nz; ++z) {
    if (!dq[z] || !h[z] || !w[z] || !dh[z]) return DRQ_EARG;
    a.dq[z] = dq[z]; a.h[z] = h[z]; a.w[z] = w[z]; a.dh[z] = dh[z];
    a.dw[z] = dw ? dw[z] : nullptr;
    a.db[z] = db ? db[z] : nullptr;
  }
  a.B = B; a.H = H;
  const size_t lds = ((size_t)B + 1024 + 64) * sizeof(float);
  if (lds > 60 * 1024) return DRQ_EARG;
  launch_qout_bwd(a, nz, lds, st);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// internal (step.hip): the twin-Q output layer backward with the TD loss (drq_td_mse) computed inside it:
// dq1/dq2 never exist in memory, sums[0..4] are written by the same launch
int drq_qout_bwd_td(const float* tq1, const float* tq2, const float* q1, const float* q2, const float* reward,
                    const float* discount, float inv_global_B, float* sums, const float* const* h,
                    const float* const* w, float* const* dh, float* const* dw, float* const* db, int B, int H,
                    hipStream_t st) {
  if (!tq1 || !tq2 || !q1 || !q2 || !reward || !discount || !sums || !h || !w || !dh || B <= 0 || H <= 0)
    return DRQ_EARG;
  QOutBwdArgs a{};
  for (int z = 0; z < 2; ++z) {
    if (!h[z] || !w[z] || !dh[z]) return DRQ_EARG;
    a.h[z] = h[z]; a.w[z] = w[z]; a.dh[z] = dh[z];
    a.dw[z] = dw ? dw[z] : nullptr;
    a.db[z] = db ? db[z] : nullptr;
  }
  a.B = B; a.H = H;
  a.td = 1; a.tq1 = tq1; a.tq2 = tq2; a.q1 = q1; a.q2 = q2; a.reward = reward; a.discount = discount;
  a.invB = inv_global_B; a.sums = sums;
  const size_t lds = ((size_t)B + 5 * 1024 + 64) * sizeof(float);      // the sums tree of workgroup (0,0) needs 5 x 1024
  if (lds > 60 * 1024) return DRQ_EARG;
  launch_qout_bwd(a, 2, lds, st);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// internal (step.hip, single-GPU schedule): the same backward (input gradient only) with the actor loss (drq_actor_loss)
// computed inside it: sums[5..6] and the host mirror are written by workgroup (0, 0) of this launch
int drq_qout_bwd_actor(const float* q1, const float* q2, const float* act, long lda, const float* mu, float std, int A,
                       float inv_global_B, float* sums, float* sums_host, unsigned seq, const float* const* h,
                       const float* const* w, float* const* dh, int B, int H, hipStream_t st) {
  if (!q1 || !q2 || !act || !mu || !sums || !h || !w || !dh || B <= 0 || H <= 0 || A <= 0) return DRQ_EARG;
  QOutBwdArgs a{};
  for (int z = 0; z < 2; ++z) {
    if (!h[z] || !w[z] || !dh[z]) return DRQ_EARG;
    a.h[z] = h[z]; a.w[z] = w[z]; a.dh[z] = dh[z];
  }
  a.B = B; a.H = H;
  a.td = 2; a.q1 = q1; a.q2 = q2; a.invB = inv_global_B; a.sums = sums;
  a.act = act; a.lda = lda; a.mu = mu; a.A = A; a.std = std; a.sums_host = sums_host; a.seq = seq;
  const size_t lds = ((size_t)B + 5 * 1024 + 64) * sizeof(float);
  if (lds > 60 * 1024) return DRQ_EARG;
  launch_qout_bwd(a, 2, lds, st);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// n (<= 4) LayerNorm+tanh problems of the same (rows, F) in one launch
DRQ_API int drq_ln_tanh_fwd_multi(int n, const float* const* z, int ldz, const float* const* gamma, const float* const* beta,
                          float* const* out, const int* ldo, float* const* xhat, float* const* rstd, int rows, int F,
                          hipStream_t st) {
  return drq_ln_tanh_fwd_multi_ex(n, z, ldz, gamma, beta, out, ldo, xhat, rstd, rows, F, nullptr, nullptr, 0, st);
}

// internal form used by the step: problem i may also copy tail_n (<= 64) columns of tail[i] behind its F outputs
int drq_ln_tanh_fwd_multi_ex(int n, const float* const* z, int ldz, const float* const* gamma,
                             const float* const* beta, float* const* out, const int* ldo, float* const* xhat,
                             float* const* rstd, int rows, int F, const float* const* tail, const int* tail_ld,
                             int tail_n, hipStream_t st) {
  return drq_ln_tanh_fwd_multi_part(n, z, ldz, gamma, beta, out, ldo, xhat, rstd, rows, F, tail, tail_ld, tail_n,
                                    nullptr, nullptr, 0, st);
}

// ... and with the pre-norm input given as split-K partials of the trunk GEMM (z may then be null)
int drq_ln_tanh_fwd_multi_part(int n, const float* const* z, int ldz, const float* const* gamma,
                               const float* const* beta, float* const* out, const int* ldo, float* const* xhat,
                               float* const* rstd, int rows, int F, const float* const* tail, const int* tail_ld,
                               int tail_n, const float* part, const float* const* bias, int splitk, hipStream_t st) {
  if (n <= 0 || n > 4 || (!z && !part) || !gamma || !beta || !out || !ldo || rows <= 0 || F <= 0 || F > 256)
    return DRQ_EARG;
  if (tail && (tail_n <= 0 || tail_n > 64 || !tail_ld)) return DRQ_EARG;
  if (part && splitk < 1) return DRQ_EARG;
  LnArgs a{};
  a.part = part;
  a.splitk = splitk;
  a.tail_n = tail ? tail_n : 0;
  for (int i = 0; i < n && tail; ++i) {
    a.tail[i] = tail[i];
    a.tail_ld[i] = tail_ld[i];
  }
  for (int i = 0; i < n; ++i) {
    if ((!part && !z[i]) || !gamma[i] || !beta[i] || !out[i]) return DRQ_EARG;
    a.z[i] = z ? z[i] : nullptr; a.gamma[i] = gamma[i]; a.beta[i] = beta[i]; a.out[i] = out[i];
    a.bias[i] = (part && bias) ? bias[i] : nullptr;
    a.xhat[i] = xhat ? xhat[i] : nullptr;
    a.rstd[i] = rstd ? rstd[i] : nullptr;
    a.ldz[i] = ldz; a.ldo[i] = ldo[i];
  }
  a.rows = rows; a.F = F;
  hipLaunchKernelGGL(ln_tanh_fwd_kernel, dim3((rows + 3) / 4, n), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_colsum(const float* dy, long ld, long dy_bs, float* out, long out_bs, int M, int N, int nbatch,
               hipStream_t st) {
  if (!dy || !out || M <= 0 || N <= 0 || nbatch <= 0) return DRQ_EARG;
  if (M >= 1024)
    hipLaunchKernelGGL(colsum_kernel<16>, dim3((N + 15) / 16, nbatch), dim3(1024), 0, st, dy, ld, dy_bs, out, out_bs, M, N);
  else
    hipLaunchKernelGGL(colsum_kernel<64>, dim3((N + 63) / 64, nbatch), dim3(1024), 0, st, dy, ld, dy_bs, out, out_bs, M, N);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_trunc_normal_sample(const float* pre_tanh, const float* noise, float std, float clip, int use_clip,
                            float* mu_out, float* a_out, long lda_out, int B, int A, hipStream_t st) {
  if (!pre_tanh || !noise || !a_out || B <= 0 || A <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(sample_action_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, pre_tanh, noise, std, clip,
                     use_clip, mu_out, a_out, lda_out, B, A);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// internal (step.hip): policy output layer forward (+ optional action sampling) and backward, see the kernels
int drq_policy_out_fwd(const float* h2, const float* w, const float* b, float* p3, int rows, int H, int A,
                       const float* noise, float std, float clip, int use_clip, int srow0, float* mu_out,
                       float* a_out, long lda_out, const float* noise0, float* mu_out0, float* a_out0, long lda_out0,
                       hipStream_t st) {
  if (!h2 || !w || !b || !p3 || rows <= 0 || H <= 0 || A <= 0 || A > 64) return DRQ_EARG;
  if (noise && (!a_out || srow0 < 0 || srow0 > rows)) return DRQ_EARG;
  if (noise0 && (!a_out0 || !noise)) return DRQ_EARG;
  PolOutArgs a{h2, w, b, p3, noise, mu_out, a_out, lda_out, rows, H, A, srow0, use_clip, std, clip,
               noise0, mu_out0, a_out0, lda_out0};
  hipLaunchKernelGGL(policy_out_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

int drq_policy_out_bwd(const float* da1, const float* da2, long ld, int col0, const float* mu, const float* p2,
                       const float* w, float* dp2, float* dw, float* db, int B, int H, int A, const float* part,
                       int splitk, hipStream_t st) {
  if (((!da1 || !da2) && !part) || !mu || !p2 || !w || !dp2 || !dw || !db || B <= 0 || H <= 0 || A <= 0 ||
      A > kPolMaxA)
    return DRQ_EARG;
  if (part && splitk < 1) return DRQ_EARG;
  const size_t lds = ((size_t)B * A + 4 * 16 * 64) * sizeof(float);
  if (lds > 60 * 1024) return DRQ_EARG;
  PolBwdArgs a{da1, da2, ld, col0, mu, p2, w, dp2, dw, db, B, H, A, part, splitk};
  if (H >= 256) {          // 16 columns per workgroup
    const dim3 g((H + 15) / 16);
    if (A <= 8) hipLaunchKernelGGL((policy_out_bwd_kernel<8, 16>), g, dim3(1024), lds, st, a);
    else if (A <= 16) hipLaunchKernelGGL((policy_out_bwd_kernel<16, 16>), g, dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((policy_out_bwd_kernel<32, 16>), g, dim3(1024), lds, st, a);
  } else {
    const dim3 g((H + 63) / 64);
    if (A <= 8) hipLaunchKernelGGL((policy_out_bwd_kernel<8, 64>), g, dim3(1024), lds, st, a);
    else if (A <= 16) hipLaunchKernelGGL((policy_out_bwd_kernel<16, 64>), g, dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((policy_out_bwd_kernel<32, 64>), g, dim3(1024), lds, st, a);
  }
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_copy_cols(const float* src, int ld_src, float* dst, long ld_dst, int B, int A, hipStream_t st) {
  if (!src || !dst || B <= 0 || A <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, src, ld_src, dst, ld_dst, B, A);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_td_mse(const float* tq1, const float* tq2, const float* q1, const float* q2, const float* reward,
               const float* discount, float* dq1, float* dq2, float* sums, int B, float inv_global_B,
               hipStream_t st) {
  if (!tq1 || !tq2 || !q1 || !q2 || !reward || !discount || !dq1 || !dq2 || !sums || B <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(256), 0, st, tq1, tq2, q1, q2, reward, discount, dq1, dq2, sums, B,
                     inv_global_B);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// internal form used by the step: also publishes the eight sums to the pinned host mirror (DrqStep.sums_host)
int drq_actor_loss_ex(const float* q1, const float* q2, const float* a, long lda, const float* mu, float std,
                      float* dq1, float* dq2, float* sums, int B, int A, float inv_global_B, float* sums_host,
                      unsigned seq, hipStream_t st) {
  if (!q1 || !q2 || !a || !mu || !dq1 || !dq2 || !sums || B <= 0 || A <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(256), 0, st, q1, q2, a, lda, mu, std, dq1, dq2, sums, B, A,
                     inv_global_B, sums_host, seq);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_actor_loss(const float* q1, const float* q2, const float* a, long lda, const float* mu, float std,
                   float* dq1, float* dq2, float* sums, int B, int A, float inv_global_B, hipStream_t st) {
  return drq_actor_loss_ex(q1, q2, a, lda, mu, std, dq1, dq2, sums, B, A, inv_global_B, nullptr, 0u, st);
}

DRQ_API int drq_actor_dmu(const float* dha1, const float* dha2, long ld, int col0, const float* mu, float* dpre, int B,
                  int A, hipStream_t st) {
  if (!dha1 || !dha2 || !mu || !dpre || B <= 0 || A <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(actor_dmu_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, dha1, dha2, ld, col0, mu, dpre,
                     B, A);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// step = 1-based Adam step count; bias corrections are formed in double on the host exactly as
// torch/optim/adam.py does.  tgt != null fuses  tgt <- tau*p_new + (1-tau)*tgt.
// two independent segments (encoder, actor) in one launch: same arithmetic per element as adam_kernel
int drq_adam_flat2(float* p0, const float* g0, float* m0, float* v0, long n0, long step0, float* p1, const float* g1,
                   float* m1, float* v1, long n1, long step1, double lr, float gscale, hipStream_t st) {
  if (!p0 || !g0 || !m0 || !v0 || !p1 || !g1 || !m1 || !v1 || n0 <= 0 || n1 <= 0 || step0 <= 0 || step1 <= 0)
    return DRQ_EARG;
  AdamSeg2 a{};
  const long steps[2] = {step0, step1};
  float* ps[2] = {p0, p1}; const float* gs[2] = {g0, g1}; float* ms[2] = {m0, m1}; float* vs[2] = {v0, v1};
  const long ns[2] = {n0, n1};
  for (int i = 0; i < 2; ++i) {
    const double bc1 = 1.0 - pow(0.9, (double)steps[i]);
    const double bc2 = 1.0 - pow(0.999, (double)steps[i]);
    a.p[i] = ps[i]; a.g[i] = gs[i]; a.m[i] = ms[i]; a.v[i] = vs[i]; a.n[i] = ns[i];
    a.neg_step_size[i] = (float)(-(lr / bc1));
    a.sqrt_bc2[i] = (float)sqrt(bc2);
  }
  a.gscale = gscale;
  const long nmax = n0 > n1 ? n0 : n1;
  hipLaunchKernelGGL(adam2_kernel, dim3(grid_for(nmax), 2), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_adam_flat(float* p, const float* g, float* m, float* v, long n, double lr, long step, float gscale,
                  float* tgt, double tau, hipStream_t st) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return DRQ_EARG;
  const double bc1 = 1.0 - pow(0.9, (double)step);
  const double bc2 = 1.0 - pow(0.999, (double)step);
  const float neg_step_size = (float)(-(lr / bc1));
  const float sqrt_bc2 = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, neg_step_size, sqrt_bc2,
                     gscale, tgt, (float)tau, (float)(1.0 - tau));
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_adam_reduce_flat(float* p, const float* recv, long stride, int world, float* m, float* v, long n, double lr,
                                 long step, float gscale, hipStream_t st) {
  if (!p || !recv || !m || !v || n <= 0 || step <= 0 || world < 1 || stride < n) return DRQ_EARG;
  const double bc1 = 1.0 - pow(0.9, (double)step);
  const double bc2 = 1.0 - pow(0.999, (double)step);
  hipLaunchKernelGGL(adam_reduce_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, recv, stride, world, m, v, n,
                     (float)(-(lr / bc1)), (float)sqrt(bc2), gscale);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_sum_slices(const float* in, long stride, int world, float* out, long n, hipStream_t st) {
  if (!in || !out || n <= 0 || world < 1 || stride < n) return DRQ_EARG;
  hipLaunchKernelGGL(sum_slices_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, stride, world, out, n);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_ema_flat(const float* p, float* t, long n, double tau, hipStream_t st) {
  if (!p || !t || n <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, t, n, (float)tau, (float)(1.0 - tau));
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// Device-side replay batch assembly (replay_buffer.py:142-160), see nstep_gather_kernel.
DRQ_API int drq_nstep_gather(const uint8_t* frames, const float* action, const float* reward, const float* discount,
                     const long* pos, int B, int A, long frame_bytes, int nstep, float gamma, uint8_t* obs,
                     float* act_out, float* rew_out, float* disc_out, uint8_t* next_obs, hipStream_t st) {
  if (!action || !reward || !discount || !pos || !act_out || !rew_out || !disc_out) return DRQ_EARG;
  if ((obs != nullptr) != (next_obs != nullptr)) return DRQ_EARG;
  if (obs && !frames) return DRQ_EARG;
  if (B <= 0 || A <= 0 || nstep <= 0 || frame_bytes <= 0 || frame_bytes % 16) return DRQ_EARG;
  if (((uintptr_t)frames | (uintptr_t)obs | (uintptr_t)next_obs) & 15) return DRQ_EARG;
  NstepArgs a{frames, action, reward, discount, pos, obs, next_obs, act_out, rew_out, disc_out, frame_bytes, B, A, nstep,
              gamma};
  // obs == next_obs == NULL: action rows and n-step reward / discount only (the frames stay in the store and the
  // update reads them through DrqStep.obs_index / next_obs_index)
  if (obs) hipLaunchKernelGGL(nstep_gather_kernel, dim3(B, 3), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(nstep_gather_kernel, dim3((B + 255) / 256, 1), dim3(256), 0, st, a);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// Encoder input conversion without augmentation (drqv2.py:64 on a uint8 frame, as act() does)
DRQ_API int drq_u8_normalize(const uint8_t* x, float* y, long n, hipStream_t st) {
  if (!x || !y || n <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(u8_normalize_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_tanh(const float* x, float* y, long n, hipStream_t st) {
  if (!x || !y || n <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(tanh_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

DRQ_API int drq_fill(float* p, long n, float v, hipStream_t st) {
  if (!p || n <= 0) return DRQ_EARG;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, n, v);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

}  // extern "C"
