// U = G g G^T of the 1024 (cout, cin) 3x3 filters of a 32->32 layer, written in the MFMA-lane order that
// conv3x3_wino_kernel reads: image[((pos*8 + c)*64 + k4*16 + (oc&15))*2 + (oc>>4)], pos = 4*i + j of the 4x4
// transform, reduction channel ic = 4*c + k4.  wmode 0: forward (oc = cout, ic = cin); 1: input gradient (oc = cin,
// ic = cout, taps flipped).  Shared by the kernel's own prologue (image in LDS) and by the riders of
// conv1_aug_kernel that prepare the six images of an update in global memory (conv1aug.hip).
#pragma once

// `stage`: >= 32*289 floats of LDS; `img`: 16384 floats (LDS or global; may alias `stage` when SYNC_BEFORE_WRITE).
// 256 threads.  Contains __syncthreads(): every thread of the workgroup must call it.
template <bool ALIASED>
__device__ __forceinline__ void wino_u_image(const float* __restrict__ w, int wmode, float* stage, float* img, int tid) {
#pragma clang fp contract(off)
  // The canonical weights are copied into LDS first (contiguous global reads; rows of 32 filters at a pitch of 289
  // floats), every thread then pulls the nine taps of its four filters into registers.  Lanes walk the OUTPUT
  // channel: the staging reads are conflict-free in both gather modes (row pitch 289 / filter pitch 9, both odd) and
  // the image writes land in 32 distinct banks per half-wave (a lane-ordered gather from global memory costs
  // several us per workgroup, and so did image writes that walked the input channel: 32 lanes on one bank).
  constexpr int SP = 289;
  for (int i = tid; i < 32 * 288; i += 256) stage[(i / 288) * SP + (i % 288)] = w[i];
  __syncthreads();
  float g[4][9];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = it * 256 + tid;
    const int oc = q & 31, ic = q >> 5;
    const float* src = wmode == 0 ? stage + oc * SP + ic * 9 : stage + ic * SP + oc * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) g[it][t] = src[wmode == 0 ? t : 8 - t];
  }
  if (ALIASED) __syncthreads();            // the image overwrites the staging area
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = it * 256 + tid;
    const int oc = q & 31, ic = q >> 5;
    float tm[4][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = g[it][kx], g1 = g[it][3 + kx], g2 = g[it][6 + kx];
      tm[0][kx] = g0;
      tm[1][kx] = 0.5f * ((g0 + g2) + g1);
      tm[2][kx] = 0.5f * ((g0 + g2) - g1);
      tm[3][kx] = g2;
    }
    const int c = ic >> 2, k4 = ic & 3, h = oc >> 4;
    float* dst = img + ((size_t)c * 64 + k4 * 16 + (oc & 15)) * 2 + h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t0 = tm[i][0], t1 = tm[i][1], t2 = tm[i][2];
      dst[(i * 4 + 0) * (8 * 64 * 2)] = t0;
      dst[(i * 4 + 1) * (8 * 64 * 2)] = 0.5f * ((t0 + t2) + t1);
      dst[(i * 4 + 2) * (8 * 64 * 2)] = 0.5f * ((t0 + t2) - t1);
      dst[(i * 4 + 3) * (8 * 64 * 2)] = t2;
    }
  }
}
