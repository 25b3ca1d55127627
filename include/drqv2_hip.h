/* libdrqv2_hip.so -- C ABI of the MI355X-native DrQ-v2 update path.
 *
 * The reference (johannah/drqv2) has no native boundary of its own: DrQV2Agent.update()
 * (drqv2.py:230-262) reaches ATen/cuDNN through torch.nn / torch.optim.  Each entry point below
 * replaces the ATen kernels behind the cited reference lines; the Python host (drqv2.py, utils.py at
 * the repo root, drqv2_amd/) binds them with ctypes.  Conventions:
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (PyTorch-ROCm);
 *   - every call is enqueued on `stream` (a hipStream_t, passed as void*), nothing synchronises;
 *   - return 0 = ok, <0 = argument/shape/workspace error, >0 = hipError_t of the launch;
 *   - no allocation and no state that outlives a call, except two host-side caches indexed by the current HIP
 *     device: the CU count, and "dynamic-LDS attribute already set" flags of the weight-gradient kernels;
 *   - the library is built with -fvisibility=hidden: the functions declared here are ALL it exports
 *     (tests/test_cpu_interface.py checks both directions); development ablations and time-stamp hooks exist only
 *     in the -DDRQ_DEV build that tools/ makes for itself (drqv2_amd.build --dev -> libdrqv2_hip_dev.so);
 *   - tensors are fp32 row-major / NCHW unless stated.
 */
#ifndef DRQV2_HIP_H
#define DRQV2_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* drq_stream_t; /* hipStream_t */

#define DRQ_OK 0
#define DRQ_EARG (-1)
#define DRQ_EWS (-2)

int drq_abi_version(void);

/* ---- RandomShiftsAug.forward (drqv2.py:19-45) [+ Encoder's obs/255-0.5, drqv2.py:64, if fuse_norm]
 * obs u8 [n][c][hw][hw]; shift_xy f32 [n][2] = the torch.randint(0,2*pad+1,(n,1,1,2)) draw (x,y);
 * base_grid f32 [hw] = linspace(-1+1/S,1-1/S,S)[:hw]; out f32 [n][c][hw][hw]. */
int drq_aug_fwd(const uint8_t* obs, const float* shift_xy, const float* base_grid, float* out, int n, int c,
                int hw, int pad, int fuse_norm, drq_stream_t stream);
int drq_aug_fwd_f32(const float* x, const float* shift_xy, const float* base_grid, float* out, int n, int c,
                    int hw, int pad, drq_stream_t stream);

/* ---- RandomShiftsAug of BOTH views (drqv2.py:241-242) + obs/255-0.5 (:64) + the first encoder layer
 * Conv2d(9,32,3,stride 2)+ReLU (:55) in one launch that reads the uint8 frames once (frame_stack 3, 84x84, pad 4).
 * obs / obs1 u8 [n][9][84][84] (4-byte aligned), shift / shift1 f32 [n][2], base_grid f32 [84], w [32][9][3][3],
 * bias [32].  y f32 [2n][32][41][41]: frames [0,n) = obs view, [n,2n) = obs1 view.  xaug f32 [2n][9][84][84]:
 * the augmented, normalised encoder input of frames [0, n_store) is also written there (the update stores the
 * obs view, n_store = n, for conv1's weight gradient; tests store both views; 0 = none, xaug may be NULL).
 * Values are bit-identical to drq_aug_fwd(fuse_norm=1) followed by drq_conv3x3_fwd. */
int drq_conv1_aug_fwd(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                      const float* base_grid, const float* w, const float* bias, float* xaug, float* y, int n,
                      int n_store, drq_stream_t stream);
/* the same with the layer's products on the bf16 MFMA (the bf16 update path, see the bf16 conv entries below): the
 * augmentation and the stored encoder input are the fp32 ones, bit for bit; y = relu(conv(bf16(x), bf16(w)) + b). */
/* the same with the two views given as rows idx[b] / idx1[b] of frame stores (a device-resident replay ring): the batch
 * of replay_buffer.py:142-160 is never materialised; results are bit-identical to drq_conv1_aug_fwd on the gathered
 * frames. */
int drq_conv1_aug_fwd_indexed(const uint8_t* frames, const int64_t* idx, const float* shift, const uint8_t* frames1,
                              const int64_t* idx1, const float* shift1, const float* base_grid, const float* w,
                              const float* bias, float* xaug, float* y, int n, int n_store, drq_stream_t stream);
int drq_conv1_aug_fwd_bf16(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                           const float* base_grid, const float* w, const float* bias, float* xaug, float* y, int n,
                           int n_store, drq_stream_t stream);
/* the same with y in the bf16 [2n][41][41][32 channels] layout of the bf16 update (see drq_conv3x3_fwd_bf16_nhwc):
 * the values of drq_conv1_aug_fwd_bf16's y, rounded to bf16 */
int drq_conv1_aug_fwd_bf16_nhwc(const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                                const float* base_grid, const float* w, const float* bias, float* xaug, void* y_nhwc,
                                int n, int n_store, drq_stream_t stream);

/* ---- Encoder conv layers (drqv2.py:55-59): Conv2d(cin,32,3,stride)+ReLU, 32 output channels.
 * Supported (cin,hin,stride): (9,84,2) (32,41,1) (32,39,1) (32,37,1).  y element (b,co,oy,ox) is
 * written at y[y_off + b*y_bs + co*y_cs + oy*y_rs + ox]. */
int drq_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int nb, int cin, int hin,
                    int stride, int relu, long y_bs, long y_cs, long y_rs, long y_off, drq_stream_t stream);
/* autograd of the same layers.  dy_pad = gradient w.r.t. the pre-activation, stored zero-padded by 2:
 * [nb][32][hout+4][hout+4].  dx (size hout+2) = conv_transpose(dy, w) * (mask > 0), strided store. */
int drq_conv3x3_dgrad(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout,
                      long dx_bs, long dx_cs, long dx_rs, long dx_off, drq_stream_t stream);
/* The 32->32 layers (hin 41/39/37, stride 1) and their input gradient in Winograd F(2x2,3x3) form: same contracts as
 * drq_conv3x3_fwd / drq_conv3x3_dgrad, 2.25x fewer matrix FLOPs, fp32 throughout; the result differs from the
 * direct form by rounding only (same error level against fp64).  This is what DrQV2Agent.update runs. */
int drq_conv3x3_fwd_wino(const float* x, const float* w, const float* bias, float* y, int nb, int hin, int relu,
                         long y_bs, long y_cs, long y_rs, long y_off, drq_stream_t stream);
int drq_conv3x3_dgrad_wino(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout,
                           long dx_bs, long dx_cs, long dx_rs, long dx_off, drq_stream_t stream);
/* ... and their weight / bias gradient in the same form: the contract of drq_conv3x3_wgrad for cin = 32, stride 1, with
 * ONE more precondition: dy must be the interior of a ZERO-PADDED buffer (the kernel reads row and column `hout` of
 * every plane as the empty half of the last 2x2 tile), i.e. dy_rs >= hout+1, dy_cs >= (hout+1)*dy_rs and zeros there --
 * the update's [nb][32][hout+4][hout+4] gradient buffers (pad 2) satisfy it.  A contiguous dy returns DRQ_EARG. */
int drq_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, float* db, int nb, int hin, long dy_bs,
                           long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes, drq_stream_t stream);
/* dw [32][cin][3][3], db [32]; dy addressed as dy[dy_off + b*dy_bs + co*dy_cs + oy*dy_rs + ox]. */
int drq_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int nb, int cin, int hin,
                      int stride, long dy_bs, long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes,
                      drq_stream_t stream);
size_t drq_conv3x3_wgrad_ws_bytes(void);

/* ---- bf16-MFMA variants of the 32->32 channel layers (conv2..4), BASELINE configs[4] ("bf16").  New functionality:
 * the reference is fp32 only.  Same arguments, layouts and fp32 storage as the entries above; every MFMA operand is
 * rounded to bf16 (nearest even) when staged, products are exact, accumulation is fp32: the result is the
 * fp32-accumulated convolution of the bf16-rounded operands.  hin in {41,39,37} (fwd, wgrad); hout+4 in {39,41,43}. */
int drq_conv3x3_fwd_bf16(const float* x, const float* w, const float* bias, float* y, int nb, int hin, int relu,
                         long y_bs, long y_cs, long y_rs, long y_off, drq_stream_t stream);
int drq_conv3x3_dgrad_bf16(const float* dy_pad, const float* w, const float* mask, float* dx, int nb, int hout,
                           long dx_bs, long dx_cs, long dx_rs, long dx_off, drq_stream_t stream);
int drq_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw, float* db, int nb, int hin, long dy_bs,
                           long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes, drq_stream_t stream);
/* The same three launches with activations in the layout the bf16 update keeps between the encoder layers (round 3):
 * bf16 [frame][y][x][32 channels], 64 bytes per pixel, 16-byte aligned.  A value stored in that layout is the bf16
 * rounding the entries above apply when they stage it, in the same place of the same sums, so results are identical
 * bit for bit.  fwd: x in that layout when x_nhwc (else fp32 NCHW), y in that layout when y_nhwc (else contiguous fp32
 * NCHW); dgrad: the mask in that layout ([nb][hout+2][hout+2][32]); dy_nhwc: dy_pad is bf16 [nb][hout+4][hout+4][32]
 * with its zero border; dx_nhwc: dx is the INTERIOR of a bf16 [nb][hout+6][hout+6][32] buffer padded by 2 whose
 * border the caller keeps zero (the strides are ignored); wgrad: x in that layout; dy_nhwc: dy is the base of such a
 * padded bf16 [nb][hin+2][hin+2][32] buffer (strides ignored) and db sums the bf16 values. */
int drq_conv3x3_fwd_bf16_nhwc(const void* x, const float* w, const float* bias, void* y, int nb, int hin, int relu,
                              int x_nhwc, int y_nhwc, drq_stream_t stream);
int drq_conv3x3_dgrad_bf16_nhwc(const void* dy_pad, const float* w, const void* mask_nhwc, void* dx, int nb, int hout,
                                int dy_nhwc, int dx_nhwc, long dx_bs, long dx_cs, long dx_rs, long dx_off,
                                drq_stream_t stream);
int drq_conv3x3_wgrad_bf16_nhwc(const void* x_nhwc, const void* dy, float* dw, float* db, int nb, int hin, int dy_nhwc,
                                long dy_bs, long dy_cs, long dy_rs, long dy_off, float* ws, size_t ws_bytes,
                                drq_stream_t stream);

/* ---- nn.Linear forward / backward (drqv2.py:74-81,100-111) as one strided, batched GEMM:
 *   C[b][m][n] = epi( sum_k A_b(m,k) * B_b(k,n) ),  epi(v) = relu?(v + bias[n]) * (aux[m][n] > 0)?
 *   a_kc: A(m,k)=A[m*lda+k] else A[k*lda+m];  b_kc: B(k,n)=B[n*ldb+k] else B[k*ldb+n].
 *   scatter_hw>0: C index = zero-padded (pad 2) NCHW gradient layout of a [M][32*hw*hw] feature map.
 *   tile: 0 auto, 1 = 32x32, 2 = 64x64 block tile; splitk: 0 auto (needs ws). */
int drq_gemm_f32(const float* A, long lda, int a_kc, const float* B, long ldb, int b_kc, float* C, long ldc,
                 int M, int N, int K, int nbatch, long a_bs, long b_bs, long c_bs, const float* bias, long bias_bs,
                 int relu, const float* aux, int ldaux, long aux_bs, int scatter_hw, int tile, int splitk,
                 float* ws, size_t ws_bytes, drq_stream_t stream);

/* same GEMM for nbatch (<= 8) problems of one shape with independent pointers (host arrays of device
 * pointers; bias / aux / rowsum arrays may be NULL).  rowsum[b] != NULL (needs a_kc == 0): also writes
 * rowsum[b][m] = sum_k A_b(m,k), i.e. the bias gradient of a wgrad GEMM, without a second pass. */
int drq_gemm_batched_f32(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                         int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                         const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int tile,
                         int splitk, float* ws, size_t ws_bytes, drq_stream_t stream);

/* bf16-MFMA variant of drq_gemm_batched_f32 (BASELINE configs[4]; new functionality, see the conv entries): fp32
 * storage, operands rounded to bf16 when staged, fp32 accumulation; rowsum sums the unrounded values. */
int drq_gemm_batched_bf16(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                          int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                          const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int splitk,
                          float* ws, size_t ws_bytes, drq_stream_t stream);

/* Forward form only (both operands k-contiguous, a_kc = b_kc = 1) with the split-K sum left to the caller: when
 * *splitk_out > 1 the result is ws[(b*splitk + s)*M*N + m*N + n], s < splitk, WITHOUT bias (the consumer, e.g. the
 * LayerNorm kernel of the trunk, sums the records and adds the bias itself); *splitk_out == 1: C holds the result
 * with bias.  This is the entry DrQV2Agent.update uses for the trunk Linear(39200 -> feature_dim) (drqv2.py:74,100). */
int drq_gemm_batched_partial(int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                             int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias,
                             float* ws, size_t ws_bytes, int* splitk_out, drq_stream_t stream);

/* ---- hidden layers of the policy / Q MLPs, nn.Linear(hidden, hidden)+ReLU (drqv2.py:77-81,103-111) on the LDS-DMA
 * ring kernel (csrc/gemm3.hip): fp32, nbatch (<= 8) problems of one shape, every dimension a multiple of 64 (K of
 * the forward / dgrad forms: of 32), 16-byte aligned operands, leading dimensions multiples of 4.  DRQ_EARG = shape
 * not eligible (callers fall back to drq_gemm_batched_f32).
 * drq_mlp_fwd: y_b [M][ldy] = relu?(x_b [M][ldx] w_b[N][ldw]^T + bias_b).  qw / qpart (both or neither): the weight
 * row [N] of a FOLLOWING Linear(N, 1) (the Q heads' output layer, drqv2.py:106,111): qpart_b [M][*nq_out] receives the
 * partial dots  sum_{n in column tile j} y_b[m][n] qw_b[n];  that layer's output is their sum in index order + bias.
 * drq_mlp_dgrad: dx_b [M][lddx] = (dy_b [M][lddy] w_b [K][ldw]) * (mask_b [M][ldmask] > 0)?   (N columns).
 * drq_mlp_wgrad_dgrad: both gradients of one layer in ONE launch: dw_b [Nout][Kin] = dy_b^T x_b, db_b [Nout] =
 * column sums of dy_b (db may be NULL), dx_b [Brows][lddx] = (dy_b w_b [Nout][ldw]) * (mask_b > 0)?. */
int drq_mlp_fwd(int nbatch, const float* const* x, long ldx, const float* const* w, long ldw, float* const* y,
                long ldy, int M, int N, int K, const float* const* bias, int relu, const float* const* qw,
                float* const* qpart, int* nq_out, drq_stream_t stream);
int drq_mlp_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* w, long ldw, float* const* dx,
                  long lddx, int M, int N, int K, const float* const* mask, int ldmask, drq_stream_t stream);
int drq_mlp_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                        float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                        long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin,
                        drq_stream_t stream);

/* ---- row-local stages fused with the first MLP layer that consumes them (csrc/rowblock.hip), fp32.
 * drq_ln_l1_fwd: njobs (<= 4) problems  h_j = tanh(LayerNorm(z_j)) (drqv2.py:74-75,100-101; eps 1e-5, F <= 256), with
 * z_j given either as z[j] [rows_j][F] (splitk == 0) or as the split-K records of drq_gemm_batched_partial (splitk > 0:
 * element (s, row, f) of problem j at part[j][(s*slab) + row*F + f], plus bias[j][f]); out[j] [rows_j][ldo_j] receives
 * h_j in columns [0,F) and, if tail[j] is given, tail[j] [rows_j][tail_ld_j] in columns [F, F+tail_n_j) (the critic's
 * [h, action] input, drqv2.py:117); xhat[j] / rstd[j] (optional) as drq_ln_tanh_fwd writes them -- all bit-identical
 * to drq_ln_tanh_fwd.  nheads[j] in {0,1,2}: y[2j+h] [rows_j][H] = relu(out_j[:, :F+tail_n_j] w[2j+h]^T + b[2j+h]),
 * w [H][F+tail_n_j] (16-byte aligned), the first layers of the policy / Q MLPs (drqv2.py:77,103,108); F+tail_n <= 128.
 * drq_policy_out_l1_fwd: p3 [rows][A] = p2 [rows][H] w3[A][H]^T + b3 (drqv2.py:81; H % 256 == 0, A <= 32); rows >=
 * srow0 ("hi", the next_obs rows): mu = tanh(p3), a' = clampST(mu + clamp(noise_hi*std, +-clip)) (utils.py:117-126)
 * written to columns [F, F+A) of ha_hi [rows-srow0][lda_hi] (mu_hi optional), and with nheads == 2
 * y[h] [rows-srow0][H] = relu([ha_hi[:, :F], a'] w[h]^T + b[h]); rows < srow0 ("lo"): the same sample with noise_lo into
 * ha_lo / mu_lo when noise_lo is given. */
int drq_ln_l1_fwd(int njobs, const float* const* part, const float* const* z, const float* const* bias,
                  const float* const* gamma, const float* const* beta, float* const* out, const int* ldo,
                  float* const* xhat, float* const* rstd, const float* const* tail, const int* tail_ld,
                  const int* tail_n, const int* rows, const int* nheads, const float* const* w, const float* const* b,
                  float* const* y, int F, int H, int splitk, long slab, drq_stream_t stream);
int drq_policy_out_l1_fwd(const float* p2, const float* w3, const float* b3, float* p3, int rows, int srow0, int H, int A,
                          int F, float std, float clip, int use_clip, const float* noise_hi, float* mu_hi, float* ha_hi,
                          long lda_hi, const float* noise_lo, float* mu_lo, float* ha_lo, long lda_lo, int nheads,
                          const float* const* w, const float* const* b, float* const* y, drq_stream_t stream);

/* ---- output layer of the Q heads, nn.Linear(hidden, 1) (drqv2.py:106,111), nz (<= 8) problems per launch:
 * q = h w^T + b;  backward: dh = (dq w) * (h > 0), and if dw/db are given dw = dq^T h, db = sum dq. */
int drq_qout_fwd(int nz, const float* const* h, const float* const* w, const float* const* b, float* const* q, int B,
                 int H, drq_stream_t stream);
int drq_qout_bwd(int nz, const float* const* dq, const float* const* h, const float* const* w, float* const* dh,
                 float* const* dw, float* const* db, int B, int H, drq_stream_t stream);

/* ---- nn.LayerNorm(F)+nn.Tanh (drqv2.py:74-75,100-101), eps 1e-5, F <= 256 */
int drq_ln_tanh_fwd(const float* z, int ldz, const float* gamma, const float* beta, float* out, int ldo,
                    float* xhat, float* rstd, int rows, int F, drq_stream_t stream);
int drq_ln_tanh_fwd2(const float* z0, const float* z1, int ldz, const float* gamma0, const float* beta0,
                     const float* gamma1, const float* beta1, float* out0, int ldo0, float* out1, int ldo1,
                     float* xhat0, float* rstd0, float* xhat1, float* rstd1, int rows, int F, drq_stream_t stream);
int drq_ln_tanh_fwd_multi(int n, const float* const* z, int ldz, const float* const* gamma, const float* const* beta,
                          float* const* out, const int* ldo, float* const* xhat, float* const* rstd, int rows, int F,
                          drq_stream_t stream);
int drq_ln_tanh_bwd(const float* dh0, int ld0, const float* dh1, int ld1, const float* h, int ldh,
                    const float* xhat, const float* rstd, const float* gamma, float* dz, float* dln,
                    float* dgamma, float* dbeta, int rows, int F, drq_stream_t stream);

/* ---- bias gradients: out[b][n] = sum_m dy[b][m][n] */
int drq_colsum(const float* dy, long ld, long dy_bs, float* out, long out_bs, int M, int N, int nbatch,
               drq_stream_t stream);

/* ---- Actor head + utils.TruncatedNormal.sample (drqv2.py:88-92, utils.py:112-126):
 * mu = tanh(pre_tanh); a = clamp(mu + clamp(noise*std, +-clip), +-(1-1e-6)); a -> a_out[b*lda_out + j]. */
int drq_trunc_normal_sample(const float* pre_tanh, const float* noise, float std, float clip, int use_clip,
                            float* mu_out, float* a_out, long lda_out, int B, int A, drq_stream_t stream);
int drq_copy_cols(const float* src, int ld_src, float* dst, long ld_dst, int B, int A, drq_stream_t stream);

/* ---- update_critic loss (drqv2.py:185-189): y = r + d*min(tq1,tq2); dq_k = 2(q_k-y)*inv_global_B;
 * sums[0..4] = sum r, sum y, sum q1, sum q2, sum((q1-y)^2+(q2-y)^2) over the LOCAL batch. */
int drq_td_mse(const float* tq1, const float* tq2, const float* q1, const float* q2, const float* reward,
               const float* discount, float* dq1, float* dq2, float* sums, int B, float inv_global_B,
               drq_stream_t stream);
/* ---- update_actor loss (drqv2.py:212-216,225): sums[5] = sum -min(q1,q2), sums[6] = sum log_prob */
int drq_actor_loss(const float* q1, const float* q2, const float* a, long lda, const float* mu, float std,
                   float* dq1, float* dq2, float* sums, int B, int A, float inv_global_B, drq_stream_t stream);
int drq_actor_dmu(const float* dha1, const float* dha2, long ld, int col0, const float* mu, float* dpre, int B,
                  int A, drq_stream_t stream);

/* ---- torch.optim.Adam.step (drqv2.py:148-150,201-202,221; defaults) over a flat arena; optional fused
 * utils.soft_update_params (utils.py:42-45) into tgt.  g is multiplied by gscale first (1 = exact). */
int drq_adam_flat(float* p, const float* g, float* m, float* v, long n, double lr, long step, float gscale,
                  float* tgt, double tau, drq_stream_t stream);
/* ---- data-parallel exchanges on a fully connected xGMI node (new): the `world` ranks' copies of this rank's
 * slice of a gradient bucket, as an all-to-all delivers them (copy r at in[r*stride + i]), are added in RANK ORDER (the
 * same order everywhere: every rank ends with bit-identical sums).  drq_sum_slices writes the sum; drq_adam_reduce_flat
 * (ZeRO-1: sharded optimiser state) feeds it straight into torch.optim.Adam's arithmetic (drq_adam_flat's, bit for bit)
 * for the n parameters this rank owns -- the summed gradient never returns to memory. */
int drq_sum_slices(const float* in, long stride, int world, float* out, long n, drq_stream_t stream);
int drq_adam_reduce_flat(float* p, const float* recv, long stride, int world, float* m, float* v, long n, double lr,
                         long step, float gscale, drq_stream_t stream);
int drq_ema_flat(const float* p, float* t, long n, double tau, drq_stream_t stream);
int drq_fill(float* p, long n, float v, drq_stream_t stream);
/* obs/255-0.5 on raw uint8 frames (drqv2.py:64 as reached from act(), drqv2.py:165-166); y = tanh(x) */
int drq_u8_normalize(const uint8_t* x, float* y, long n, drq_stream_t stream);

/* ---- replay batch assembly on the device (replay_buffer.py:142-160 `_sample`) ----
 * A flat store of steps in HBM (each episode contiguous, its first step the dummy reset transition):
 * frames [n][frame_bytes] u8 (frame_bytes % 16 == 0), action [n][A], reward [n], discount [n].
 * Row b of the batch is the transition at store index pos[b] (device array, the reference's `idx`):
 *   obs = frames[pos-1], action = action[pos], next_obs = frames[pos+nstep-1],
 *   reward/discount = the n-step accumulation in the reference's float32 order
 *   (reward += discount*r[pos+i]; discount *= d[pos+i]*gamma). */
int drq_nstep_gather(const uint8_t* frames, const float* action, const float* reward, const float* discount,
                     const long* pos, int B, int A, long frame_bytes, int nstep, float gamma, uint8_t* obs,
                     float* act_out, float* rew_out, float* disc_out, uint8_t* next_obs, drq_stream_t stream);
int drq_tanh(const float* x, float* y, long n, drq_stream_t stream);

/* ---- the four random draws of one update in one launch, bit-identical to the ATen launches of the reference's calls
 * (torch.randint(0, range, (B,1,1,2), dtype=float32) x2 from drqv2.py:34,241-242; torch.empty((B,A)).normal_() x2 from
 * utils.py:135 via drqv2.py:183,211): Philox4x32-10, key = seed, subsequence = element index, offsets offset + 0, 4, 8,
 * 12.  (seed, offset) = torch's CUDA generator state before the draws; the caller advances its offset by 16.
 * n_shift = 2*B, n_noise = B*A, each <= 65536 (else DRQ_EARG: ATen's launch geometry differs there).  n_noise = 0: the
 * two shift draws only (offsets + 0, 4; the caller advances by 8 and draws the noises itself). */
int drq_rng_draws(unsigned long long seed, unsigned long long offset, int n_shift, int n_noise, int range,
                  float* shift_obs, float* shift_next, float* noise_critic, float* noise_actor, drq_stream_t stream);

/* ---- whole-step entry: DrQV2Agent.update (drqv2.py:230-262) ------------------------------------ */
typedef struct {
  int B, global_B, C, A, F, H;
  const uint8_t* obs;        /* [B][C][84][84] */
  const uint8_t* next_obs;
  const float* action;       /* [B][A] */
  const float* reward;       /* [B] */
  const float* discount;     /* [B] */
  const float* shift_obs;    /* [B][2] */
  const float* shift_next;
  const float* noise_critic; /* [B][A] */
  const float* noise_actor;
  const float* base_grid;    /* [84] */
  float* params;             /* arenas laid out by drq_param_layout */
  float* grads;
  float* adam_m;
  float* adam_v;
  float* ws;                 /* drq_step_ws_bytes(), zero-initialised once */
  size_t ws_bytes;
  float* sums;               /* [8] local partial sums of the metrics */
  double lr, tau;
  float std, clip;
  long step_critic, step_enc, step_actor; /* 1-based Adam step numbers of THIS update */
  float gscale;              /* multiplies every gradient inside Adam.  The loss kernels already scale local
                              * gradients by 1/global_B, so a SUM all-reduce needs gscale = 1 (what the host
                              * passes); 1/world_size is only for hosts that feed per-rank MEAN gradients. */
  drq_stream_t stream;
  float* sums_host;          /* optional (may be NULL): device-visible pinned host memory, 16 floats.  As soon as
                              * sums[0..7] are final (after the actor loss, BEFORE the actor backward / Adam /
                              * Polyak tail of the update) phase 1 writes them to sums_host[0..7], then stores the
                              * 32-bit value (uint32_t)step_actor in slot 8 with system-scope release.  The host
                              * reads the metrics (drqv2.py:191-198,218-223: the .item() calls) by polling slot 8
                              * instead of draining the stream.  Single-GPU only: with data parallelism the sums
                              * are partial until the host has all-reduced them. */
  int store_aug_next;        /* 0: only the obs view's augmented encoder input is kept in the AUG workspace buffer (rows
                              * [0,B); conv1's weight gradient reads it).  1 (verification): the next_obs view is
                              * stored as well (rows [B,2B)), 65 MB more traffic at B=256. */
  int bf16;                  /* 0: fp32 everywhere (the reference's arithmetic).  1 (BASELINE configs[4], new functionality):
                              * conv2..4 forward / dgrad / wgrad and the nn.Linear GEMMs of the update run on the bf16
                              * MFMA (operands rounded to bf16 when staged, fp32 accumulation) -- except two products of
                              * the trunk layer that are bound by memory, not arithmetic, and keep their faster fp32
                              * kernels: its input gradient (always) and its weight gradient below batch 512; conv1 (fused with
                              * the augmentation) multiplies on the bf16 MFMA too; storage, the augmentation arithmetic,
                              * conv1's weight gradient, LayerNorm, the output heads, losses, Adam and
                              * Polyak stay fp32.  act() always runs in fp32. */
  void* const* timing_events; /* optional (may be NULL): host array of timing_n hipEvent_t created with timing enabled.
                              * Instrumentation for bench.py's roofline: pairs recorded on `stream` right before / after
                              * [0],[1] the conv2 forward launch of phase 3; [2],[3] the conv3 input-gradient launch of
                              * phase 5; [4],[5] the launch(es) of the conv2..4 weight gradients; [6],[7] phase 4 (the
                              * heads of the critic update: trunks .. trunk input gradient); [8],[9] phases 6-7 without
                              * the critic's optimiser step (the heads of the actor update).  Pairs beyond timing_n
                              * are not recorded. */
  int timing_n;              /* entries of timing_events (0, 4, 6, 8 or 10) */
  const int64_t* obs_index;      /* optional (both or neither): the batch is NOT materialised -- `obs` / `next_obs` are */
  const int64_t* next_obs_index; /* stores of frames (a device replay ring, [slots][C][84][84] u8) and row b of the batch
                              * is frame obs_index[b] of `obs` / next_obs_index[b] of `next_obs` (replay_buffer.py:152-153:
                              * idx-1 and idx+nstep-1).  The fused aug+conv1 launch gathers its source rows straight from
                              * the store. */
  int flags;                 /* schedule switches for A/B measurements and for tests that hold both forms to the oracle:
                              * DRQ_STEP_NO_ROW_FUSION (1): LayerNorm / policy output layer / first MLP layers as separate
                              * launches (the round-2 schedule) instead of csrc/rowblock.hip's fused ones;
                              * DRQ_STEP_NO_GEMM3 (2): hidden-layer gradients on the round-2 kernels instead of
                              * csrc/gemm3.hip;
                              * DRQ_STEP_BF16_FP32_ACTS (4, bf16 only): the outputs of conv1..conv3 stay fp32 NCHW between
                              * the layers (round 2's storage) instead of bf16 [frame][y][x][32] -- the same update bit
                              * for bit, more memory traffic;
                              * DRQ_STEP_BF16_FP32_GRADS (8, bf16 only): the gradients handed from one encoder input
                              * gradient to the next (of conv3's and conv2's outputs) stay fp32 instead of bf16 in that
                              * layout -- the same weight gradients bit for bit; the two bias gradients they feed then
                              * sum unrounded values.  A workspace must be zeroed when these two bits change between
                              * updates (the zero borders of the padded buffers move).  0 = the production schedule. */
} DrqStep;
#define DRQ_STEP_NO_ROW_FUSION 1
#define DRQ_STEP_NO_GEMM3 2
#define DRQ_STEP_BF16_FP32_ACTS 4
#define DRQ_STEP_BF16_FP32_GRADS 8

/* Parameter arena: tensors in parameters() order of encoder, critic, actor, critic_target, each start
 * aligned to 64 floats, each network's segment padded to a multiple of 512 floats (a segment is one optimiser
 * launch and one data-parallel exchange bucket: it splits into 2/4/8 equal slices of whole lines).  out[] receives, in this order: encoder 8 offsets, critic 16, actor 10,
 * critic_target 16, then [enc_beg, enc_end, critic_beg, critic_end, actor_beg, actor_end, target_beg,
 * target_end], then total.  Returns the number of longs written (59) or <0. */
int drq_param_layout(int C, int A, int F, int H, long* out, int cap);
#define DRQ_PARAM_LAYOUT_LEN 59

size_t drq_step_ws_bytes(int B, int C, int A, int F, int H);
/* offsets (in floats) of named buffers inside ws, for tests and tools; ids below. */
long drq_step_ws_offset(int B, int C, int A, int F, int H, int buffer_id);
enum {
  DRQ_WS_AUG = 0, DRQ_WS_ACT1, DRQ_WS_ACT2, DRQ_WS_ACT3, DRQ_WS_FEAT, DRQ_WS_Z_NEXT, DRQ_WS_Z_OBS,
  DRQ_WS_HA_T, DRQ_WS_HA_C, DRQ_WS_H_AN, DRQ_WS_H_AO, DRQ_WS_Q, DRQ_WS_TQ, DRQ_WS_DQ, DRQ_WS_MU_O,
  DRQ_WS_DY4, DRQ_WS_DY3, DRQ_WS_DY2, DRQ_WS_DY1, DRQ_WS_DZ_C, DRQ_WS_DZ_A, DRQ_WS_HA_C2,
  DRQ_WS_P1, DRQ_WS_P2,   /* policy hidden activations (post-ReLU), [2B][H]: rows [0,B) obs, [B,2B) next_obs */
  DRQ_WS_C1, DRQ_WS_C2,   /* critic Q hidden activations (post-ReLU) of the critic loss, [2 heads][B][H] */
  DRQ_WS_NBUF_PUBLIC
};

/* One update = phases 3..9 in this order (phase -1 runs them all: single GPU):
 *   3  aug + encoder forward (drqv2.py:241-246)         reads the encoder weights only
 *   4  trunks, policy, Q heads, TD loss, backward down to the encoder output (:180-200)
 *      leaves ALL critic gradients and sums[0..4]; first reader of the actor weights
 *   5  encoder backward                                  leaves the encoder gradients
 *   6  Adam(critic) + Polyak (:201,:259-260), actor loss through the updated critic (:210-216)
 *      leaves sums[5..6]; publishes sums to sums_host when that is set
 *   7  actor backward (:218-220)                         leaves the actor gradients
 *   8  Adam(encoder) (:202)   commutes to here: phases 6/7 work on features encoded before it (:255)
 *   9  Adam(actor) (:221)
 * Composite ids kept for callers that exchange at coarser points: 0 = 3,4,5; 1 = 6,7; 2 = 8,9.
 * For hosts that follow the reference's METHOD boundaries (DrQV2Agent.update_critic / update_actor, drqv2.py:177-228)
 * phase 6 is also available in pieces: 10 = Adam(critic) alone (no Polyak), 11 = the actor loss of phase 6 without the
 * optimiser step, 12 = Polyak alone (utils.soft_update_params), 13 = re-draw the actor update's action from the policy
 * output stored by phase 4 with s->noise_actor (update_critic does not know that draw yet).  update_critic = 4, 5, 10, 8;
 * update_actor = 13, 11, 7, 9; results equal phase -1 bit for bit (tests/test_hip_step.py).
 * A data-parallel host SUM-all-reduces the critic gradients while 5 runs, the encoder gradients while 6/7 run,
 * the metric sums after 6 and the actor gradients after 7; 8 may be deferred until the next update's phase 3
 * and 9 until its phase 4 (or drq_act_forward), so both exchanges overlap compute. */
int drq_update_phase(const DrqStep* s, int phase);

/* sums[0..7] -> sums_host[0..7], then seq -> slot 8 with system-scope release (see DrqStep.sums_host); for hosts
 * that reduce the sums themselves before publishing them. */
int drq_publish_sums(const float* sums, float* sums_host, unsigned seq, drq_stream_t stream);

/* Encoder+actor forward for DrQV2Agent.act (drqv2.py:164-175): obs u8 [n][C][84][84] -> mu [n][A]
 * (n <= 2*B; uses s->params, s->ws, s->stream only; must not run between the phases of an update). */
int drq_act_forward(const DrqStep* s, const uint8_t* obs, int n, float* mu_out);

#ifdef __cplusplus
}
#endif
#endif
