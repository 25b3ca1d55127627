// Whole-step orchestration of DrQV2Agent.update (drqv2.py:230-262) on one HIP stream.
// Host code only: sequences the kernels of conv.hip / gemm.hip / elementwise.hip over a caller-owned
// workspace.  No allocation, no synchronisation, no global state.
#include "common.h"
#include "../../include/drqv2_hip.h"

// elementwise.hip (internal): LayerNorm that sums the split-K partials of drq_gemm_batched_partial
extern "C" int drq_ln_tanh_fwd_multi_part(int n, const float* const* z, int ldz, const float* const* gamma,
                                          const float* const* beta, float* const* out, const int* ldo,
                                          float* const* xhat, float* const* rstd, int rows, int F,
                                          const float* const* tail, const int* tail_ld, int tail_n, const float* part,
                                          const float* const* bias, int splitk, hipStream_t st);
// elementwise.hip (internal)
extern "C" int drq_qout_bwd_td(const float* tq1, const float* tq2, const float* q1, const float* q2, const float* reward,
                               const float* discount, float inv_global_B, float* sums, const float* const* h,
                               const float* const* w, float* const* dh, float* const* dw, float* const* db, int B,
                               int H, hipStream_t st);
extern "C" int drq_qout_bwd_actor(const float* q1, const float* q2, const float* act, long lda, const float* mu, float std,
                                    int A, float inv_global_B, float* sums, float* sums_host, unsigned seq,
                                    const float* const* h, const float* const* w, float* const* dh, int B, int H,
                                    hipStream_t st);
extern "C" int drq_policy_out_fwd(const float* h2, const float* w, const float* b, float* p3, int rows, int H, int A,
                                  const float* noise, float std, float clip, int use_clip, int srow0, float* mu_out,
                                  float* a_out, long lda_out, const float* noise0, float* mu_out0, float* a_out0,
                                  long lda_out0, hipStream_t st);
extern "C" int drq_policy_out_bwd(const float* da1, const float* da2, long ld, int col0, const float* mu,
                                  const float* p2, const float* w, float* dp2, float* dw, float* db, int B, int H,
                                  int A, const float* part, int splitk, hipStream_t st);
extern "C" int drq_ln_tanh_bwd_part(const float* dh0, int ld0, const float* dh1, int ld1, const float* h, int ldh,
                                    const float* xhat, const float* rstd, const float* gamma, float* dz, float* dln,
                                    float* dgamma, float* dbeta, int rows, int F, const float* part, int splitk,
                                    int nprob, int ldp, hipStream_t st);
extern "C" int drq_ln_tanh_fwd_multi_ex(int n, const float* const* z, int ldz, const float* const* gamma,
                                        const float* const* beta, float* const* out, const int* ldo,
                                        float* const* xhat, float* const* rstd, int rows, int F,
                                        const float* const* tail, const int* tail_ld, int tail_n, hipStream_t st);
extern "C" int drq_actor_loss_ex(const float* q1, const float* q2, const float* a, long lda, const float* mu, float std,
                                 float* dq1, float* dq2, float* sums, int B, int A, float inv_global_B,
                                 float* sums_host, unsigned seq, hipStream_t st);
extern "C" int drq_adam_flat2(float* p0, const float* g0, float* m0, float* v0, long n0, long step0, float* p1,
                              const float* g1, float* m1, float* v1, long n1, long step1, double lr, float gscale,
                              hipStream_t st);
// gemm.hip (internal): either precision (bf16 != 0: bf16-MFMA kernel, fp32 storage)
int drq_gemm_batched_any(int bf16, int nbatch, const float* const* A, long lda, int a_kc, const float* const* B, long ldb,
                         int b_kc, float* const* C, long ldc, int M, int N, int K, const float* const* bias, int relu,
                         const float* const* aux, int ldaux, float* const* rowsum, int scatter_hw, int tile,
                         int splitk, float* ws, size_t ws_bytes, hipStream_t st);
int drq_gemm_batched_partial_any(int bf16, int nbatch, const float* const* A, long lda, int a_kc, const float* const* B,
                                 long ldb, int b_kc, float* const* C, long ldc, int M, int N, int K,
                                 const float* const* bias, float* ws, size_t ws_bytes, int* splitk_out, hipStream_t st);
// conv1aug.hip (internal): bf_mma selects the bf16-MFMA form of the layer's products
int drq_conv1_aug_fwd_any(int bf_mma, const uint8_t* obs, const float* shift, const uint8_t* obs1, const float* shift1,
                          const float* base_grid, const float* w, const float* bias, float* xaug, float* y, int n,
                          int n_store, hipStream_t st, const float* const* wino_w, float* wino_u, const long* fidx0,
                          const long* fidx1);
// conv_wino.hip (internal): the Winograd kernels with the layer's U image prepared by conv1_aug_kernel's rider
int drq_conv3x3_fwd_wino_pre(const float* x, const float* w, const float* u_image, const float* bias, float* y, int nb,
                             int hin, int relu, long y_bs, long y_cs, long y_rs, long y_off, hipStream_t st);
int drq_conv3x3_dgrad_wino_pre(const float* dy_pad, const float* w, const float* u_image, const float* mask, float* dx,
                               int nb, int hout, long dx_bs, long dx_cs, long dx_rs, long dx_off, hipStream_t st);
// gemm2.hip (internal): the trunk weight gradient with the LayerNorm parameter gradients riding in the same launch
int drq_trunk_wgrad_ln(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
                       float* rowsum, const float* ln_dln, const float* ln_xhat, float* ln_dgamma, float* ln_dbeta,
                       int ln_rows, int ln_F, hipStream_t st);
extern "C" int drq_ln_param_grad(const float* dln, const float* xhat, float* dgamma, float* dbeta, int rows, int F,
                                 hipStream_t st);
// gemm2.hip (internal): weight gradient + input gradient of one hidden layer in one launch
int drq_gemm2_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                          float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                          long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin, hipStream_t st);
// gemm3.hip (internal): the hidden x hidden layers on the LDS-DMA ring kernel (DRQ_EARG = shape not eligible)
int drq_gemm3_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* w, long ldw, float* const* dx,
                    long lddx, int M, int N, int K, const float* const* mask, int ldmask, hipStream_t st);
int drq_gemm3_wgrad_dgrad(int nbatch, const float* const* dy, long lddy, const float* const* x, long ldx,
                          float* const* dw, float* const* db, const float* const* w, long ldw, float* const* dx,
                          long lddx, const float* const* mask, int ldmask, int Brows, int Nout, int Kin, hipStream_t st);
// rowblock.hip (internal): LayerNorm+tanh fused with the first MLP layers; policy output layer + sample fused with the
// target critic's first layers (DRQ_EARG = shape not eligible)
int drq_lnl1_fwd(int njobs, const float* const* part, const float* const* z, const float* const* bias,
                 const float* const* gamma, const float* const* beta, float* const* out, const int* ldo,
                 float* const* xhat, float* const* rstd, const float* const* tail, const int* tail_ld, const int* tail_n,
                 const int* rows, const int* nheads, const float* const* w, const float* const* b, float* const* y, int F,
                 int H, int splitk, long slab, hipStream_t st);
int drq_polout_l1_fwd(const float* p2, const float* w3, const float* b3, float* p3, int rows, int srow0, int H, int A,
                      int F, float std, float clip, int use_clip, const float* noise_hi, float* mu_hi, float* ha_hi,
                      long lda_hi, const float* noise_lo, float* mu_lo, float* ha_lo, long lda_lo, int nheads,
                      const float* const* w, const float* const* b, float* const* y, hipStream_t st);
// conv_bf16.hip (internal): the bf16 launches with activations in bf16 [frame][y][x][32] where the flags say so
int drq_conv3x3_fwd_bf16_lay(const void* x, const float* w, const float* bias, void* y, int nb, int hin, int relu,
                             long y_bs, long y_cs, long y_rs, long y_off, int lay, hipStream_t st);
int drq_conv3x3_dgrad_bf16_lay(const void* dy_pad, const float* w, const void* mask, void* dx, int nb, int hout,
                               long dx_bs, long dx_cs, long dx_rs, long dx_off, int lay, hipStream_t st);
int drq_conv3x3_wgrad_partial_bf16_lay(const void* x, const void* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                       long dy_off, float* part, size_t part_bytes, int* nblocks, int lay,
                                       hipStream_t st);
int drq_conv3x3_wgrad_partial_bf16(const float* x, const float* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                   long dy_off, float* part, size_t part_bytes, int* nblocks, hipStream_t st);
// conv.hip (internal)
int drq_conv3x3_wgrad_partial(const float* x, const float* dy, int nb, int cin, int hin, int stride, long dy_bs,
                              long dy_cs, long dy_rs, long dy_off, float* part, size_t part_bytes, int* nblocks,
                              hipStream_t st);
// conv_wino_wgrad.hip (internal): the 32->32 layers' records in Winograd form (same format, same reduction)
int drq_conv3x3_wgrad_partial_wino(const float* x, const float* dy, int nb, int hin, long dy_bs, long dy_cs, long dy_rs,
                                   long dy_off, float* part, size_t part_bytes, int* nblocks, hipStream_t st);
int drq_conv3x3_wgrad_partial_wino3(const float* const* x, const float* const* dy, int nb, const long* dy_bs,
                                    const long* dy_cs, const long* dy_rs, const long* dy_off, float* const* part,
                                    size_t part_bytes, int* nblocks, hipStream_t st);
int drq_conv3x3_wgrad_reduce_multi(int n, const float* const* part, const int* nblocks, const int* cin,
                                   float* const* dw, float* const* db, hipStream_t st);

namespace {

constexpr long R = 32L * 35 * 35;   // repr_dim (drqv2.py:53)
inline long al64(long x) { return (x + 63) & ~63L; }
// segment (= optimiser / exchange bucket) boundaries: multiples of 512 floats, so that every bucket splits into
// 2, 4 or 8 equal slices of whole 64-float lines (reduce-scatter style exchanges, sharded optimiser steps)
inline long al512(long x) { return (x + 511) & ~511L; }

struct HeadOff {
  long trunk_w, trunk_b, ln_g, ln_b;
  long w[2][3], b[2][3];
};
struct ParamLayout {
  long enc_w[4], enc_b[4];
  HeadOff critic, actor, target;
  long seg[8];   // enc, critic, actor, target [beg,end)
  long total;
  long flat[50];
  int nflat;
};

ParamLayout param_layout(int C, int A, int F, int H) {
  ParamLayout L{};
  long off = 0;
  int nf = 0;
  auto take = [&](long n) {
    const long o = off;
    off = al64(off + n);
    L.flat[nf++] = o;
    return o;
  };
  L.seg[0] = off;
  for (int l = 0; l < 4; ++l) {
    L.enc_w[l] = take(32L * (l == 0 ? C : 32) * 9);
    L.enc_b[l] = take(32);
  }
  off = al512(off);
  L.seg[1] = off;
  auto head = [&](HeadOff& h, int nq, int in_extra, int out_last) {
    h.trunk_w = take((long)F * R);
    h.trunk_b = take(F);
    h.ln_g = take(F);
    h.ln_b = take(F);
    for (int q = 0; q < nq; ++q) {
      h.w[q][0] = take((long)H * (F + in_extra));
      h.b[q][0] = take(H);
      h.w[q][1] = take((long)H * H);
      h.b[q][1] = take(H);
      h.w[q][2] = take((long)out_last * H);
      h.b[q][2] = take(out_last);
    }
  };
  L.seg[2] = off;
  head(L.critic, 2, A, 1);
  off = al512(off);
  L.seg[3] = off;
  L.seg[4] = off;
  head(L.actor, 1, 0, A);
  off = al512(off);
  L.seg[5] = off;
  L.seg[6] = off;
  head(L.target, 2, A, 1);
  off = al512(off);
  L.seg[7] = off;
  L.total = off;
  L.nflat = nf;
  return L;
}

struct WsLayout {
  long off[64];
  long total;
};
enum {
  W_AUG = DRQ_WS_AUG, W_ACT1 = DRQ_WS_ACT1, W_ACT2 = DRQ_WS_ACT2, W_ACT3 = DRQ_WS_ACT3, W_FEAT = DRQ_WS_FEAT,
  W_Z_NEXT = DRQ_WS_Z_NEXT, W_Z_OBS = DRQ_WS_Z_OBS, W_HA_T = DRQ_WS_HA_T, W_HA_C = DRQ_WS_HA_C,
  W_H_AN = DRQ_WS_H_AN, W_H_AO = DRQ_WS_H_AO, W_Q = DRQ_WS_Q, W_TQ = DRQ_WS_TQ, W_DQ = DRQ_WS_DQ,
  W_MU_O = DRQ_WS_MU_O, W_DY4 = DRQ_WS_DY4, W_DY3 = DRQ_WS_DY3, W_DY2 = DRQ_WS_DY2, W_DY1 = DRQ_WS_DY1,
  W_DZ_C = DRQ_WS_DZ_C, W_DZ_A = DRQ_WS_DZ_A, W_HA_C2 = DRQ_WS_HA_C2,
  W_P1 = DRQ_WS_P1, W_P2 = DRQ_WS_P2,             // policy activations over 2B rows (obs, next)
  W_C1 = DRQ_WS_C1, W_C2 = DRQ_WS_C2,             // critic-Q hidden activations
  W_XHAT_C = DRQ_WS_NBUF_PUBLIC, W_RSTD_C, W_XHAT_A, W_RSTD_A, W_Z_C2,
  W_HROWS, W_P3, W_Z4,                            // actor trunk output / policy output over 2B rows
  W_T1, W_T2,                                     // target-Q hidden activations, reused by the actor step
  W_DC2, W_DC1, W_DHA, W_DLN, W_DPRE, W_DP2, W_DP1, W_DH_A, W_DA,
  W_GEMM_WS, W_CONV_WS,
  W_WINO_U,                                       // six Winograd weight images of the update: [layer 2..4][fwd, dgrad][16384]
  W_COUNT
};

WsLayout ws_layout(int B, int C, int A, int F, int H) {
  WsLayout w{};
  long off = 0;
  auto take = [&](int id, long n) {
    w.off[id] = off;
    off = al64(off + n);
  };
  const long B2 = 2L * B;
  take(W_AUG, B2 * C * 84 * 84);
  take(W_ACT1, B2 * 32 * 41 * 41);
  take(W_ACT2, B2 * 32 * 39 * 39);
  take(W_ACT3, B2 * 32 * 37 * 37);
  take(W_FEAT, B2 * R);
  take(W_Z_NEXT, (long)B * 2 * F);
  take(W_Z_OBS, (long)B * 2 * F);
  take(W_HA_T, (long)B * (F + A));
  take(W_HA_C, (long)B * (F + A));
  take(W_H_AN, (long)B * F);
  take(W_H_AO, (long)B * F);
  take(W_Q, 2L * B);
  take(W_TQ, 2L * B);
  take(W_DQ, 2L * B);
  take(W_MU_O, (long)B * A);
  take(W_DY4, (long)B * 32 * 39 * 39);
  take(W_DY3, (long)B * 32 * 41 * 41);
  take(W_DY2, (long)B * 32 * 43 * 43);
  take(W_DY1, (long)B * 32 * 45 * 45);
  take(W_DZ_C, (long)B * F);
  take(W_DZ_A, (long)B * F);
  take(W_HA_C2, (long)B * (F + A));
  take(W_XHAT_C, (long)B * F);
  take(W_RSTD_C, B);
  take(W_XHAT_A, (long)B * F);
  take(W_RSTD_A, B);
  take(W_Z_C2, (long)B * F);
  take(W_HROWS, 2L * B * F);
  take(W_P1, 2L * B * H);
  take(W_P2, 2L * B * H);
  take(W_P3, 2L * B * A);
  take(W_Z4, 4L * B * F);
  take(W_T1, 2L * B * H);
  take(W_T2, 2L * B * H);
  take(W_C1, 2L * B * H);
  take(W_C2, 2L * B * H);
  take(W_DC2, 2L * B * H);
  take(W_DC1, 2L * B * H);
  take(W_DHA, 2L * B * (F + A));
  take(W_DLN, (long)B * F);
  take(W_DPRE, (long)B * A);
  take(W_DP2, (long)B * H);
  take(W_DP1, (long)B * H);
  take(W_DH_A, (long)B * F);
  take(W_DA, 2L * B * A);
  // split-K partials: the widest user is the trunk forward (2 nets x B x F x splits) and the
  // H x H weight gradients; 64 MiB covers every shape the step issues (checked per call).
  take(W_GEMM_WS, 16L * 1024 * 1024);
  take(W_CONV_WS, (long)(drq_conv3x3_wgrad_ws_bytes() / sizeof(float)));
  take(W_WINO_U, 6L * 16384);
  w.total = off;
  return w;
}

#define CK(expr)                \
  do {                          \
    const int rc__ = (expr);    \
    if (rc__ != 0) return rc__; \
  } while (0)

struct Ctx {
  const DrqStep* s;
  ParamLayout P;
  WsLayout W;
  hipStream_t st;
  // phases 6 and 7 issued back to back by one call (single-GPU schedule): the actor loss then rides in the first
  // launch of the backward (drq_qout_bwd_actor) instead of a launch of its own.  A host that exchanges the metric
  // sums between the two phases calls them separately and keeps the separate loss launch.
  bool fuse_actor_loss = false;
  bool actor_loss_fused() const {
    return fuse_actor_loss && ((size_t)s->B + 5 * 1024 + 16) * 4 <= 60 * 1024;
  }
  // row-local stages fused with the first MLP layers (rowblock.hip): fp32, shapes its kernels take
  bool fuse_rows() const {
    // measured (tools/ab_flags.py, same process, interleaved): cheetah_run B=256 1021.9 us fused against 1024.2, but
    // humanoid_run (F+A = 121: odd, dword weight loads, 128-long padded reduction) 540 against 508 us at B=32 and 1220
    // against 1203 at B=256 -- the fused kernels are chains of dependent memory round trips like the launches they
    // replace, and only win where every weight row can be fetched with 8- or 16-byte loads into a 64-long reduction
    return !(s->flags & DRQ_STEP_NO_ROW_FUSION) && !s->bf16 && s->A <= 32 && s->F + s->A <= 64 && s->F % 2 == 0 &&
           s->A % 2 == 0 && s->H % 256 == 0;
  }
  bool use_gemm3() const { return !(s->flags & DRQ_STEP_NO_GEMM3) && !s->bf16; }
  // timing pair k of DrqStep.timing_events (bench.py's roofline), recorded when the host asked for that many
  int stamp(int idx) const {
    if (s->timing_events && idx < s->timing_n && hipEventRecord((hipEvent_t)s->timing_events[idx], st) != hipSuccess)
      return DRQ_EARG;
    return 0;
  }
  float* ws(int id) const { return s->ws + W.off[id]; }
  float* p(long off) const { return s->params + off; }
  float* g(long off) const { return s->grads + off; }
  int bf16() const { return s->bf16 ? 1 : 0; }
  // bf16 update: the outputs of conv1..conv3 (ACT1..3) are bf16 [frame][y][x][32] (conv_bf16.hip); the features (ACT4),
  // the gradients and the encoder input stay fp32
  bool acts16() const { return s->bf16 && !(s->flags & DRQ_STEP_BF16_FP32_ACTS); }
  // ... and so are the gradients that pass between the encoder's input-gradient launches (DY3, DY2: zero-padded by 2);
  // DY4 (written by the trunk's masked scatter) and DY1 (read by conv1's fp32 weight gradient) stay fp32
  bool grads16() const { return acts16() && !(s->flags & DRQ_STEP_BF16_FP32_GRADS); }
  float* gemm_ws() const { return ws(W_GEMM_WS); }
  size_t gemm_ws_bytes() const { return (size_t)16 * 1024 * 1024 * sizeof(float); }

  // n problems  y_i = act(x_i W_i^T + b_i)
  int fwd(int n, const float* const* x, long ldx, const float* const* w, const float* const* b, float* const* y,
          long ldy, int M, int N, int K, int relu) const {
    return drq_gemm_batched_any(bf16(), n, x, ldx, 1, w, K, 1, y, ldy, M, N, K, b, relu, nullptr, 0, nullptr, 0, 0, 0,
                                gemm_ws(), gemm_ws_bytes(), st);
  }
  // n problems  dx_i = (dy_i W_i) * (mask_i > 0);  W_i is [K][ldw] row-major, the first Nout columns are used
  int dgrad(int n, const float* const* dy, long lddy, const float* const* w, long ldw, float* const* dx, long lddx,
            int M, int Nout, int K, const float* const* mask, int ldmask) const {
    // two or more hidden x hidden problems: the LDS-DMA ring kernel (measured 15.8 against 19.2 us for the two heads of
    // the actor update; one problem of 2B rows: the LDS-free kernel is as fast)
    if (use_gemm3() && n >= 2 && K >= 256 && Nout >= 256) {
      const int rc = drq_gemm3_dgrad(n, dy, lddy, w, ldw, dx, lddx, M, Nout, K, mask, ldmask, st);
      if (rc != DRQ_EARG) return rc;
    }
    return drq_gemm_batched_any(bf16(), n, dy, lddy, 1, w, ldw, 0, dx, lddx, M, Nout, K, nullptr, 0, mask, ldmask, nullptr, 0,
                                0, 0, gemm_ws(), gemm_ws_bytes(), st);
  }
  // n problems  dW_i = dy_i^T x_i ([N][K] row-major), db_i = column sums of dy_i (fused)
  int wgrad(int n, const float* const* dy, long lddy, const float* const* x, long ldx, float* const* dw,
            float* const* db, int Brows, int N, int K) const {
    // the trunk's weight gradient (N = feature_dim rows, K = 39200 columns, reduction over the batch) at small
    // batches: the register-resident fp32 kernel beats the tiled bf16 GEMM (19 vs 31 us at B=256; 269 vs 153 at 2048)
    const int prec = (bf16() && N <= 128 && K >= 4096 && Brows < 512) ? 0 : bf16();
    return drq_gemm_batched_any(prec, n, dy, lddy, 0, x, ldx, 0, dw, K, N, K, Brows, nullptr, 0, nullptr, 0, db, 0, 0, 0,
                                gemm_ws(), gemm_ws_bytes(), st);
  }
  // The trunk's weight gradient dW = dz^T feat (+ bias gradient) with the LayerNorm parameter gradients of the same
  // trunk riding in the launch (one extra workgroup) when the dedicated kernel takes the shape; `ride` says whether
  // the LayerNorm backward left them out (ln_rides()).
  bool ln_rides() const { return !bf16() && s->F <= 128 && (s->B == 128 || s->B == 256 || s->B == 512); }
  int trunk_wgrad(const float* dz, const float* feat, float* gw, float* gb, bool ride, const float* dln,
                  const float* xhat, float* dgamma, float* dbeta) const {
    const int B = s->B, F = s->F;
    if (ride) {
      const int rc = drq_trunk_wgrad_ln(dz, F, feat, R, gw, R, F, (int)R, B, gb, dln, xhat, dgamma, dbeta, B, F, st);
      if (rc != DRQ_EARG) return rc;
      const int rc2 = drq_ln_param_grad(dln, xhat, dgamma, dbeta, B, F, st);
      if (rc2 != 0) return rc2;
    }
    const float *dzp[1] = {dz}, *xp[1] = {feat};
    float *gwp[1] = {gw}, *gbp[1] = {gb};
    return wgrad(1, dzp, F, xp, R, gwp, gbp, B, F, (int)R);
  }
  // both gradients of one layer (they read the same dy and are independent): one launch when the shape allows
  int wgrad_dgrad(int n, const float* const* dy, long lddy, const float* const* x, long ldx, float* const* dw,
                  float* const* db, const float* const* w, long ldw, float* const* dx, long lddx,
                  const float* const* mask, int ldmask, int Brows, int N, int K) const {
    if (use_gemm3() && n >= 2) {   // both heads of the critic: the LDS-DMA ring kernel (25.5 against 32.6 us)
      const int rc = drq_gemm3_wgrad_dgrad(n, dy, lddy, x, ldx, dw, db, w, ldw, dx, lddx, mask, ldmask, Brows, N, K, st);
      if (rc != DRQ_EARG) return rc;
    }
    if (!bf16()) {
      const int rc = drq_gemm2_wgrad_dgrad(n, dy, lddy, x, ldx, dw, db, w, ldw, dx, lddx, mask, ldmask, Brows, N, K, st);
      if (rc != DRQ_EARG) return rc;
    }
    const int rc = wgrad(n, dy, lddy, x, ldx, dw, db, Brows, N, K);
    if (rc != 0) return rc;
    return dgrad(n, dy, lddy, w, ldw, dx, lddx, Brows, K, N, mask, ldmask);
  }
};

// x == nullptr: a1 already holds the first layer's output (fused aug + conv1), start at layer 2
int encoder_forward(const Ctx& c, const float* x, int nb, float* a1, float* a2, float* a3, float* a4,
                    bool timed = false) {
  const ParamLayout& P = c.P;
  float* outs[4] = {a1, a2, a3, a4};
  const float* in = x ? x : a1;
  void* const* ev = (timed && c.s->timing_n >= 2) ? c.s->timing_events : nullptr;
  for (int l = x ? 0 : 1; l < 4; ++l) {
    const int hin = kEncH[l], hout = kEncH[l + 1];
    if (ev && l == 1 && hipEventRecord((hipEvent_t)ev[0], c.st) != hipSuccess) return DRQ_EARG;
    if (c.bf16() && l > 0)
      CK(drq_conv3x3_fwd_bf16_lay(in, c.p(P.enc_w[l]), c.p(P.enc_b[l]), outs[l], nb, hin, 1, 32L * hout * hout,
                                  (long)hout * hout, hout, 0, (x == nullptr && c.acts16()) ? (l < 3 ? 3 : 1) : 0, c.st));
    else if (l > 0)   // the 32->32 layers in Winograd F(2x2,3x3) form (conv_wino.hip); in the update (x == nullptr) the
                      // weight images come from the riders of the fused aug+conv1 launch
      CK(drq_conv3x3_fwd_wino_pre(in, c.p(P.enc_w[l]), x ? nullptr : c.ws(W_WINO_U) + (2L * (l - 1)) * 16384,
                                  c.p(P.enc_b[l]), outs[l], nb, hin, 1, 32L * hout * hout, (long)hout * hout, hout, 0,
                                  c.st));
    else
    CK(drq_conv3x3_fwd(in, c.p(P.enc_w[l]), c.p(P.enc_b[l]), outs[l], nb, l == 0 ? c.s->C : 32, hin, l == 0 ? 2 : 1,
                       1, 32L * hout * hout, (long)hout * hout, hout, 0, c.st));
    if (ev && l == 1 && hipEventRecord((hipEvent_t)ev[1], c.st) != hipSuccess) return DRQ_EARG;
    in = outs[l];
  }
  return 0;
}

// policy MLP (drqv2.py:77-81) on `rows` rows of h -> pre-tanh output p3
// sample_from >= 0: rows [sample_from, rows) also get their action sampled with `noise` into a_out (the output-layer
// kernel does both); needs A <= 64, else the output layer is a GEMM and the caller samples separately (returns 1)
// noise0 (with sample_from > 0): rows [0, sample_from) are sampled too, with their own noise, into (mu_out0, a_out0);
// returns 0, or 2 when that second job could not ride along (the caller then samples those rows itself)
int policy_forward(const Ctx& c, const float* h, int rows, float* p1, float* p2, float* p3, int sample_from = -1,
                   const float* noise = nullptr, float* a_out = nullptr, long lda_out = 0,
                   const float* noise0 = nullptr, float* mu_out0 = nullptr, float* a_out0 = nullptr, long lda_out0 = 0,
                   bool* did_rows0 = nullptr) {
  const DrqStep* s = c.s;
  const HeadOff& a = c.P.actor;
  const int H = s->H, F = s->F, A = s->A;
  const float *x0[1] = {h}, *x1[1] = {p1}, *x2[1] = {p2};
  const float *w0[1] = {c.p(a.w[0][0])}, *w1[1] = {c.p(a.w[0][1])}, *w2[1] = {c.p(a.w[0][2])};
  const float *b0[1] = {c.p(a.b[0][0])}, *b1[1] = {c.p(a.b[0][1])}, *b2[1] = {c.p(a.b[0][2])};
  float *y0[1] = {p1}, *y1[1] = {p2}, *y2[1] = {p3};
  CK(c.fwd(1, x0, F, w0, b0, y0, H, rows, H, F, 1));
  CK(c.fwd(1, x1, H, w1, b1, y1, H, rows, H, H, 1));
  if (A <= 64) {
    CK(drq_policy_out_fwd(p2, w2[0], b2[0], p3, rows, H, A, sample_from >= 0 ? noise : nullptr, s->std, s->clip, 1,
                          sample_from >= 0 ? sample_from : 0, nullptr, a_out, lda_out,
                          sample_from > 0 ? noise0 : nullptr, mu_out0, a_out0, lda_out0, c.st));
    if (did_rows0) *did_rows0 = sample_from > 0 && noise0 != nullptr;
    return 0;
  }
  CK(c.fwd(1, x2, H, w2, b2, y2, A, rows, A, H, 0));
  if (sample_from >= 0)
    CK(drq_trunc_normal_sample(p3 + (long)sample_from * A, noise, s->std, s->clip, 1, nullptr, a_out, lda_out,
                               rows - sample_from, A, c.st));
  return 0;
}

// Twin-Q forward for `nn` (net, input) pairs at once: 2*nn problems per layer (drqv2.py:103-111,117-119)
// l1_done: the first layers' outputs are already in h1 (fused with the row-local stage that produced their input)
int q_forward(const Ctx& c, int nn, const HeadOff* const* nets, const float* const* ha, float* const* h1,
              float* const* h2, float* const* q, bool l1_done = false) {
  const DrqStep* s = c.s;
  const int B = s->B, H = s->H, FA = s->F + s->A;
  const long BH = (long)B * H;
  const float *x[8], *w[8], *b[8], *hh1[8], *hh2[8];
  float *y1[8], *y2[8], *qq[8];
  for (int l = 0; l < 3; ++l) {
    for (int i = 0; i < nn; ++i)
      for (int h = 0; h < 2; ++h) {
        const int z = 2 * i + h;
        w[z] = c.p(nets[i]->w[h][l]);
        b[z] = c.p(nets[i]->b[h][l]);
        x[z] = ha[i];
        y1[z] = h1[i] + h * BH; hh1[z] = y1[z];
        y2[z] = h2[i] + h * BH; hh2[z] = y2[z];
        qq[z] = q[i] + h * B;
      }
    if (l == 0 && !l1_done) CK(c.fwd(2 * nn, x, FA, w, b, y1, H, B, H, FA, 1));
    if (l == 1) CK(c.fwd(2 * nn, hh1, H, w, b, y2, H, B, H, H, 1));
    if (l == 2) CK(drq_qout_fwd(2 * nn, hh2, w, b, qq, B, H, c.st));
  }
  return 0;
}

// ---- phase 0 = phases 3, 4, 5 -------------------------------------------------------------------
// phase 3: augmentation + encoder forward (reads encoder weights only)
int phase_encode(const Ctx& c) {
  const DrqStep* s = c.s;
  const int B = s->B, C = s->C;
  hipStream_t st = c.st;
  float* aug = c.ws(W_AUG);
  // aug (drqv2.py:241-242) + /255-0.5 (:64) + conv1 (:55) in one kernel that reads the uint8 frames once; rows
  // [0,B) = obs, [B,2B) = next_obs.  Only the obs view's encoder input is kept (conv1's weight gradient reads it).
  (void)C;
  // riders of the same launch: the Winograd images of the conv2..4 weights for this update's forward and backward
  const float* wino_w[3] = {c.p(c.P.enc_w[1]), c.p(c.P.enc_w[2]), c.p(c.P.enc_w[3])};
  CK(drq_conv1_aug_fwd_any(c.acts16() ? 2 : c.bf16(), s->obs, s->shift_obs, s->next_obs, s->shift_next, s->base_grid,
                           c.p(c.P.enc_w[0]), c.p(c.P.enc_b[0]), aug, c.ws(W_ACT1), B, s->store_aug_next ? 2 * B : B, st,
                           c.bf16() ? nullptr : wino_w, c.bf16() ? nullptr : c.ws(W_WINO_U), s->obs_index,
                           s->next_obs_index));
  // layers 2..4 on both views in one pass (:244-246)
  CK(encoder_forward(c, nullptr, 2 * B, c.ws(W_ACT1), c.ws(W_ACT2), c.ws(W_ACT3), c.ws(W_FEAT), true));
  return 0;
}

// phase 4: trunks, policy, Q heads, TD loss and its backward down to the encoder output
// (leaves ALL critic gradients and sums[0..4]; reads actor, critic and target weights)
int phase_critic_heads(const Ctx& c) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, A = s->A, F = s->F, H = s->H, FA = F + A;
  hipStream_t st = c.st;
  float* feat = c.ws(W_FEAT);
  float* feat_obs = feat;
  float* feat_next = feat + (long)B * R;
  const HeadOff &cr = P.critic, &ac = P.actor, &tg = P.target;
  const long BH = (long)B * H;

  // all four trunks in one launch: critic(obs), actor(obs), actor(next), target(next).  actor(obs) is needed
  // only by the actor step; the actor's weights do not change before that, so it is evaluated here.
  float* z4 = c.ws(W_Z4);
  float* hrows = c.ws(W_HROWS);        // actor trunk outputs: rows [0,B) obs, [B,2B) next
  bool rows_fused = false;
  {
    const float* x[4] = {feat_obs, feat_obs, feat_next, feat_next};
    const float* w[4] = {c.p(cr.trunk_w), c.p(ac.trunk_w), c.p(ac.trunk_w), c.p(tg.trunk_w)};
    const float* b[4] = {c.p(cr.trunk_b), c.p(ac.trunk_b), c.p(ac.trunk_b), c.p(tg.trunk_b)};
    float* y[4] = {z4, z4 + (long)B * F, z4 + 2L * B * F, z4 + 3L * B * F};
    // the GEMM leaves its split-K partials in the workspace: the LayerNorm kernel sums them (+ bias) itself
    int sk = 1;
    CK(drq_gemm_batched_partial_any(c.bf16(), 4, x, R, 1, w, R, 1, y, F, B, F, (int)R, b, c.gemm_ws(), c.gemm_ws_bytes(), &sk, st));
    const float* zz[4] = {y[0], y[1], y[2], y[3]};
    const float* gm[4] = {c.p(cr.ln_g), c.p(ac.ln_g), c.p(ac.ln_g), c.p(tg.ln_g)};
    const float* bt[4] = {c.p(cr.ln_b), c.p(ac.ln_b), c.p(ac.ln_b), c.p(tg.ln_b)};
    float* out[4] = {c.ws(W_HA_C), hrows, hrows + (long)B * F, c.ws(W_HA_T)};
    const int ldo[4] = {FA, F, F, FA};
    float* xh[4] = {c.ws(W_XHAT_C), c.ws(W_XHAT_A), nullptr, nullptr};
    float* rs[4] = {c.ws(W_RSTD_C), c.ws(W_RSTD_A), nullptr, nullptr};
    // the critic's [h, action] input: the action columns ride along with problem 0 (drqv2.py:106)
    const float* tail[4] = {s->action, nullptr, nullptr, nullptr};
    const int tld[4] = {A, 0, 0, 0};
    const bool with_tail = A <= 64;
    if (c.fuse_rows() && sk > 1) {
      // ... and the first layers ride along with the LayerNorm: both Q heads of the critic on [h, action], the policy's
      // first layer on the actor's 2B rows (rowblock.hip; one launch instead of three)
      const float* part[4];
      for (int g = 0; g < 4; ++g) part[g] = c.gemm_ws() + (long)g * sk * B * F;
      const int tn[4] = {A, 0, 0, 0};
      const int rows[4] = {B, B, B, B};
      const int nheads[4] = {2, 1, 1, 0};
      const float* w1[8] = {c.p(cr.w[0][0]), c.p(cr.w[1][0]), c.p(ac.w[0][0]), nullptr, c.p(ac.w[0][0]), nullptr, nullptr, nullptr};
      const float* b1[8] = {c.p(cr.b[0][0]), c.p(cr.b[1][0]), c.p(ac.b[0][0]), nullptr, c.p(ac.b[0][0]), nullptr, nullptr, nullptr};
      float* y1[8] = {c.ws(W_C1), c.ws(W_C1) + BH, c.ws(W_P1), nullptr, c.ws(W_P1) + BH, nullptr, nullptr, nullptr};
      const int rc = drq_lnl1_fwd(4, part, nullptr, b, gm, bt, out, ldo, xh, rs, tail, tld, tn, rows, nheads, w1, b1, y1, F, H, sk,
                                  (long)B * F, st);
      if (rc == DRQ_OK) rows_fused = true;
      else if (rc != DRQ_EARG) return rc;
    }
    if (!rows_fused) {
    CK(drq_ln_tanh_fwd_multi_part(4, zz, F, gm, bt, out, ldo, xh, rs, B, F, with_tail ? tail : nullptr,
                                  with_tail ? tld : nullptr, with_tail ? A : 0, sk > 1 ? c.gemm_ws() : nullptr, b, sk,
                                  st));
    if (!with_tail) CK(drq_copy_cols(s->action, A, c.ws(W_HA_C) + F, FA, B, A, st));
    }
  }
  // policy MLP once on the 2B stacked rows
  // ... and, in the output-layer kernel, a' ~ TruncN(actor(next)) for the target (:180-183)
  // ... and the actor update's own draw for the obs rows (:210-211; same policy output, the actor's weights do
  // not change in between): mu and the action columns of the second critic input are ready for phase 6
  bool q_l1_done = false;
  if (rows_fused) {
    const float *x1[1] = {c.ws(W_P1)}, *w1[1] = {c.p(ac.w[0][1])}, *b1[1] = {c.p(ac.b[0][1])};
    float* y1[1] = {c.ws(W_P2)};
    CK(c.fwd(1, x1, H, w1, b1, y1, H, 2 * B, H, H, 1));
    // output layer + both samples + the target critic's first layers on [h_target, a'] in one launch
    const float* wt[2] = {c.p(tg.w[0][0]), c.p(tg.w[1][0])};
    const float* bt2[2] = {c.p(tg.b[0][0]), c.p(tg.b[1][0])};
    float* yt[2] = {c.ws(W_T1), c.ws(W_T1) + BH};
    CK(drq_polout_l1_fwd(c.ws(W_P2), c.p(ac.w[0][2]), c.p(ac.b[0][2]), c.ws(W_P3), 2 * B, B, H, A, F, s->std, s->clip, 1,
                         s->noise_critic, nullptr, c.ws(W_HA_T), FA, s->noise_actor, c.ws(W_MU_O), c.ws(W_HA_C2), FA, 2, wt,
                         bt2, yt, st));
    q_l1_done = true;
  } else {
  bool did0 = false;
  CK(policy_forward(c, hrows, 2 * B, c.ws(W_P1), c.ws(W_P2), c.ws(W_P3), B, s->noise_critic, c.ws(W_HA_T) + F, FA,
                    s->noise_actor, c.ws(W_MU_O), c.ws(W_HA_C2) + F, FA, &did0));
  if (!did0)
    CK(drq_trunc_normal_sample(c.ws(W_P3), s->noise_actor, s->std, s->clip, 1, c.ws(W_MU_O), c.ws(W_HA_C2) + F, FA, B,
                               A, st));
  }

  // y = r + d*min Q_target(next, a')   (:184-186);  critic(obs, action) (:188)
  {
    const HeadOff* nets[2] = {&tg, &cr};
    const float* ha[2] = {c.ws(W_HA_T), c.ws(W_HA_C)};
    float* h1[2] = {c.ws(W_T1), c.ws(W_C1)};
    float* h2[2] = {c.ws(W_T2), c.ws(W_C2)};
    float* q[2] = {c.ws(W_TQ), c.ws(W_Q)};
    CK(q_forward(c, 2, nets, ha, h1, h2, q, q_l1_done));
  }
  const float invB = 1.0f / (float)s->global_B;

  // ---- backward of the critic loss (:200), both heads per launch
  int sk_dha = 1;
  {
    float *c1[2] = {c.ws(W_C1), c.ws(W_C1) + BH}, *c2[2] = {c.ws(W_C2), c.ws(W_C2) + BH};
    float *dc1[2] = {c.ws(W_DC1), c.ws(W_DC1) + BH}, *dc2[2] = {c.ws(W_DC2), c.ws(W_DC2) + BH};
    float* dha[2] = {c.ws(W_DHA), c.ws(W_DHA) + (long)B * FA};
    const float *dq[2] = {c.ws(W_DQ), c.ws(W_DQ) + B};
    const float *c1c[2] = {c1[0], c1[1]}, *c2c[2] = {c2[0], c2[1]}, *dc1c[2] = {dc1[0], dc1[1]},
                *dc2c[2] = {dc2[0], dc2[1]}, *hac[2] = {c.ws(W_HA_C), c.ws(W_HA_C)};
    const float *w0[2] = {c.p(cr.w[0][0]), c.p(cr.w[1][0])}, *w1[2] = {c.p(cr.w[0][1]), c.p(cr.w[1][1])},
                *w2[2] = {c.p(cr.w[0][2]), c.p(cr.w[1][2])};
    float *gw0[2] = {c.g(cr.w[0][0]), c.g(cr.w[1][0])}, *gw1[2] = {c.g(cr.w[0][1]), c.g(cr.w[1][1])},
          *gw2[2] = {c.g(cr.w[0][2]), c.g(cr.w[1][2])};
    float *gb0[2] = {c.g(cr.b[0][0]), c.g(cr.b[1][0])}, *gb1[2] = {c.g(cr.b[0][1]), c.g(cr.b[1][1])},
          *gb2[2] = {c.g(cr.b[0][2]), c.g(cr.b[1][2])};
    // TD target + twin MSE (:185-189) and layer 3 (hidden -> 1) backward in one pass: dq never leaves the chip,
    // sums[0..4] come from the same launch
    if (((size_t)B + 5 * 1024 + 16) * 4 <= 60 * 1024) {
      CK(drq_qout_bwd_td(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_Q), c.ws(W_Q) + B, s->reward, s->discount, invB, s->sums,
                         c2c, w2, dc2, gw2, gb2, B, H, st));
    } else {
      CK(drq_td_mse(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_Q), c.ws(W_Q) + B, s->reward, s->discount, c.ws(W_DQ),
                    c.ws(W_DQ) + B, s->sums, B, invB, st));
      CK(drq_qout_bwd(2, dq, c2c, w2, dc2, gw2, gb2, B, H, st));
    }
    // layer 2
    CK(c.wgrad_dgrad(2, dc2c, H, c1c, H, gw1, gb1, w1, H, dc1, H, c1c, H, B, H, H));
    // layer 1 (input = [h, action], shared by both heads)
    CK(c.wgrad(2, dc1c, H, hac, FA, gw0, gb0, B, H, FA));
    // the split-K partials of this dgrad stay in the workspace: the LayerNorm backward sums them (both heads)
    CK(drq_gemm_batched_partial_any(c.bf16(), 2, dc1c, H, 1, w0, FA, 0, dha, FA, B, FA, H, nullptr, c.gemm_ws(), c.gemm_ws_bytes(),
                                &sk_dha, st));
  }
  // trunk: LayerNorm+tanh backward, then Linear(R -> F)
  const bool ride = c.ln_rides();
  CK(drq_ln_tanh_bwd_part(c.ws(W_DHA), FA, c.ws(W_DHA) + (long)B * FA, FA, c.ws(W_HA_C), FA, c.ws(W_XHAT_C),
                          c.ws(W_RSTD_C), c.p(cr.ln_g), c.ws(W_DZ_C), c.ws(W_DLN), ride ? nullptr : c.g(cr.ln_g),
                          ride ? nullptr : c.g(cr.ln_b), B, F, sk_dha > 1 ? c.gemm_ws() : nullptr, sk_dha, 2, FA, st));
  {
    const float *dz[1] = {c.ws(W_DZ_C)}, *w[1] = {c.p(cr.trunk_w)}, *mk[1] = {feat_obs};
    float* dy4[1] = {c.ws(W_DY4)};
    CK(c.trunk_wgrad(c.ws(W_DZ_C), feat_obs, c.g(cr.trunk_w), c.g(cr.trunk_b), ride, c.ws(W_DLN), c.ws(W_XHAT_C),
                     c.g(cr.ln_g), c.g(cr.ln_b)));
    // d feat = dz W_t, masked by relu(conv4) and scattered into the padded conv-gradient layout
    // (fp32 in both precisions: this product is bound by its 8 bytes per output element, not by arithmetic, and the
    // dedicated kernel moves them faster than the tiled bf16 GEMM: 29 vs 49 us at B=256, 317 vs 371 us at B=2048)
    CK(drq_gemm_batched_any(0, 1, dz, F, 1, w, R, 0, dy4, 0, B, (int)R, F, nullptr, 0, mk, (int)R, nullptr, 35, 0, 0,
                            c.gemm_ws(), c.gemm_ws_bytes(), st));
  }

  return 0;
}

// phase 5: encoder backward, conv4 .. conv1 (wgrad all, dgrad 4..2): leaves the encoder gradients
int phase_conv_backward(const Ctx& c) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, C = s->C;
  hipStream_t st = c.st;
  const int dyid[4] = {W_DY1, W_DY2, W_DY3, W_DY4};
  const int actid[4] = {W_AUG, W_ACT1, W_ACT2, W_ACT3};   // layer inputs
  // partial sums of the four weight gradients go to four quarters of the conv workspace; ONE reduction launch
  // at the end turns them into dW/db (nothing reads those before the optimiser step)
  float* cws = c.ws(W_CONV_WS);
  const size_t quarter = drq_conv3x3_wgrad_ws_bytes() / 4;
  const float* parts[4];
  int nblk[4], cins[4];
  float *dws[4], *dbs[4];
  // fp32: the input gradients first (a chain: dY4 -> dY3 -> dY2 -> dY1), then the weight gradients of conv2..4 in ONE
  // Winograd launch (each needs only its layer's dY and input) and conv1's; bf16: layer by layer as before
  const bool merged = !c.bf16();
  for (int l = 3; l >= 0; --l) {
    const int hin = kEncH[l], hout = kEncH[l + 1], hp = hout + 4;
    const float* dy = c.ws(dyid[l]);
    float* part = cws + (size_t)l * (quarter / sizeof(float));
    parts[l] = part; cins[l] = l == 0 ? C : 32; dws[l] = c.g(P.enc_w[l]); dbs[l] = c.g(P.enc_b[l]);
    if (!merged) {
      if (l > 0)
        CK(drq_conv3x3_wgrad_partial_bf16_lay(c.ws(actid[l]), dy, B, hin, 32L * hp * hp, (long)hp * hp, hp, 2L * hp + 2,
                                              part, quarter, &nblk[l],
                                              (c.acts16() ? 1 : 0) | (c.grads16() && l < 3 ? 2 : 0), st));
      else
        CK(drq_conv3x3_wgrad_partial(c.ws(actid[l]), dy, B, C, hin, 2, 32L * hp * hp, (long)hp * hp, hp, 2L * hp + 2,
                                     part, quarter, &nblk[l], st));
    }
    if (l >= 1) {
      const int hpi = hin + 4;   // padded size of the next (shallower) gradient buffer
      void* const* ev = s->timing_n >= 4 ? s->timing_events : nullptr;
      if (ev && l == 2 && hipEventRecord((hipEvent_t)ev[2], st) != hipSuccess) return DRQ_EARG;
      if (c.bf16())
        CK(drq_conv3x3_dgrad_bf16_lay(dy, c.p(P.enc_w[l]), c.ws(actid[l]), c.ws(dyid[l - 1]), B, hout, 32L * hpi * hpi,
                                      (long)hpi * hpi, hpi, 2L * hpi + 2,
                                      (c.acts16() ? 4 : 0) | (c.grads16() && l < 3 ? 1 : 0) | (c.grads16() && l > 1 ? 2 : 0),
                                      st));
      else
      CK(drq_conv3x3_dgrad_wino_pre(dy, c.p(P.enc_w[l]), c.ws(W_WINO_U) + (2L * (l - 1) + 1) * 16384, c.ws(actid[l]),
                                    c.ws(dyid[l - 1]), B, hout, 32L * hpi * hpi, (long)hpi * hpi, hpi, 2L * hpi + 2, st));
      if (ev && l == 2 && hipEventRecord((hipEvent_t)ev[3], st) != hipSuccess) return DRQ_EARG;
    }
  }
  if (merged) {
    const float *xs[3], *dys[3];
    float* ps[3];
    long bs[3], cs[3], rs[3], off[3];
    for (int k = 0; k < 3; ++k) {            // k = 0, 1, 2 -> layers 1, 2, 3 (conv2, conv3, conv4: hin 41, 39, 37)
      const int l = k + 1, hp = kEncH[l + 1] + 4;
      xs[k] = c.ws(actid[l]); dys[k] = c.ws(dyid[l]); ps[k] = const_cast<float*>(parts[l]);
      bs[k] = 32L * hp * hp; cs[k] = (long)hp * hp; rs[k] = hp; off[k] = 2L * hp + 2;
    }
    int nb3[3];
    CK(c.stamp(4));
    int rc = drq_conv3x3_wgrad_partial_wino3(xs, dys, B, bs, cs, rs, off, ps, quarter, nb3, st);
    if (rc == DRQ_EARG) {                    // tiny batch: one launch per layer
      for (int k = 0; k < 3; ++k)
        CK(drq_conv3x3_wgrad_partial_wino(xs[k], dys[k], B, kEncH[k + 1], bs[k], cs[k], rs[k], off[k], ps[k], quarter,
                                          &nb3[k], st));
    } else if (rc != 0) {
      return rc;
    }
    CK(c.stamp(5));
    for (int k = 0; k < 3; ++k) nblk[k + 1] = nb3[k];
    const int hp0 = kEncH[1] + 4;
    CK(drq_conv3x3_wgrad_partial(c.ws(actid[0]), c.ws(dyid[0]), B, C, kEncH[0], 2, 32L * hp0 * hp0, (long)hp0 * hp0, hp0,
                                 2L * hp0 + 2, const_cast<float*>(parts[0]), quarter, &nblk[0], st));
  }
  CK(drq_conv3x3_wgrad_reduce_multi(4, parts, nblk, cins, dws, dbs, st));
  return 0;
}

// ---- phase 1 ---------------------------------------------------------------------------------
// metrics mirror: sums -> pinned host memory, then the sequence word (system-scope release)
__global__ void publish_sums_kernel(const float* sums, float* host, unsigned seq) {
  if (threadIdx.x < 8) host[threadIdx.x] = sums[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store((unsigned*)(host + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int publish_sums(const float* sums, float* sums_host, unsigned seq, hipStream_t st) {
  if (!sums || !sums_host) return DRQ_EARG;
  hipLaunchKernelGGL(publish_sums_kernel, dim3(1), dim3(64), 0, st, sums, sums_host, seq);
  DRQ_LAUNCH_CHECK();
  return DRQ_OK;
}

// ---- phase 1 = phases 6, 7 ------------------------------------------------------------------------
// phase 6: critic_opt.step() (:201) + Polyak (:259-260, fused: it reads the stepped critic; nothing between here
// and the end of the update reads the target), then the actor loss through the UPDATED critic (:210-216).
// encoder_opt.step() (:202) is phase 8: the actor update works on obs.detach() encoded BEFORE that step
// (:255), so nothing in phases 6/7 reads the encoder weights and the step commutes to the end of the update.
// with_opt = false: the forward only (the critic's optimiser step was issued on its own: phase 10)
int phase_actor_forward(const Ctx& c, bool with_opt = true) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, A = s->A, F = s->F, FA = F + A;
  hipStream_t st = c.st;
  float* feat_obs = c.ws(W_FEAT);
  const HeadOff& cr = P.critic;

  if (with_opt)
  CK(drq_adam_flat(c.p(P.seg[2]), c.g(P.seg[2]), s->adam_m + P.seg[2], s->adam_v + P.seg[2], P.seg[3] - P.seg[2],
                   s->lr, s->step_critic, s->gscale, c.p(P.seg[6]), s->tau, st));
  bool q_l1_done = false;
  CK(c.stamp(8));

  // a ~ TruncN(actor(obs.detach())) (:210-211) was drawn in phase 4 together with the policy output
  // updated critic on (obs, a) (:213)
  {
    const float *x[1] = {feat_obs}, *w[1] = {c.p(cr.trunk_w)}, *b[1] = {c.p(cr.trunk_b)};
    float* y[1] = {c.ws(W_Z_C2)};
    int sk = 1;
    CK(drq_gemm_batched_partial_any(c.bf16(), 1, x, R, 1, w, R, 1, y, F, B, F, (int)R, b, c.gemm_ws(), c.gemm_ws_bytes(), &sk, st));
    const float *zz[1] = {y[0]}, *gm[1] = {c.p(cr.ln_g)}, *bt[1] = {c.p(cr.ln_b)};
    float* out[1] = {c.ws(W_HA_C2)};
    const int ldo[1] = {FA};
    if (c.fuse_rows() && sk > 1) {
      // LayerNorm + both first layers of the (updated) critic on [h, a]: a already sits in the action columns of the
      // row (phase 4 drew it), the "tail" is those columns themselves
      const float* part[1] = {c.gemm_ws()};
      const float* tail[1] = {c.ws(W_HA_C2) + F};
      const int tld[1] = {FA}, tn[1] = {A}, rows[1] = {B}, nheads[1] = {2};
      const float* w1[2] = {c.p(cr.w[0][0]), c.p(cr.w[1][0])};
      const float* b1[2] = {c.p(cr.b[0][0]), c.p(cr.b[1][0])};
      float* y1[2] = {c.ws(W_T1), c.ws(W_T1) + (long)B * s->H};
      const int rc = drq_lnl1_fwd(1, part, nullptr, b, gm, bt, out, ldo, nullptr, nullptr, tail, tld, tn, rows, nheads, w1, b1,
                                  y1, F, s->H, sk, (long)B * F, st);
      if (rc == DRQ_OK) q_l1_done = true;
      else if (rc != DRQ_EARG) return rc;
    }
    if (!q_l1_done)
    CK(drq_ln_tanh_fwd_multi_part(1, zz, F, gm, bt, out, ldo, nullptr, nullptr, B, F, nullptr, nullptr, 0,
                                  sk > 1 ? c.gemm_ws() : nullptr, b, sk, st));
  }
  {
    const HeadOff* nets[1] = {&cr};
    const float* ha[1] = {c.ws(W_HA_C2)};
    float *h1[1] = {c.ws(W_T1)}, *h2[1] = {c.ws(W_T2)}, *q[1] = {c.ws(W_TQ)};
    CK(q_forward(c, 1, nets, ha, h1, h2, q, q_l1_done));
  }
  const float invB = 1.0f / (float)s->global_B;
  // the loss kernel also publishes the eight sums to the host mirror when one is given
  if (!c.actor_loss_fused())
    CK(drq_actor_loss_ex(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_HA_C2) + F, FA, c.ws(W_MU_O), s->std, c.ws(W_DQ),
                         c.ws(W_DQ) + B, s->sums, B, A, invB, s->sums_host, (unsigned)s->step_actor, st));
  return 0;
}

// phase 7: backward of the actor loss (:218-221) -> actor gradients
int phase_actor_backward(const Ctx& c) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, A = s->A, F = s->F, H = s->H, FA = F + A;
  hipStream_t st = c.st;
  float* feat_obs = c.ws(W_FEAT);
  const HeadOff &cr = P.critic, &ac = P.actor;
  const long BH = (long)B * H;

  // output layer backward in one kernel (dpre, dW3, db3, dp2) when dpre fits its LDS
  const bool fused_head = A <= 32 && ((size_t)B * A + 4 * 1024) * 4 <= 60 * 1024;
  int sk_da = 1;
  // backward through the critic to the action only (critic weight grads are never used: SURVEY A7(iii))
  {
    const float *dq[2] = {c.ws(W_DQ), c.ws(W_DQ) + B};
    const float *t1[2] = {c.ws(W_T1), c.ws(W_T1) + BH}, *t2[2] = {c.ws(W_T2), c.ws(W_T2) + BH};
    float *dc1[2] = {c.ws(W_DC1), c.ws(W_DC1) + BH}, *dc2[2] = {c.ws(W_DC2), c.ws(W_DC2) + BH};
    const float *dc1c[2] = {dc1[0], dc1[1]}, *dc2c[2] = {dc2[0], dc2[1]};
    const float *w0a[2] = {c.p(cr.w[0][0]) + F, c.p(cr.w[1][0]) + F}, *w1[2] = {c.p(cr.w[0][1]), c.p(cr.w[1][1])},
                *w2[2] = {c.p(cr.w[0][2]), c.p(cr.w[1][2])};
    float* da[2] = {c.ws(W_DA), c.ws(W_DA) + (long)B * A};
    if (c.actor_loss_fused())
      CK(drq_qout_bwd_actor(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_HA_C2) + F, FA, c.ws(W_MU_O), s->std, A,
                            1.0f / (float)s->global_B, s->sums, s->sums_host, (unsigned)s->step_actor, t2, w2, dc2, B, H,
                            st));
    else
      CK(drq_qout_bwd(2, dq, t2, w2, dc2, nullptr, nullptr, B, H, st));
    CK(c.dgrad(2, dc2c, H, w1, H, dc1, H, B, H, H, t1, H));
    // action columns of layer 1 only; with the fused output-layer backward the split-K partials stay in the
    // workspace and that kernel sums them
    if (fused_head)
      CK(drq_gemm_batched_partial_any(c.bf16(), 2, dc1c, H, 1, w0a, FA, 0, da, A, B, A, H, nullptr, c.gemm_ws(), c.gemm_ws_bytes(),
                                  &sk_da, st));
    else
      CK(c.dgrad(2, dc1c, H, w0a, FA, da, A, B, A, H, nullptr, 0));
  }

  if (!fused_head) CK(drq_actor_dmu(c.ws(W_DA), c.ws(W_DA) + (long)B * A, A, 0, c.ws(W_MU_O), c.ws(W_DPRE), B, A, st));

  // policy MLP backward (rows [0,B) of the stacked activations are the obs rows)
  int sk_dh = 1;
  {
    const float *dpre[1] = {c.ws(W_DPRE)}, *p1[1] = {c.ws(W_P1)}, *p2[1] = {c.ws(W_P2)}, *h[1] = {c.ws(W_HROWS)};
    float *dp1[1] = {c.ws(W_DP1)}, *dp2[1] = {c.ws(W_DP2)}, *dh[1] = {c.ws(W_DH_A)};
    const float *dp1c[1] = {dp1[0]}, *dp2c[1] = {dp2[0]};
    const float *w0[1] = {c.p(ac.w[0][0])}, *w1[1] = {c.p(ac.w[0][1])}, *w2[1] = {c.p(ac.w[0][2])};
    float *gw0[1] = {c.g(ac.w[0][0])}, *gw1[1] = {c.g(ac.w[0][1])}, *gw2[1] = {c.g(ac.w[0][2])};
    float *gb0[1] = {c.g(ac.b[0][0])}, *gb1[1] = {c.g(ac.b[0][1])}, *gb2[1] = {c.g(ac.b[0][2])};
    if (fused_head) {
      CK(drq_policy_out_bwd(c.ws(W_DA), c.ws(W_DA) + (long)B * A, A, 0, c.ws(W_MU_O), p2[0], w2[0], dp2[0], gw2[0],
                            gb2[0], B, H, A, sk_da > 1 ? c.gemm_ws() : nullptr, sk_da, st));
    } else {
      CK(c.wgrad(1, dpre, A, p2, H, gw2, gb2, B, A, H));
      CK(c.dgrad(1, dpre, A, w2, H, dp2, H, B, H, A, p2, H));
    }
    CK(c.wgrad_dgrad(1, dp2c, H, p1, H, gw1, gb1, w1, H, dp1, H, p1, H, B, H, H));
    CK(c.wgrad(1, dp1c, H, h, F, gw0, gb0, B, H, F));
    CK(drq_gemm_batched_partial_any(c.bf16(), 1, dp1c, H, 1, w0, F, 0, dh, F, B, F, H, nullptr, c.gemm_ws(), c.gemm_ws_bytes(), &sk_dh,
                                st));
  }
  const bool ride = c.ln_rides();
  CK(drq_ln_tanh_bwd_part(c.ws(W_DH_A), F, nullptr, 0, c.ws(W_HROWS), F, c.ws(W_XHAT_A), c.ws(W_RSTD_A), c.p(ac.ln_g),
                          c.ws(W_DZ_A), c.ws(W_DLN), ride ? nullptr : c.g(ac.ln_g), ride ? nullptr : c.g(ac.ln_b), B, F,
                          sk_dh > 1 ? c.gemm_ws() : nullptr, sk_dh, 1, F, st));
  CK(c.trunk_wgrad(c.ws(W_DZ_A), feat_obs, c.g(ac.trunk_w), c.g(ac.trunk_b), ride, c.ws(W_DLN), c.ws(W_XHAT_A),
                   c.g(ac.ln_g), c.g(ac.ln_b)));
  return 0;
}

// ---- phase 2 = phases 8, 9 ------------------------------------------------------------------------
int phase_encoder_opt(const Ctx& c) {   // encoder_opt.step() (:202)
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  return drq_adam_flat(c.p(P.seg[0]), c.g(P.seg[0]), s->adam_m + P.seg[0], s->adam_v + P.seg[0], P.seg[1] - P.seg[0],
                       s->lr, s->step_enc, s->gscale, nullptr, 0.0, c.st);
}

int phase_actor_opt(const Ctx& c) {     // actor_opt.step() (:221)
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  return drq_adam_flat(c.p(P.seg[4]), c.g(P.seg[4]), s->adam_m + P.seg[4], s->adam_v + P.seg[4], P.seg[5] - P.seg[4],
                       s->lr, s->step_actor, s->gscale, nullptr, 0.0, c.st);
}

int check_step(const DrqStep* s) {
  if (!s) return DRQ_EARG;
  if (s->B <= 0 || s->global_B < s->B || s->C <= 0 || s->C > 32 || s->A <= 0 || s->F <= 0 || s->F > 256 || s->H <= 0)
    return DRQ_EARG;
  if (s->C != 9) return DRQ_EARG;   // conv1 kernel is instantiated for frame_stack=3 (cfgs/config.yaml:7)
  if (!s->params || !s->ws) return DRQ_EARG;
  if ((s->obs_index != nullptr) != (s->next_obs_index != nullptr)) return DRQ_EARG;
  if (s->ws_bytes < drq_step_ws_bytes(s->B, s->C, s->A, s->F, s->H)) return DRQ_EWS;
  return 0;
}

}  // namespace

extern "C" {

DRQ_API int drq_abi_version(void) { return 7; }

DRQ_API int drq_param_layout(int C, int A, int F, int H, long* out, int cap) {
  if (!out || cap < DRQ_PARAM_LAYOUT_LEN || C <= 0 || A <= 0 || F <= 0 || H <= 0) return DRQ_EARG;
  const ParamLayout L = param_layout(C, A, F, H);
  int n = 0;
  for (int i = 0; i < L.nflat; ++i) out[n++] = L.flat[i];
  for (int i = 0; i < 8; ++i) out[n++] = L.seg[i];
  out[n++] = L.total;
  return n;
}

DRQ_API size_t drq_step_ws_bytes(int B, int C, int A, int F, int H) {
  if (B <= 0 || C <= 0 || A <= 0 || F <= 0 || H <= 0) return 0;
  return (size_t)ws_layout(B, C, A, F, H).total * sizeof(float);
}

DRQ_API long drq_step_ws_offset(int B, int C, int A, int F, int H, int id) {
  if (id < 0 || id >= DRQ_WS_NBUF_PUBLIC) return -1;
  return ws_layout(B, C, A, F, H).off[id];
}

DRQ_API int drq_update_phase(const DrqStep* s, int phase) {
  CK(check_step(s));
  if (!s->obs || !s->next_obs || !s->action || !s->reward || !s->discount || !s->shift_obs || !s->shift_next ||
      !s->noise_critic || !s->noise_actor || !s->base_grid || !s->grads || !s->adam_m || !s->adam_v || !s->sums)
    return DRQ_EARG;
  Ctx c{s, param_layout(s->C, s->A, s->F, s->H), ws_layout(s->B, s->C, s->A, s->F, s->H), (hipStream_t)s->stream};
  if (phase < -1 || phase > 13) return DRQ_EARG;
  c.fuse_actor_loss = phase == -1 || phase == 1;
  if (phase == 3 || phase == 0 || phase == -1) CK(phase_encode(c));
  if (phase == 4 || phase == 0 || phase == -1) {
    CK(c.stamp(6));
    CK(phase_critic_heads(c));
    CK(c.stamp(7));
  }
  if (phase == 5 || phase == 0 || phase == -1) CK(phase_conv_backward(c));
  if (phase == 6 || phase == 1 || phase == -1) CK(phase_actor_forward(c));
  if (phase == 7 || phase == 1 || phase == -1) {
    CK(phase_actor_backward(c));
    CK(c.stamp(9));
  }
  if (phase == 2 || phase == -1) {          // both optimiser steps in one launch
    const ParamLayout& P = c.P;
    CK(drq_adam_flat2(c.p(P.seg[0]), c.g(P.seg[0]), s->adam_m + P.seg[0], s->adam_v + P.seg[0], P.seg[1] - P.seg[0],
                      s->step_enc, c.p(P.seg[4]), c.g(P.seg[4]), s->adam_m + P.seg[4], s->adam_v + P.seg[4],
                      P.seg[5] - P.seg[4], s->step_actor, s->lr, s->gscale, c.st));
  }
  if (phase == 8) CK(phase_encoder_opt(c));
  if (phase == 9) CK(phase_actor_opt(c));
  // the reference's method boundaries (DrQV2Agent.update_critic / update_actor, drqv2.py:177-228) cut phase 6 apart:
  if (phase == 10) {                        // critic_opt.step() alone (:201), no Polyak
    const ParamLayout& P = c.P;
    CK(drq_adam_flat(c.p(P.seg[2]), c.g(P.seg[2]), s->adam_m + P.seg[2], s->adam_v + P.seg[2], P.seg[3] - P.seg[2], s->lr,
                     s->step_critic, s->gscale, nullptr, 0.0, c.st));
  }
  if (phase == 11) CK(phase_actor_forward(c, false));   // actor loss through the (already stepped) critic (:210-216)
  if (phase == 12) {                        // utils.soft_update_params(critic, critic_target, tau) (:259-260)
    const ParamLayout& P = c.P;
    CK(drq_ema_flat(c.p(P.seg[2]), c.p(P.seg[6]), P.seg[3] - P.seg[2], s->tau, c.st));
  }
  if (phase == 13) {                        // the actor update's own draw (:210-211) from the stored policy output
    const int F = s->F, A = s->A, FA = F + A;
    CK(drq_trunc_normal_sample(c.ws(W_P3), s->noise_actor, s->std, s->clip, 1, c.ws(W_MU_O), c.ws(W_HA_C2) + F, FA, s->B,
                               A, c.st));
  }
  return 0;
}

DRQ_API int drq_publish_sums(const float* sums, float* sums_host, unsigned seq, drq_stream_t stream) {
  return publish_sums(sums, sums_host, seq, (hipStream_t)stream);
}

DRQ_API int drq_act_forward(const DrqStep* s, const uint8_t* obs, int n, float* mu_out) {
  CK(check_step(s));
  if (!obs || !mu_out || n <= 0 || n > 2 * s->B) return DRQ_EARG;
  Ctx c{s, param_layout(s->C, s->A, s->F, s->H), ws_layout(s->B, s->C, s->A, s->F, s->H), (hipStream_t)s->stream};
  const ParamLayout& P = c.P;
  const int F = s->F, H = s->H, A = s->A;
  CK(drq_u8_normalize(obs, c.ws(W_AUG), (long)n * s->C * 84 * 84, c.st));
  CK(encoder_forward(c, c.ws(W_AUG), n, c.ws(W_ACT1), c.ws(W_ACT2), c.ws(W_ACT3), c.ws(W_FEAT)));
  const HeadOff& a = P.actor;
  // n <= 2B rows: every buffer used below holds 2B rows
  {
    const float *x[1] = {c.ws(W_FEAT)}, *w[1] = {c.p(a.trunk_w)}, *b[1] = {c.p(a.trunk_b)};
    float* y[1] = {c.ws(W_Z4)};
    CK(c.fwd(1, x, R, w, b, y, F, n, F, (int)R, 0));
  }
  CK(drq_ln_tanh_fwd(c.ws(W_Z4), F, c.p(a.ln_g), c.p(a.ln_b), c.ws(W_HROWS), F, nullptr, nullptr, n, F, c.st));
  CK(policy_forward(c, c.ws(W_HROWS), n, c.ws(W_P1), c.ws(W_P2), c.ws(W_P3)));
  (void)H;
  return drq_tanh(c.ws(W_P3), mu_out, (long)n * A, c.st);
}

}  // extern "C"
