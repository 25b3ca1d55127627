// Whole-step orchestration of DrQV2Agent.update (drqv2.py:230-262) on one HIP stream.
// Host code only: sequences the kernels of conv.hip / gemm.hip / elementwise.hip over a caller-owned
// workspace.  No allocation, no synchronisation, no global state.
#include "common.h"
#include "../../include/drqv2_hip.h"

namespace {

constexpr long R = 32L * 35 * 35;   // repr_dim (drqv2.py:53)
inline long al64(long x) { return (x + 63) & ~63L; }

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
  L.seg[3] = off;
  L.seg[4] = off;
  head(L.actor, 1, 0, A);
  L.seg[5] = off;
  L.seg[6] = off;
  head(L.target, 2, A, 1);
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
  W_XHAT_C = DRQ_WS_NBUF_PUBLIC, W_RSTD_C, W_XHAT_A, W_RSTD_A, W_Z_C2,
  W_PN1, W_PN2, W_PN3, W_PO1, W_PO2, W_PO3,       // policy activations (next / obs)
  W_T1, W_T2,                                     // target-Q hidden activations, reused by the actor step
  W_C1, W_C2,                                     // critic-Q hidden activations
  W_DC2, W_DC1, W_DHA, W_DLN, W_DPRE, W_DP2, W_DP1, W_DH_A, W_DA,
  W_GEMM_WS, W_CONV_WS, W_COUNT
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
  take(W_PN1, (long)B * H);
  take(W_PN2, (long)B * H);
  take(W_PN3, (long)B * A);
  take(W_PO1, (long)B * H);
  take(W_PO2, (long)B * H);
  take(W_PO3, (long)B * A);
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
  float* ws(int id) const { return s->ws + W.off[id]; }
  float* p(long off) const { return s->params + off; }
  float* g(long off) const { return s->grads + off; }
  float* gemm_ws() const { return ws(W_GEMM_WS); }
  size_t gemm_ws_bytes() const { return (size_t)16 * 1024 * 1024 * sizeof(float); }

  // y[b] = act(x W^T + bias), batched over nb heads with strides between heads
  int linear_fwd(const float* x, long ldx, long x_bs, const float* w, long w_bs, const float* bias, long bias_bs,
                 float* y, long ldy, long y_bs, int M, int N, int K, int nb, int relu) const {
    return drq_gemm_f32(x, ldx, 1, w, K, 1, y, ldy, M, N, K, nb, x_bs, w_bs, y_bs, bias, bias_bs, relu, nullptr, 0,
                        0, 0, 0, 0, gemm_ws(), gemm_ws_bytes(), st);
  }
  // dx[b] = (dy W) * (mask > 0)
  int linear_dgrad(const float* dy, long lddy, long dy_bs, const float* w, long ldw, long w_bs, float* dx, long lddx,
                   long dx_bs, int M, int Nout, int K, int nb, const float* mask, int ldmask, long mask_bs) const {
    return drq_gemm_f32(dy, lddy, 1, w, ldw, 0, dx, lddx, M, Nout, K, nb, dy_bs, w_bs, dx_bs, nullptr, 0, 0, mask,
                        ldmask, mask_bs, 0, 0, 0, gemm_ws(), gemm_ws_bytes(), st);
  }
  // dW[b] = dy^T x  ([N][K] row-major), db[b] = colsum(dy)
  int linear_wgrad(const float* dy, long lddy, long dy_bs, const float* x, long ldx, long x_bs, float* dw, long dw_bs,
                   float* db, long db_bs, int Brows, int N, int K, int nb) const {
    CK(drq_gemm_f32(dy, lddy, 0, x, ldx, 0, dw, K, N, K, Brows, nb, dy_bs, x_bs, dw_bs, nullptr, 0, 0, nullptr, 0, 0,
                    0, 0, 0, gemm_ws(), gemm_ws_bytes(), st));
    return drq_colsum(dy, lddy, dy_bs, db, db_bs, Brows, N, nb, st);
  }
};

int encoder_forward(const Ctx& c, const float* x, int nb, float* a1, float* a2, float* a3, float* a4) {
  const ParamLayout& P = c.P;
  const int C = c.s->C;
  (void)C;
  float* outs[4] = {a1, a2, a3, a4};
  const float* in = x;
  for (int l = 0; l < 4; ++l) {
    const int hin = kEncH[l], hout = kEncH[l + 1];
    CK(drq_conv3x3_fwd(in, c.p(P.enc_w[l]), c.p(P.enc_b[l]), outs[l], nb, l == 0 ? c.s->C : 32, hin, l == 0 ? 2 : 1,
                       1, 32L * hout * hout, (long)hout * hout, hout, 0, c.st));
    in = outs[l];
  }
  return 0;
}

// Q(h,a) for both heads of `net` on rows of `ha` -> q[2][B]; hidden activations into h1,h2 [2][B][H]
int q_forward(const Ctx& c, const HeadOff& net, const float* ha, float* h1, float* h2, float* q) {
  const DrqStep* s = c.s;
  const int B = s->B, H = s->H, FA = s->F + s->A;
  const long wbs = net.w[1][0] - net.w[0][0];   // distance between the Q1 and Q2 parameter blocks
  CK(c.linear_fwd(ha, FA, 0, c.p(net.w[0][0]), wbs, c.p(net.b[0][0]), wbs, h1, H, (long)B * H, B, H, FA, 2, 1));
  CK(c.linear_fwd(h1, H, (long)B * H, c.p(net.w[0][1]), wbs, c.p(net.b[0][1]), wbs, h2, H, (long)B * H, B, H, H, 2, 1));
  CK(c.linear_fwd(h2, H, (long)B * H, c.p(net.w[0][2]), wbs, c.p(net.b[0][2]), wbs, q, 1, B, B, 1, H, 2, 0));
  return 0;
}

int policy_forward(const Ctx& c, const float* h, float* p1, float* p2, float* p3) {
  const DrqStep* s = c.s;
  const HeadOff& a = c.P.actor;
  const int B = s->B, H = s->H, F = s->F, A = s->A;
  CK(c.linear_fwd(h, F, 0, c.p(a.w[0][0]), 0, c.p(a.b[0][0]), 0, p1, H, 0, B, H, F, 1, 1));
  CK(c.linear_fwd(p1, H, 0, c.p(a.w[0][1]), 0, c.p(a.b[0][1]), 0, p2, H, 0, B, H, H, 1, 1));
  CK(c.linear_fwd(p2, H, 0, c.p(a.w[0][2]), 0, c.p(a.b[0][2]), 0, p3, A, 0, B, A, H, 1, 0));
  return 0;
}

// ---- phase 0 ---------------------------------------------------------------------------------
int phase_critic(const Ctx& c) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, C = s->C, A = s->A, F = s->F, H = s->H, FA = F + A;
  hipStream_t st = c.st;
  float* aug = c.ws(W_AUG);
  float* feat = c.ws(W_FEAT);
  float* feat_obs = feat;
  float* feat_next = feat + (long)B * R;

  // aug (drqv2.py:241-242) + /255-0.5 (:64); rows [0,B) = obs, [B,2B) = next_obs
  CK(drq_aug_fwd(s->obs, s->shift_obs, s->base_grid, aug, B, C, 84, 4, 1, st));
  CK(drq_aug_fwd(s->next_obs, s->shift_next, s->base_grid, aug + (long)B * C * 84 * 84, B, C, 84, 4, 1, st));
  // encoder on both views in one pass (:244-246)
  CK(encoder_forward(c, aug, 2 * B, c.ws(W_ACT1), c.ws(W_ACT2), c.ws(W_ACT3), feat));

  // trunks: next -> (actor, target); obs -> (critic, actor)
  CK(c.linear_fwd(feat_next, R, 0, c.p(P.actor.trunk_w), P.target.trunk_w - P.actor.trunk_w, c.p(P.actor.trunk_b),
                  P.target.trunk_b - P.actor.trunk_b, c.ws(W_Z_NEXT), 2 * F, F, B, F, (int)R, 2, 0));
  CK(c.linear_fwd(feat_obs, R, 0, c.p(P.critic.trunk_w), P.actor.trunk_w - P.critic.trunk_w, c.p(P.critic.trunk_b),
                  P.actor.trunk_b - P.critic.trunk_b, c.ws(W_Z_OBS), 2 * F, F, B, F, (int)R, 2, 0));
  CK(drq_ln_tanh_fwd2(c.ws(W_Z_NEXT), c.ws(W_Z_NEXT) + F, 2 * F, c.p(P.actor.ln_g), c.p(P.actor.ln_b),
                      c.p(P.target.ln_g), c.p(P.target.ln_b), c.ws(W_H_AN), F, c.ws(W_HA_T), FA, nullptr, nullptr,
                      nullptr, nullptr, B, F, st));
  CK(drq_ln_tanh_fwd2(c.ws(W_Z_OBS), c.ws(W_Z_OBS) + F, 2 * F, c.p(P.critic.ln_g), c.p(P.critic.ln_b),
                      c.p(P.actor.ln_g), c.p(P.actor.ln_b), c.ws(W_HA_C), FA, c.ws(W_H_AO), F, c.ws(W_XHAT_C),
                      c.ws(W_RSTD_C), c.ws(W_XHAT_A), c.ws(W_RSTD_A), B, F, st));

  // target: a' ~ TruncN(actor(next)), y = r + d*min Q_target(next, a')   (:180-186)
  CK(policy_forward(c, c.ws(W_H_AN), c.ws(W_PN1), c.ws(W_PN2), c.ws(W_PN3)));
  CK(drq_trunc_normal_sample(c.ws(W_PN3), s->noise_critic, s->std, s->clip, 1, nullptr, c.ws(W_HA_T) + F, FA, B, A, st));
  CK(q_forward(c, P.target, c.ws(W_HA_T), c.ws(W_T1), c.ws(W_T2), c.ws(W_TQ)));
  // critic(obs, action) (:188)
  CK(drq_copy_cols(s->action, A, c.ws(W_HA_C) + F, FA, B, A, st));
  CK(q_forward(c, P.critic, c.ws(W_HA_C), c.ws(W_C1), c.ws(W_C2), c.ws(W_Q)));
  const float invB = 1.0f / (float)s->global_B;
  CK(drq_td_mse(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_Q), c.ws(W_Q) + B, s->reward, s->discount, c.ws(W_DQ),
                c.ws(W_DQ) + B, s->sums, B, invB, st));

  // ---- backward of the critic loss (:200)
  const HeadOff& cr = P.critic;
  const long wbs = cr.w[1][0] - cr.w[0][0];
  const long BH = (long)B * H;
  // layer 3: q = h2 W3^T + b3
  CK(c.linear_wgrad(c.ws(W_DQ), 1, B, c.ws(W_C2), H, BH, c.g(cr.w[0][2]), wbs, c.g(cr.b[0][2]), wbs, B, 1, H, 2));
  CK(c.linear_dgrad(c.ws(W_DQ), 1, B, c.p(cr.w[0][2]), H, wbs, c.ws(W_DC2), H, BH, B, H, 1, 2, c.ws(W_C2), H, BH));
  // layer 2
  CK(c.linear_wgrad(c.ws(W_DC2), H, BH, c.ws(W_C1), H, BH, c.g(cr.w[0][1]), wbs, c.g(cr.b[0][1]), wbs, B, H, H, 2));
  CK(c.linear_dgrad(c.ws(W_DC2), H, BH, c.p(cr.w[0][1]), H, wbs, c.ws(W_DC1), H, BH, B, H, H, 2, c.ws(W_C1), H, BH));
  // layer 1 (input = [h, action], shared by both heads)
  CK(c.linear_wgrad(c.ws(W_DC1), H, BH, c.ws(W_HA_C), FA, 0, c.g(cr.w[0][0]), wbs, c.g(cr.b[0][0]), wbs, B, H, FA, 2));
  CK(c.linear_dgrad(c.ws(W_DC1), H, BH, c.p(cr.w[0][0]), FA, wbs, c.ws(W_DHA), FA, (long)B * FA, B, FA, H, 2, nullptr,
                    0, 0));
  // trunk: LayerNorm+tanh backward, then Linear(R -> F)
  CK(drq_ln_tanh_bwd(c.ws(W_DHA), FA, c.ws(W_DHA) + (long)B * FA, FA, c.ws(W_HA_C), FA, c.ws(W_XHAT_C),
                     c.ws(W_RSTD_C), c.p(cr.ln_g), c.ws(W_DZ_C), c.ws(W_DLN), c.g(cr.ln_g), c.g(cr.ln_b), B, F, st));
  CK(c.linear_wgrad(c.ws(W_DZ_C), F, 0, feat_obs, R, 0, c.g(cr.trunk_w), 0, c.g(cr.trunk_b), 0, B, F, (int)R, 1));
  // d feat = dz W_t, masked by relu(conv4) and scattered into the padded conv-gradient layout
  CK(drq_gemm_f32(c.ws(W_DZ_C), F, 1, c.p(cr.trunk_w), R, 0, c.ws(W_DY4), 0, B, (int)R, F, 1, 0, 0, 0, nullptr, 0, 0,
                  feat_obs, (int)R, 0, 35, 0, 1, c.gemm_ws(), c.gemm_ws_bytes(), st));

  // ---- encoder backward: conv4 .. conv1 (wgrad all, dgrad 4..2)
  const int dyid[4] = {W_DY1, W_DY2, W_DY3, W_DY4};
  const int actid[4] = {W_AUG, W_ACT1, W_ACT2, W_ACT3};   // layer inputs
  float* cws = c.ws(W_CONV_WS);
  const size_t cws_bytes = drq_conv3x3_wgrad_ws_bytes();
  for (int l = 3; l >= 0; --l) {
    const int hin = kEncH[l], hout = kEncH[l + 1], hp = hout + 4;
    const float* dy = c.ws(dyid[l]);
    CK(drq_conv3x3_wgrad(c.ws(actid[l]), dy, c.g(P.enc_w[l]), c.g(P.enc_b[l]), B, l == 0 ? C : 32, hin, l == 0 ? 2 : 1,
                         32L * hp * hp, (long)hp * hp, hp, 2L * hp + 2, cws, cws_bytes, st));
    if (l >= 1) {
      const int hpi = hin + 4;   // padded size of the next (shallower) gradient buffer
      CK(drq_conv3x3_dgrad(dy, c.p(P.enc_w[l]), c.ws(actid[l]), c.ws(dyid[l - 1]), B, hout, 32L * hpi * hpi,
                           (long)hpi * hpi, hpi, 2L * hpi + 2, st));
    }
  }
  return 0;
}

// ---- phase 1 ---------------------------------------------------------------------------------
int phase_actor(const Ctx& c) {
  const DrqStep* s = c.s;
  const ParamLayout& P = c.P;
  const int B = s->B, A = s->A, F = s->F, H = s->H, FA = F + A;
  hipStream_t st = c.st;
  float* feat_obs = c.ws(W_FEAT);

  // critic_opt.step(); encoder_opt.step() (:201-202) + Polyak (:259-260, fused: it reads the stepped critic)
  CK(drq_adam_flat(c.p(P.seg[2]), c.g(P.seg[2]), s->adam_m + P.seg[2], s->adam_v + P.seg[2], P.seg[3] - P.seg[2],
                   s->lr, s->step_critic, s->gscale, c.p(P.seg[6]), s->tau, st));
  CK(drq_adam_flat(c.p(P.seg[0]), c.g(P.seg[0]), s->adam_m + P.seg[0], s->adam_v + P.seg[0], P.seg[1] - P.seg[0],
                   s->lr, s->step_enc, s->gscale, nullptr, 0.0, st));

  // actor(obs.detach()) (:210-211)
  CK(policy_forward(c, c.ws(W_H_AO), c.ws(W_PO1), c.ws(W_PO2), c.ws(W_PO3)));
  CK(drq_trunc_normal_sample(c.ws(W_PO3), s->noise_actor, s->std, s->clip, 1, c.ws(W_MU_O), c.ws(W_HA_C2) + F, FA, B,
                             A, st));
  // updated critic on (obs, a) (:213)
  const HeadOff& cr = P.critic;
  CK(c.linear_fwd(feat_obs, R, 0, c.p(cr.trunk_w), 0, c.p(cr.trunk_b), 0, c.ws(W_Z_C2), F, 0, B, F, (int)R, 1, 0));
  CK(drq_ln_tanh_fwd(c.ws(W_Z_C2), F, c.p(cr.ln_g), c.p(cr.ln_b), c.ws(W_HA_C2), FA, nullptr, nullptr, B, F, st));
  CK(q_forward(c, cr, c.ws(W_HA_C2), c.ws(W_T1), c.ws(W_T2), c.ws(W_TQ)));
  const float invB = 1.0f / (float)s->global_B;
  CK(drq_actor_loss(c.ws(W_TQ), c.ws(W_TQ) + B, c.ws(W_HA_C2) + F, FA, c.ws(W_MU_O), s->std, c.ws(W_DQ),
                    c.ws(W_DQ) + B, s->sums, B, A, invB, st));

  // backward through the critic to the action only (critic weight grads are never used: SURVEY A7(iii))
  const long wbs = cr.w[1][0] - cr.w[0][0];
  const long BH = (long)B * H;
  CK(c.linear_dgrad(c.ws(W_DQ), 1, B, c.p(cr.w[0][2]), H, wbs, c.ws(W_DC2), H, BH, B, H, 1, 2, c.ws(W_T2), H, BH));
  CK(c.linear_dgrad(c.ws(W_DC2), H, BH, c.p(cr.w[0][1]), H, wbs, c.ws(W_DC1), H, BH, B, H, H, 2, c.ws(W_T1), H, BH));
  CK(c.linear_dgrad(c.ws(W_DC1), H, BH, c.p(cr.w[0][0]) + F, FA, wbs, c.ws(W_DA), A, (long)B * A, B, A, H, 2, nullptr,
                    0, 0));
  CK(drq_actor_dmu(c.ws(W_DA), c.ws(W_DA) + (long)B * A, A, 0, c.ws(W_MU_O), c.ws(W_DPRE), B, A, st));

  // policy MLP backward
  const HeadOff& ac = P.actor;
  CK(c.linear_wgrad(c.ws(W_DPRE), A, 0, c.ws(W_PO2), H, 0, c.g(ac.w[0][2]), 0, c.g(ac.b[0][2]), 0, B, A, H, 1));
  CK(c.linear_dgrad(c.ws(W_DPRE), A, 0, c.p(ac.w[0][2]), H, 0, c.ws(W_DP2), H, 0, B, H, A, 1, c.ws(W_PO2), H, 0));
  CK(c.linear_wgrad(c.ws(W_DP2), H, 0, c.ws(W_PO1), H, 0, c.g(ac.w[0][1]), 0, c.g(ac.b[0][1]), 0, B, H, H, 1));
  CK(c.linear_dgrad(c.ws(W_DP2), H, 0, c.p(ac.w[0][1]), H, 0, c.ws(W_DP1), H, 0, B, H, H, 1, c.ws(W_PO1), H, 0));
  CK(c.linear_wgrad(c.ws(W_DP1), H, 0, c.ws(W_H_AO), F, 0, c.g(ac.w[0][0]), 0, c.g(ac.b[0][0]), 0, B, H, F, 1));
  CK(c.linear_dgrad(c.ws(W_DP1), H, 0, c.p(ac.w[0][0]), F, 0, c.ws(W_DH_A), F, 0, B, F, H, 1, nullptr, 0, 0));
  CK(drq_ln_tanh_bwd(c.ws(W_DH_A), F, nullptr, 0, c.ws(W_H_AO), F, c.ws(W_XHAT_A), c.ws(W_RSTD_A), c.p(ac.ln_g),
                     c.ws(W_DZ_A), c.ws(W_DLN), c.g(ac.ln_g), c.g(ac.ln_b), B, F, st));
  CK(c.linear_wgrad(c.ws(W_DZ_A), F, 0, feat_obs, R, 0, c.g(ac.trunk_w), 0, c.g(ac.trunk_b), 0, B, F, (int)R, 1));
  return 0;
}

int phase_actor_opt(const Ctx& c) {
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
  if (s->ws_bytes < drq_step_ws_bytes(s->B, s->C, s->A, s->F, s->H)) return DRQ_EWS;
  return 0;
}

}  // namespace

extern "C" {

int drq_abi_version(void) { return 1; }

int drq_param_layout(int C, int A, int F, int H, long* out, int cap) {
  if (!out || cap < DRQ_PARAM_LAYOUT_LEN || C <= 0 || A <= 0 || F <= 0 || H <= 0) return DRQ_EARG;
  const ParamLayout L = param_layout(C, A, F, H);
  int n = 0;
  for (int i = 0; i < L.nflat; ++i) out[n++] = L.flat[i];
  for (int i = 0; i < 8; ++i) out[n++] = L.seg[i];
  out[n++] = L.total;
  return n;
}

size_t drq_step_ws_bytes(int B, int C, int A, int F, int H) {
  if (B <= 0 || C <= 0 || A <= 0 || F <= 0 || H <= 0) return 0;
  return (size_t)ws_layout(B, C, A, F, H).total * sizeof(float);
}

long drq_step_ws_offset(int B, int C, int A, int F, int H, int id) {
  if (id < 0 || id >= DRQ_WS_NBUF_PUBLIC) return -1;
  return ws_layout(B, C, A, F, H).off[id];
}

int drq_update_phase(const DrqStep* s, int phase) {
  CK(check_step(s));
  if (!s->obs || !s->next_obs || !s->action || !s->reward || !s->discount || !s->shift_obs || !s->shift_next ||
      !s->noise_critic || !s->noise_actor || !s->base_grid || !s->grads || !s->adam_m || !s->adam_v || !s->sums)
    return DRQ_EARG;
  Ctx c{s, param_layout(s->C, s->A, s->F, s->H), ws_layout(s->B, s->C, s->A, s->F, s->H), (hipStream_t)s->stream};
  if (phase == 0 || phase == -1) CK(phase_critic(c));
  if (phase == 1 || phase == -1) CK(phase_actor(c));
  if (phase == 2 || phase == -1) CK(phase_actor_opt(c));
  if (phase < -1 || phase > 2) return DRQ_EARG;
  return 0;
}

int drq_act_forward(const DrqStep* s, const uint8_t* obs, int n, float* mu_out) {
  CK(check_step(s));
  if (!obs || !mu_out || n <= 0 || n > 2 * s->B) return DRQ_EARG;
  Ctx c{s, param_layout(s->C, s->A, s->F, s->H), ws_layout(s->B, s->C, s->A, s->F, s->H), (hipStream_t)s->stream};
  const ParamLayout& P = c.P;
  const int F = s->F, H = s->H, A = s->A;
  CK(drq_u8_normalize(obs, c.ws(W_AUG), (long)n * s->C * 84 * 84, c.st));
  CK(encoder_forward(c, c.ws(W_AUG), n, c.ws(W_ACT1), c.ws(W_ACT2), c.ws(W_ACT3), c.ws(W_FEAT)));
  const HeadOff& a = P.actor;
  // n <= 2B rows: z and the hidden activations fit the [2][B] sized buffers
  CK(c.linear_fwd(c.ws(W_FEAT), R, 0, c.p(a.trunk_w), 0, c.p(a.trunk_b), 0, c.ws(W_Z_OBS), F, 0, n, F, (int)R, 1, 0));
  CK(drq_ln_tanh_fwd(c.ws(W_Z_OBS), F, c.p(a.ln_g), c.p(a.ln_b), c.ws(W_Z_NEXT), F, nullptr, nullptr, n, F, c.st));
  CK(c.linear_fwd(c.ws(W_Z_NEXT), F, 0, c.p(a.w[0][0]), 0, c.p(a.b[0][0]), 0, c.ws(W_T1), H, 0, n, H, F, 1, 1));
  CK(c.linear_fwd(c.ws(W_T1), H, 0, c.p(a.w[0][1]), 0, c.p(a.b[0][1]), 0, c.ws(W_T2), H, 0, n, H, H, 1, 1));
  CK(c.linear_fwd(c.ws(W_T2), H, 0, c.p(a.w[0][2]), 0, c.p(a.b[0][2]), 0, c.ws(W_DA), A, 0, n, A, H, 1, 0));
  return drq_tanh(c.ws(W_DA), mu_out, (long)n * A, c.st);
}

}  // extern "C"
