"""CPU restatement of the DrQ-v2 update step -- TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP hot path.  It is a from-scratch,
functional restatement (plain torch CPU tensor algebra, fp32 or fp64) of the
math that the reference reaches through torch.nn / torch.optim.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; nothing under ``drqv2_amd/``, ``drqv2.py`` or ``utils.py`` does.

Pinning: the reference has no tests or golden vectors of its own.  The oracle
is pinned against outputs of the reference itself, imported on CPU in the
build container by ``tests/golden/make_golden.py`` (fixtures committed under
``tests/golden/``), and ``tests/test_oracle_golden.py`` re-checks the oracle
against those fixtures wherever the tests run.

Reference citations (``/root/reference``):
  aug                drqv2.py:19-45      (RandomShiftsAug.forward)
  encoder            drqv2.py:63-67
  trunk / heads      drqv2.py:74-81, 85-93, 100-111, 115-121
  trunc-normal       utils.py:112-126
  critic update      drqv2.py:177-204
  actor update       drqv2.py:206-228
  update ordering    drqv2.py:230-262
  Adam               torch/optim/adam.py (single-tensor path; defaults)
  Polyak             utils.py:42-45
  schedule           utils.py:129-149
"""
import math
import re

import numpy as np
from collections import OrderedDict

import torch
import torch.nn.functional as F

ENC_KEYS = [f"convnet.{i}.{p}" for i in (0, 2, 4, 6) for p in ("weight", "bias")]
ACTOR_KEYS = (["trunk.0.weight", "trunk.0.bias", "trunk.1.weight", "trunk.1.bias"] +
              [f"policy.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")])
CRITIC_KEYS = (["trunk.0.weight", "trunk.0.bias", "trunk.1.weight", "trunk.1.bias"] +
               [f"{q}.{i}.{p}" for q in ("Q1", "Q2") for i in (0, 2, 4)
                for p in ("weight", "bias")])

LN_EPS = 1e-5          # torch.nn.LayerNorm default (drqv2.py:75,101)
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # torch.optim.Adam defaults


# --------------------------------------------------------------------------
# host scalar: utils.py:129-149
# --------------------------------------------------------------------------
def schedule(schdl, step):
    try:
        return float(schdl)
    except ValueError:
        pass
    m = re.match(r"linear\((.+),(.+),(.+)\)", schdl)
    if m:
        a, b, T = (float(g) for g in m.groups())
        mix = min(max(step / T, 0.0), 1.0)
        return (1.0 - mix) * a + mix * b
    m = re.match(r"step_linear\((.+),(.+),(.+),(.+),(.+)\)", schdl)
    if m:
        a, b1, T1, b2, T2 = (float(g) for g in m.groups())
        if step <= T1:
            mix = min(max(step / T1, 0.0), 1.0)
            return (1.0 - mix) * a + mix * b1
        mix = min(max((step - T1) / T2, 0.0), 1.0)
        return (1.0 - mix) * b1 + mix * b2
    raise NotImplementedError(schdl)


# --------------------------------------------------------------------------
# RandomShiftsAug: drqv2.py:19-45 restated as an explicit 4-tap gather
# --------------------------------------------------------------------------
def aug_base_grid(h, pad, dtype=torch.float32, device="cpu"):
    """First h entries of linspace(-1+1/S, 1-1/S, S), S=h+2*pad (drqv2.py:24-29)."""
    S = h + 2 * pad
    eps = 1.0 / S
    return torch.linspace(-1.0 + eps, 1.0 - eps, S, device=device, dtype=dtype)[:h]


def random_shifts_aug(x, shift_xy, pad=4, base=None):
    """x: [n,c,h,h] float (values 0..255); shift_xy: [n,2] integers in [0,2*pad]
    with column 0 = x/width shift and column 1 = y/height shift (drqv2.py:30-38).

    Restates pad(replicate) + grid_sample(bilinear, zeros, align_corners=False)
    as: un-normalise the grid coordinate (ATen GridSampler.h:27-36), floor, blend
    the four taps of the replicate-padded frame, taps outside [0,S-1] count 0."""
    n, c, h, w = x.shape
    assert h == w
    dt = x.dtype
    S = h + 2 * pad
    if base is None:
        base = aug_base_grid(h, pad, dt, x.device)
    sc = torch.tensor(2.0 / S, dtype=dt)
    sh = shift_xy.to(dt) * sc                                  # drqv2.py:39
    gx = base.view(1, 1, w) + sh[:, 0].view(n, 1, 1)           # [n,1,w]
    gy = base.view(1, h, 1) + sh[:, 1].view(n, 1, 1)           # [n,h,1]
    ix = ((gx + 1) * S - 1) / 2
    iy = ((gy + 1) * S - 1) / 2
    ix0 = torch.floor(ix)
    iy0 = torch.floor(iy)
    wx1 = ix - ix0
    wx0 = (ix0 + 1) - ix
    wy1 = iy - iy0
    wy0 = (iy0 + 1) - iy
    ix0 = ix0.long()
    iy0 = iy0.long()
    out = torch.zeros(n, c, h, w, dtype=dt)
    bidx = torch.arange(n).view(n, 1, 1)
    for dy, wy in ((0, wy0), (1, wy1)):           # order nw, ne, sw, se
        for dx, wx in ((0, wx0), (1, wx1)):
            px = (ix0 + dx).expand(n, h, w)
            py = (iy0 + dy).expand(n, h, w)
            ok = (px >= 0) & (px <= S - 1) & (py >= 0) & (py <= S - 1)
            sx = (px - pad).clamp(0, w - 1)       # replicate padding
            sy = (py - pad).clamp(0, h - 1)
            v = x[bidx.expand(n, h, w), :, sy, sx]            # [n,h,w,c]
            v = v.permute(0, 3, 1, 2)
            wgt = (wx * wy).expand(n, h, w) * ok.to(dt)
            out = out + v * wgt.unsqueeze(1)
    return out


def aug_integer_crop(x, shift_xy, pad=4):
    """Exact-arithmetic form of the same op: crop of the replicate-padded frame."""
    n, c, h, w = x.shape
    xp = F.pad(x, (pad,) * 4, mode="replicate")
    out = torch.empty_like(x)
    for i in range(n):
        sx, sy = int(shift_xy[i, 0]), int(shift_xy[i, 1])
        out[i] = xp[i, :, sy:sy + h, sx:sx + w]
    return out


# --------------------------------------------------------------------------
# networks (functional): drqv2.py:48-121
# --------------------------------------------------------------------------
def encoder_forward(p, obs, return_acts=False, normalized=False, relu_masks=None):
    """p: dict with ENC_KEYS.  obs: [B,C,84,84] float (0..255 scale; already /255-0.5 if normalized).
    relu_masks (tests only): four boolean tensors that replace the ReLU decisions of the four layers.  ReLU makes
    the gradient a discontinuous function of the pre-activations: an implementation that rounds differently
    flips the sign of a few near-zero pre-activations among millions, and each flip is an O(1) change of that
    element's gradient path (1e-3 normwise at batch 40-64).  With the decisions of the implementation under
    test injected, the remaining difference is rounding only."""
    x = obs if normalized else obs / 255.0 - 0.5                 # drqv2.py:64
    acts = [x]
    for li, i in enumerate((0, 2, 4, 6)):
        x = F.conv2d(x, p[f"convnet.{i}.weight"], p[f"convnet.{i}.bias"],
                     stride=2 if li == 0 else 1)
        x = torch.relu(x) if relu_masks is None else x * relu_masks[li].to(x.dtype)
        acts.append(x)
    feat = x.reshape(x.shape[0], -1)                             # drqv2.py:66
    return (feat, acts) if return_acts else feat


def layer_norm(x, g, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)                 # biased
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def trunk_forward(p, feat):
    z = feat @ p["trunk.0.weight"].t() + p["trunk.0.bias"]
    return torch.tanh(layer_norm(z, p["trunk.1.weight"], p["trunk.1.bias"]))


def mlp3(p, prefix, x, masks=None, rec=None):
    """masks (tests only): two boolean tensors replacing the ReLU decisions of the two hidden layers (same reason as
    encoder_forward's relu_masks).  rec (tests only): a list that receives, per hidden layer, (pre-activation, layer
    input) of THIS evaluation -- the pre-activations are the function's own, before any injected decision."""
    z1 = x @ p[f"{prefix}.0.weight"].t() + p[f"{prefix}.0.bias"]
    h1 = torch.relu(z1) if masks is None else z1 * masks[0].to(z1.dtype)
    z2 = h1 @ p[f"{prefix}.2.weight"].t() + p[f"{prefix}.2.bias"]
    h2 = torch.relu(z2) if masks is None else z2 * masks[1].to(z2.dtype)
    if rec is not None:
        rec.append((z1.detach(), x.detach()))
        rec.append((z2.detach(), h1.detach()))
    return h2 @ p[f"{prefix}.4.weight"].t() + p[f"{prefix}.4.bias"]


def actor_mu(p, feat):
    return torch.tanh(mlp3(p, "policy", trunk_forward(p, feat)))  # drqv2.py:86-89


def critic_q(p, feat, action, masks=None, rec=None):
    """masks / rec (tests only): {"Q1": (m1, m2), "Q2": (m1, m2)} / {"Q1": [], "Q2": []}, see mlp3."""
    h = trunk_forward(p, feat)
    ha = torch.cat([h, action], dim=-1)                           # drqv2.py:117
    mk = lambda q: None if masks is None else masks[q]
    rc = lambda q: None if rec is None else rec[q]
    return mlp3(p, "Q1", ha, mk("Q1"), rc("Q1")), mlp3(p, "Q2", ha, mk("Q2"), rc("Q2"))


def trunc_normal_sample(mu, noise, std, clip):
    """utils.py:112-126.  Value = clamp(mu + clamp(noise*std, +-clip), +-(1-1e-6));
    gradient w.r.t. mu is the identity (straight-through)."""
    eps = noise * torch.ones_like(mu).mul(std)
    if clip is not None:
        eps = torch.clamp(eps, -clip, clip)
    x = mu + eps
    lo = torch.tensor(-1.0 + 1e-6, dtype=mu.dtype)
    hi = torch.tensor(1.0 - 1e-6, dtype=mu.dtype)
    clamped = torch.maximum(torch.minimum(x, hi), lo)
    return x - x.detach() + clamped.detach()


def normal_log_prob(a, mu, std):
    var = std * std
    return -((a - mu) ** 2) / (2 * var) - math.log(std) - math.log(math.sqrt(2 * math.pi))


def normal_entropy(std):
    return 0.5 + 0.5 * math.log(2 * math.pi) + math.log(std)


# --------------------------------------------------------------------------
# optimiser / target update
# --------------------------------------------------------------------------
def _fma(a, b, c):
    """Single-rounding a*b+c.  For fp32 the product is exact in fp64; for fp64
    inputs this is the ordinary two-rounding expression."""
    if a.dtype == torch.float32:
        return (a.double() * b.double() + c.double()).float()
    return a * b + c


def adam_step(p, g, m, v, t, lr):
    """One torch.optim.Adam step (defaults), t = 1-based step count, in place.
    torch/optim/adam.py: lerp_ / mul_.addcmul_ / sqrt / div / add_ / addcdiv_.
    The rounding sequence below is the one torch's CPU kernels were measured to
    follow bit-for-bit (tests/golden/elementwise.npz):
      m = fma(1-b1, g-m, m);  v = fma(fl((1-b2)*g), g, fl(b2*v));
      denom = fl(fl(sqrt(v)/sqrt(bc2)) + eps);  p = fl(p + fl(fl(-lr/bc1 * m) / denom))."""
    dt = p.dtype
    c = lambda x: torch.tensor(x, dtype=dt)
    m.copy_(_fma(c(1 - ADAM_B1).expand_as(g), g - m, m))
    v.copy_(_fma(g * c(1 - ADAM_B2), g, v * c(ADAM_B2)))
    bc1 = 1 - ADAM_B1 ** t
    bc2 = 1 - ADAM_B2 ** t
    step_size = lr / bc1
    # IEEE-correct square root: torch's vectorised CPU sqrt is only ~1 ulp and differs between CPUs
    # (0.7 % of fp32 values on the build container), numpy's is the hardware sqrtps
    sq = torch.from_numpy(np.sqrt(v.detach().numpy())) if dt == torch.float32 else v.sqrt()
    denom = sq / c(bc2 ** 0.5) + c(ADAM_EPS)
    p.add_((m * c(-step_size)) / denom)


def polyak(p, t, tau):
    """utils.py:42-45: t <- tau*p + (1-tau)*t, two rounded products and an add."""
    dt = p.dtype
    t.copy_(p * torch.tensor(tau, dtype=dt) + t * torch.tensor(1 - tau, dtype=dt))


# --------------------------------------------------------------------------
# the agent
# --------------------------------------------------------------------------
def _orthogonal(rows, cols, gain, gen):
    a = torch.randn(max(rows, cols), min(rows, cols), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
    if rows < cols:
        q = q.t()
    return (gain * q[:rows, :cols]).contiguous()


def init_params(obs_shape, action_dim, feature_dim, hidden_dim, seed=0, dtype=torch.float32):
    """Synthetic weights of the architecture's shapes (orthogonal, zero bias,
    LN gamma=1 beta=0 as utils.py:52-61 leaves them).  NOT the reference RNG
    stream -- fixtures carry the reference's own weights where that matters."""
    gen = torch.Generator().manual_seed(seed)
    C = obs_shape[0]
    R = 32 * 35 * 35
    enc = OrderedDict()
    cin = C
    for i in (0, 2, 4, 6):
        enc[f"convnet.{i}.weight"] = _orthogonal(32, cin * 9, math.sqrt(2.0), gen).view(32, cin, 3, 3).to(dtype)
        enc[f"convnet.{i}.bias"] = torch.zeros(32, dtype=dtype)
        cin = 32

    def head(prefix_dims):
        d = OrderedDict()
        d["trunk.0.weight"] = _orthogonal(feature_dim, R, 1.0, gen).to(dtype)
        d["trunk.0.bias"] = torch.zeros(feature_dim, dtype=dtype)
        d["trunk.1.weight"] = torch.ones(feature_dim, dtype=dtype)
        d["trunk.1.bias"] = torch.zeros(feature_dim, dtype=dtype)
        for prefix, dims in prefix_dims:
            for i, (o, k) in zip((0, 2, 4), dims):
                d[f"{prefix}.{i}.weight"] = _orthogonal(o, k, 1.0, gen).to(dtype)
                d[f"{prefix}.{i}.bias"] = torch.zeros(o, dtype=dtype)
        return d

    H, Fd, A = hidden_dim, feature_dim, action_dim
    actor = head([("policy", [(H, Fd), (H, H), (A, H)])])
    critic = head([("Q1", [(H, Fd + A), (H, H), (1, H)]), ("Q2", [(H, Fd + A), (H, H), (1, H)])])
    return enc, actor, critic


class OracleAgent:
    """Stateful restatement of DrQV2Agent (drqv2.py:124-262) with every random
    draw injected (shifts as integers, noises as tensors), so that it can be
    replayed bit-for-bit next to the HIP path."""

    def __init__(self, enc, actor, critic, lr, critic_target_tau=0.01,
                 stddev_schedule="linear(1.0,0.1,500000)", stddev_clip=0.3,
                 update_every_steps=2, critic_target=None, dtype=torch.float32):
        cv = lambda d: OrderedDict((k, v.detach().clone().to(dtype)) for k, v in d.items())
        self.enc, self.actor, self.critic = cv(enc), cv(actor), cv(critic)
        self.critic_target = cv(critic_target if critic_target is not None else critic)
        self.lr, self.tau = lr, critic_target_tau
        self.stddev_schedule, self.stddev_clip = stddev_schedule, stddev_clip
        self.update_every_steps = update_every_steps
        self.dtype = dtype
        z = lambda d: OrderedDict((k, torch.zeros_like(v)) for k, v in d.items())
        self.m = {"enc": z(self.enc), "actor": z(self.actor), "critic": z(self.critic)}
        self.v = {"enc": z(self.enc), "actor": z(self.actor), "critic": z(self.critic)}
        self.t = {"enc": 0, "actor": 0, "critic": 0}
        self.last = {}

    # -- helpers ---------------------------------------------------------
    def _adam(self, name, params, grads):
        self.t[name] += 1
        for k in params:
            adam_step(params[k], grads[k], self.m[name][k], self.v[name][k], self.t[name], self.lr)

    def act_mean(self, obs_u8):
        feat = encoder_forward(self.enc, obs_u8.to(self.dtype).unsqueeze(0))
        return actor_mu(self.actor, feat)[0]

    # -- the hot path ----------------------------------------------------
    def update(self, batch, step, shifts_obs, shifts_next, noise_critic, noise_actor,
               aug_base=None, aug_override=None, enc_in_override=None, keep=False, relu_masks=None,
               critic_relu_masks=None):
        """batch = (obs u8 [B,C,84,84], action [B,A], reward [B,1], discount [B,1],
        next_obs u8).  Returns the 8-key metrics dict of drqv2.py (python floats)."""
        if step % self.update_every_steps != 0:
            return {}
        dt = self.dtype
        obs_u8, action, reward, discount, next_u8 = batch
        action, reward, discount = action.to(dt), reward.to(dt), discount.to(dt)
        noise_critic, noise_actor = noise_critic.to(dt), noise_actor.to(dt)
        std = schedule(self.stddev_schedule, step)
        clip = self.stddev_clip
        B = obs_u8.shape[0]
        metrics = {}

        normalized = enc_in_override is not None
        if normalized:          # the encoder inputs (after aug and /255-0.5) are injected
            obs_a, next_a = (t.to(dt) for t in enc_in_override)
        elif aug_override is not None:
            obs_a, next_a = (t.to(dt) for t in aug_override)
        else:
            base = None if aug_base is None else aug_base.to(dt)
            obs_a = random_shifts_aug(obs_u8.to(dt), shifts_obs, 4, base)       # drqv2.py:241
            next_a = random_shifts_aug(next_u8.to(dt), shifts_next, 4, base)    # drqv2.py:242

        req = lambda d: OrderedDict((k, v.detach().requires_grad_(True)) for k, v in d.items())
        enc, critic = req(self.enc), req(self.critic)
        feat, acts = encoder_forward(enc, obs_a, return_acts=True, normalized=normalized,
                                     relu_masks=relu_masks)                                 # :244
        with torch.no_grad():
            feat_next = encoder_forward(self.enc, next_a, normalized=normalized)            # :245-246
        metrics["batch_reward"] = reward.mean().item()

        # ---- critic step: drqv2.py:177-204
        with torch.no_grad():
            mu_n = actor_mu(self.actor, feat_next)
            a_next = trunc_normal_sample(mu_n, noise_critic, std, clip)
            tq1, tq2 = critic_q(self.critic_target, feat_next, a_next)
            target_q = reward + discount * torch.minimum(tq1, tq2)
        crit_rec = {"Q1": [], "Q2": []} if keep else None
        q1, q2 = critic_q(critic, feat, action, critic_relu_masks, crit_rec)
        critic_loss = ((q1 - target_q) ** 2).mean() + ((q2 - target_q) ** 2).mean()
        metrics["critic_target_q"] = target_q.mean().item()
        metrics["critic_q1"] = q1.mean().item()
        metrics["critic_q2"] = q2.mean().item()
        metrics["critic_loss"] = critic_loss.item()
        names = list(enc) + list(critic)
        gl = torch.autograd.grad(critic_loss, list(enc.values()) + list(critic.values()))
        g_enc = OrderedDict(zip(list(enc), gl[:len(enc)]))
        g_critic = OrderedDict(zip(list(critic), gl[len(enc):]))
        self._adam("critic", self.critic, g_critic)                              # :201
        self._adam("enc", self.enc, g_enc)                                       # :202

        # ---- actor step: drqv2.py:206-228 (features detached, critic already updated)
        featd = feat.detach()
        actor = req(self.actor)
        mu = actor_mu(actor, featd)
        a = trunc_normal_sample(mu, noise_actor, std, clip)
        logp = normal_log_prob(a, mu, std).sum(-1, keepdim=True)
        aq1, aq2 = critic_q(self.critic, featd, a)
        actor_loss = -torch.minimum(aq1, aq2).mean()
        ga = torch.autograd.grad(actor_loss, list(actor.values()))
        g_actor = OrderedDict(zip(list(actor), ga))
        self._adam("actor", self.actor, g_actor)
        metrics["actor_loss"] = actor_loss.item()
        metrics["actor_logprob"] = logp.mean().item()
        metrics["actor_ent"] = float(normal_entropy(std) * a.shape[-1])

        # ---- target: drqv2.py:259-260
        for k in self.critic:
            polyak(self.critic[k], self.critic_target[k], self.tau)

        if keep:
            self.last = dict(obs_a=obs_a, next_a=next_a, feat=featd, feat_next=feat_next,
                             acts=[t.detach() for t in acts], mu_next=mu_n, a_next=a_next,
                             target_q=target_q, q1=q1.detach(), q2=q2.detach(),
                             g_enc=g_enc, g_critic=g_critic, g_actor=g_actor,
                             mu=mu.detach(), a=a.detach(), aq1=aq1.detach(), aq2=aq2.detach(),
                             critic_pre=crit_rec)
        return metrics


# ------------------------------------------------------------------------------------------------
# replay sampling (replay_buffer.py:142-160 `_sample`): pinned by tests/golden/nstep.json
# ------------------------------------------------------------------------------------------------
def nstep_sample(episode, idx, nstep, gamma):
    """episode: dict of numpy arrays (observation [T+1,...], action [T+1,A], reward [T+1,1], discount [T+1,1]), idx the
    reference's index (1-based: 0 is the dummy reset transition).  float32 arithmetic, one rounding per operation,
    in the reference's order.  Returns (obs, action, reward, discount, next_obs)."""
    import numpy as np
    obs = episode["observation"][idx - 1]
    action = episode["action"][idx]
    next_obs = episode["observation"][idx + nstep - 1]
    reward = np.zeros_like(episode["reward"][idx])
    discount = np.ones_like(episode["discount"][idx])
    for i in range(nstep):
        reward = reward + discount * episode["reward"][idx + i]
        discount = discount * (episode["discount"][idx + i] * np.float32(gamma))
    return obs, action, reward, discount, next_obs
