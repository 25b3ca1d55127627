"""Step-level parity: DrQV2Agent.update() on the GPU (HIP path, through libdrqv2_hip.so) against the
CPU oracle on identical replay samples, shifts and noise, and against the reference's own outputs
(tests/golden/steps.json.gz).  Tolerances: SURVEY.md App. B / BASELINE.md section 4."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from drqv2_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

CASES = {
    # name: C, A, F, H, B, lr, sched, wseed, bseed, updates, step0, smooth   (same as make_golden.py)
    "cheetah_b8": dict(C=9, A=6, F=50, H=1024, B=8, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=0, bseed=0,
                       updates=3, step0=0, smooth=True),
    "humanoid_b4": dict(C=9, A=21, F=100, H=1024, B=4, lr=8e-5, sched="linear(1.0,0.1,2000000)", wseed=1, bseed=10,
                        updates=2, step0=1000, smooth=True),
    "cartpole_b32": dict(C=9, A=1, F=50, H=1024, B=32, lr=1e-4, sched="linear(1.0,0.1,100000)", wseed=2, bseed=20,
                         updates=2, step0=50000, smooth=False),
    "small_h64_b6": dict(C=9, A=3, F=20, H=64, B=6, lr=1e-3, sched="0.2", wseed=3, bseed=30, updates=3, step0=0,
                         smooth=True),
}


def nerr(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def make_agent(cfg):
    import drqv2
    ag = drqv2.DrQV2Agent((cfg["C"], 84, 84), (cfg["A"],), "cuda", cfg["lr"], cfg["F"], cfg["H"], 0.01, 2000, 2,
                          cfg["sched"], 0.3, True)
    enc, actor, critic = synth.make_weights(cfg["C"], cfg["A"], cfg["F"], cfg["H"], cfg["wseed"])
    ag.encoder.load_state_dict(enc)
    ag.actor.load_state_dict(actor)
    ag.critic.load_state_dict(critic)
    ag.critic_target.load_state_dict(critic)
    # the fused aug+conv1 kernel keeps only the obs view's encoder input (conv1's weight gradient needs it); the
    # tests also look at the next_obs view, so they ask the same kernel to store it as well
    ag._engine.store_aug_next = True
    return ag


def make_oracle(cfg, dtype):
    from oracle import drq_oracle as O
    enc, actor, critic = synth.make_weights(cfg["C"], cfg["A"], cfg["F"], cfg["H"], cfg["wseed"])
    return O.OracleAgent(enc, actor, critic, cfg["lr"], stddev_schedule=cfg["sched"], dtype=dtype)


def run_hip(ag, cfg, u):
    batch = synth.make_batch(cfg["B"], cfg["A"], cfg["C"], seed=cfg["bseed"] + u, smooth=cfg["smooth"])
    draws = synth.make_draws(cfg["B"], cfg["A"], seed=cfg["bseed"] + u)
    ag._draw_hook = lambda n, A: tuple(t.float().cuda() for t in draws)
    m = ag.update(iter([tuple(x.numpy() for x in batch)]), cfg["step0"] + 2 * u)
    return m, batch, draws


def check_encoder_inputs_bitwise(ag, cfg, batch, sh_o, sh_n):
    """The in-step aug + /255-0.5 buffer (both views, the stacked launch the update really issues) against the
    oracle's RandomShiftsAug restatement: every element identical.  Returns the buffer on the CPU."""
    from oracle import drq_oracle as O
    B = cfg["B"]
    base = O.aug_base_grid(84, 4)
    xin = ag._engine.ws_view("AUG", B, (2 * B, cfg["C"], 84, 84)).cpu()
    for view, frames, sh in ((xin[:B], batch[0], sh_o), (xin[B:], batch[4], sh_n)):
        want = O.random_shifts_aug(frames.float(), sh, 4, base) / 255.0 - 0.5
        ndiff = int((view != want).sum())
        assert ndiff == 0, (ndiff, float((view - want).abs().max()))
    return xin


def check_relu_decisions(ag, cfg, o64, xin_obs):
    """Run the fp64 oracle's OWN encoder (its own ReLU decisions) on the obs view and compare every ReLU decision
    of the HIP encoder with it.  A disagreement is only legitimate where fp32 cannot decide the sign: the fp64
    pre-activation must lie inside the fp32 forward-error bound of its own dot product,
    |pre| <= 4 * gamma_n * (|w| * |x| + |b|), gamma_n = n*2^-24, n = 9*Cin+1 (the factor 4 covers the
    rounding already carried by the layer's inputs).  The number of such units is bounded too.
    Returns (#disagreements, #units)."""
    import torch.nn.functional as Fn
    B = cfg["B"]
    eng = ag._engine
    hs = (41, 39, 37, 35)
    hip = [eng.ws_view(nm, B, (2 * B, 32, h, h))[:B].cpu() for nm, h in zip(("ACT1", "ACT2", "ACT3"), hs)]
    hip.append(eng.ws_view("FEAT", B, (2 * B, 32, 35, 35))[:B].cpu())
    from oracle import drq_oracle as O
    feat, acts = O.encoder_forward(o64.enc, xin_obs.double(), return_acts=True, normalized=True)
    flips = total = 0
    for li, i in enumerate((0, 2, 4, 6)):
        w, b = o64.enc[f"convnet.{i}.weight"], o64.enc[f"convnet.{i}.bias"]
        st = 2 if li == 0 else 1
        pre = Fn.conv2d(acts[li], w, b, stride=st)
        dis = (pre > 0) != (hip[li] > 0)
        n = int(dis.sum())
        total += pre.numel()
        if n:
            mag = Fn.conv2d(acts[li].abs(), w.abs(), b.abs(), stride=st)
            gamma = (9 * w.shape[1] + 1) * 2.0 ** -24
            worst = float((pre.abs()[dis] / (4 * gamma * mag[dis])).max())
            assert worst <= 1.0, (li, n, worst)
            # where the HIP side says "on", the value it kept is rounding-sized as well
            assert float(hip[li][dis].abs().max()) <= float((8 * gamma * mag[dis]).max())
        flips += n
    # measured: 0-3 per update at B<=512 (~1e7..7e7 units); anything systematic would be thousands
    assert flips <= max(8, total // 2_000_000), (flips, total)
    # with its own decisions the oracle's features still agree with the HIP features (a flipped unit is ~0 either way)
    feat_hip = eng.ws_view("FEAT", B, (2 * B, 39200))[:B]
    assert nerr(feat_hip, feat) <= 2e-6
    return flips, total


def hip_masks(ag, cfg):
    """The ReLU decisions the HIP update took: the four encoder layers (obs view) and the two hidden layers of both
    Q heads in the critic loss.  Handed to the oracle for the GRADIENT comparison only (gradients are a discontinuous
    function of these decisions: one flipped unit among millions moves a whole gradient tensor by 1e-4..1e-3
    normwise); check_relu_decisions / check_critic_decisions bound how many differ from the oracle's own."""
    B, H = cfg["B"], cfg["H"]
    eng = ag._engine
    enc = [eng.ws_view(nm, B, (2 * B, 32, h, h))[:B].cpu() > 0 for nm, h in zip(("ACT1", "ACT2", "ACT3"), (41, 39, 37))]
    enc.append(eng.ws_view("FEAT", B, (2 * B, 32, 35, 35))[:B].cpu() > 0)
    c1 = eng.ws_view("C1", B, (2, B, H)).cpu() > 0
    c2 = eng.ws_view("C2", B, (2, B, H)).cpu() > 0
    return enc, {"Q1": (c1[0], c2[0]), "Q2": (c1[1], c2[1])}


def check_critic_decisions(o64, critic_before, crit_masks):
    """ReLU decisions of the critic's hidden layers (critic loss forward): the fp64 oracle's own pre-activations
    (o64.last["critic_pre"], recorded before any injected decision is applied) against the HIP decisions.  A
    disagreement must sit inside the fp32 forward-error bound of its dot product; their number is bounded."""
    flips = total = 0
    for q in ("Q1", "Q2"):
        for li, (z, x) in enumerate(o64.last["critic_pre"][q]):
            w, b = critic_before[f"{q}.{2 * li}.weight"], critic_before[f"{q}.{2 * li}.bias"]
            dis = (z > 0) != crit_masks[q][li]
            n = int(dis.sum())
            total += z.numel()
            if n:
                mag = x.abs() @ w.abs().t() + b.abs()
                gamma = (w.shape[1] + 1) * 2.0 ** -24
                worst = float((z.abs()[dis] / (4 * gamma * mag[dis])).max())
                assert worst <= 1.0, (q, li, n, worst)
            flips += n
    assert flips <= max(4, total // 250_000), (flips, total)
    return flips, total


def sync_oracle_state(o, ag):
    """Copy the HIP agent's complete training state (weights, target, Adam moments and step counts) into an
    oracle, so that update k>1 is compared from IDENTICAL state instead of through k-1 updates of drift."""
    dt = o.dtype
    eng = ag._engine
    for name, mod, net in (("enc", ag.encoder, "enc"), ("actor", ag.actor, "actor"), ("critic", ag.critic, "critic")):
        dst = getattr(o, name)
        for (k, p), off in zip(mod.named_parameters(), eng.layout[net]):
            n = p.numel()
            dst[k] = p.detach().cpu().to(dt).clone()
            o.m[name][k] = eng.adam_m[off:off + n].view(p.shape).cpu().to(dt).clone()
            o.v[name][k] = eng.adam_v[off:off + n].view(p.shape).cpu().to(dt).clone()
    for k, p in ag.critic_target.named_parameters():
        o.critic_target[k] = p.detach().cpu().to(dt).clone()
    o.t = {"enc": ag.encoder_opt.t, "actor": ag.actor_opt.t, "critic": ag.critic_opt.t}


@pytest.fixture(scope="module")
def golden_steps():
    with gzip.open(os.path.join(G, "steps.json.gz"), "rt") as f:
        return json.load(f)


@pytest.mark.parametrize("name", list(CASES))
def test_update_matches_oracle(name, golden_steps):
    cfg = CASES[name]
    from oracle import drq_oracle as O
    ag = make_agent(cfg)
    o32, o64 = make_oracle(cfg, torch.float32), make_oracle(cfg, torch.float64)
    for u in range(cfg["updates"]):
        step = cfg["step0"] + 2 * u
        if u > 0:
            # later updates start from the HIP agent's own state (weights, target, Adam moments, step counts):
            # every update is then held to the first-update tolerances, not to 3e-3 through accumulated drift
            sync_oracle_state(o32, ag)
            sync_oracle_state(o64, ag)
        m, batch, (sh_o, sh_n, n_c, n_a) = run_hip(ag, cfg, u)
        # both oracles get the encoder inputs the HIP step produced (aug + /255-0.5; that op is pinned
        # on its own in test_hip_ops): everything downstream then sees identical upstream tensors
        Bq = cfg["B"]
        xin = check_encoder_inputs_bitwise(ag, cfg, batch, sh_o, sh_n)     # so injecting them changes nothing
        ov = (xin[:Bq], xin[Bq:])
        # ... and the ReLU decisions of the HIP step (encoder, critic hidden layers): a pre-activation within
        # rounding of zero can fall on either side in two correct fp32 evaluations, and ONE flipped unit at these
        # batch sizes moves the first conv layer's gradient by ~5e-4 (see oracle.encoder_forward)
        check_relu_decisions(ag, cfg, o64, xin[:Bq])
        acts, crit_masks = hip_masks(ag, cfg)
        critic_before = {k: v.clone() for k, v in o64.critic.items()}
        kw = dict(enc_in_override=ov, keep=True, relu_masks=acts, critic_relu_masks=crit_masks)
        m32 = o32.update(batch, step, sh_o, sh_n, n_c, n_a, **kw)
        m64 = o64.update(batch, step, sh_o, sh_n, n_c, n_a, **kw)
        check_critic_decisions(o64, critic_before, crit_masks)
        assert list(m.keys()) == list(m64.keys())
        for k in m64:
            assert m[k] == pytest.approx(m64[k], rel=1e-5, abs=1e-5), (u, k, m[k], m32[k], m64[k])
        B = cfg["B"]
        eng = ag._engine
        # forward: encoder features of both views
        feat = eng.ws_view("FEAT", B, (2 * B, 39200))
        assert nerr(feat[:B], o64.last["feat"]) <= 2e-6
        assert nerr(feat[B:], o64.last["feat_next"]) <= 2e-6
        # gradients (identical upstream state): error vs fp64 no worse than the fp32 oracle's own
        for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"),
                             ("actor", ag.actor, "g_actor")):
            for (pn, p), g64, g32 in zip(mod.named_parameters(), o64.last[key].values(), o32.last[key].values()):
                e_hip, e_o32 = nerr(p.grad, g64), nerr(g32, g64)
                lim = max(2.0 * e_o32, 2e-5 if nm != "actor" else 2e-3)
                assert e_hip <= lim, (u, nm, pn, e_hip, e_o32)
    # reference's own first-update metrics (real grid_sample aug), fixtures from make_golden.py
    ref = golden_steps[name]["ref_fp32_aug"][0]["metrics"]
    ag2 = make_agent(cfg)
    m0, _, _ = run_hip(ag2, cfg, 0)
    for k, v in ref.items():
        assert m0[k] == pytest.approx(v, rel=2e-5, abs=2e-5), (k, m0[k], v)


WIDE = {
    # batch sizes that are multiples of 32 reach the kernels the training shapes use (gemm2.hip, skinny.hip,
    # full conv tiles); not in the golden file: checked against the pinned oracle directly
    "cheetah_b64": dict(C=9, A=6, F=50, H=1024, B=64, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=4, bseed=40,
                        updates=1, step0=0, smooth=True),
    "humanoid_b32": dict(C=9, A=21, F=100, H=1024, B=32, lr=8e-5, sched="linear(1.0,0.1,2000000)", wseed=5, bseed=50,
                         updates=1, step0=1000, smooth=True),
    # the training shapes themselves (BASELINE configs[1] at batch 256, and the per-rank batch of a 2-GPU strong
    # split): the trunk kernels (gemm2.hip: K = batch 128 / 256), the mask-free forward GEMM, full conv tile runs
    "cheetah_b128": dict(C=9, A=6, F=50, H=1024, B=128, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=6, bseed=60,
                         updates=1, step0=0, smooth=True),
    "cheetah_b256": dict(C=9, A=6, F=50, H=1024, B=256, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=7, bseed=70,
                         updates=1, step0=0, smooth=True),
    # feature_dim 100 (BASELINE configs[3]/[4] shapes): four column tiles per wave in the trunk forward, four row
    # tiles in its weight gradient
    "humanoid_b128": dict(C=9, A=21, F=100, H=1024, B=128, lr=8e-5, sched="linear(1.0,0.1,2000000)", wseed=8, bseed=80,
                          updates=1, step0=1000, smooth=True),
    # BASELINE configs[2] at its full shape: quadruped_walk, A=12, batch_size=512
    "quadruped_b512": dict(C=9, A=12, F=50, H=1024, B=512, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=9, bseed=90,
                           updates=1, step0=0, smooth=True),
    # BASELINE configs[3] at its full single-GPU shape: humanoid_run, A=21, feature_dim=100, batch_size=256
    "humanoid_b256": dict(C=9, A=21, F=100, H=1024, B=256, lr=8e-5, sched="linear(1.0,0.1,2000000)", wseed=10,
                          bseed=100, updates=1, step0=1000, smooth=True),
}


@pytest.mark.parametrize("name", list(WIDE))
def test_wide_batch_update_matches_oracle(name):
    cfg = WIDE[name]
    from oracle import drq_oracle as O
    ag = make_agent(cfg)
    o32, o64 = make_oracle(cfg, torch.float32), make_oracle(cfg, torch.float64)
    m, batch, (sh_o, sh_n, n_c, n_a) = run_hip(ag, cfg, 0)
    B = cfg["B"]
    eng = ag._engine
    # (a) the encoder inputs the update produced (stacked two-view launch) equal the oracle's aug bit for bit, so
    # handing them to the oracle below injects nothing the oracle would not have computed itself
    xin = check_encoder_inputs_bitwise(ag, cfg, batch, sh_o, sh_n)
    ov = (xin[:B], xin[B:])
    # (b) the oracle's OWN ReLU decisions against the HIP encoder's: counted and bounded, not waived
    flips, units = check_relu_decisions(ag, cfg, o64, xin[:B])
    print(f"{name}: {flips} ReLU decisions of {units} differ from the fp64 oracle's (all inside the fp32 error bound)")
    # (c) gradients are a discontinuous function of those decisions (each flipped unit is an O(1) change on its
    # path: 1e-3 normwise on the conv gradients), so the gradient comparison hands the HIP decisions to the oracle
    acts, crit_masks = hip_masks(ag, cfg)
    critic_before = {k: v.clone() for k, v in o64.critic.items()}
    kw = dict(enc_in_override=ov, keep=True, relu_masks=acts, critic_relu_masks=crit_masks)
    m32 = o32.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, **kw)
    m64 = o64.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, **kw)
    cflips, cunits = check_critic_decisions(o64, critic_before, crit_masks)
    print(f"{name}: {cflips} of {cunits} critic hidden-layer decisions differ (inside the fp32 error bound)")
    for k in m64:
        assert m[k] == pytest.approx(m64[k], rel=1e-5, abs=1e-5), (k, m[k], m32[k], m64[k])
    feat = eng.ws_view("FEAT", B, (2 * B, 39200))
    assert nerr(feat[:B], o64.last["feat"]) <= 2e-6
    for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"),
                         ("actor", ag.actor, "g_actor")):
        for (pn, p), g64, g32 in zip(mod.named_parameters(), o64.last[key].values(), o32.last[key].values()):
            e_hip, e_o32 = nerr(p.grad, g64), nerr(g32, g64)
            lim = max(2.0 * e_o32, 2e-5 if nm != "actor" else 2e-3)
            assert e_hip <= lim, (nm, pn, e_hip, e_o32)
    # parameters after the step: the oracle's Adam fed with the HIP gradients reproduces them bit for bit
    for k, p in ag.actor.named_parameters():
        p0 = synth.make_weights(cfg["C"], cfg["A"], cfg["F"], cfg["H"], cfg["wseed"])[1][k].clone()
        mm, vv = torch.zeros_like(p0), torch.zeros_like(p0)
        O.adam_step(p0, p.grad.detach().cpu(), mm, vv, 1, cfg["lr"])
        assert torch.equal(p.detach().cpu(), p0), k


def test_params_after_update_match_oracle_with_same_grads():
    """Adam + Polyak inside the step: feed the oracle's Adam the HIP gradients -> identical parameters."""
    cfg = CASES["small_h64_b6"]
    from oracle import drq_oracle as O
    ag = make_agent(cfg)
    before = {n: {k: v.detach().cpu().clone() for k, v in getattr(ag, n).state_dict().items()}
              for n in ("encoder", "critic", "actor", "critic_target")}
    run_hip(ag, cfg, 0)
    for n in ("encoder", "critic", "actor"):
        mod = getattr(ag, n)
        for (k, p) in mod.named_parameters():
            p0 = before[n][k].clone()
            m, v = torch.zeros_like(p0), torch.zeros_like(p0)
            O.adam_step(p0, p.grad.detach().cpu(), m, v, 1, cfg["lr"])
            assert torch.equal(p.detach().cpu(), p0), (n, k)
            if n == "critic":
                t0 = before["critic_target"][k].clone()
                O.polyak(p0, t0, 0.01)
                assert torch.equal(dict(ag.critic_target.named_parameters())[k].detach().cpu(), t0), k


def test_gating_and_metric_keys():
    cfg = CASES["small_h64_b6"]
    ag = make_agent(cfg)
    assert ag.update(iter([]), 1) == {}
    m, _, _ = run_hip(ag, cfg, 0)
    with open(os.path.join(G, "interface.json")) as f:
        iface = json.load(f)
    assert list(m.keys()) == iface["metric_keys"]
    assert all(isinstance(v, float) for v in m.values())
    ag.use_tb = False
    assert run_hip(ag, cfg, 1)[0] == {}


def test_metrics_as_device_tensors_option():
    """with metrics_on_device the update returns 0-d device tensors (no host wait in update());
    the reference's Logger.log takes tensors (logger.py:143-144: `value.item()`).  Same values as the float form."""
    cfg = CASES["small_h64_b6"]
    a, b = make_agent(cfg), make_agent(cfg)
    b.metrics_on_device = True
    for u in range(2):
        ma, _, _ = run_hip(a, cfg, u)
        mb, _, _ = run_hip(b, cfg, u)
        assert list(ma.keys()) == list(mb.keys())
        assert all(torch.is_tensor(v) and v.dim() == 0 for v in mb.values())
        assert all(v.is_cuda for k, v in mb.items() if k != "actor_ent")
        for k in ma:
            assert float(mb[k].item()) == pytest.approx(ma[k], rel=1e-6, abs=1e-7), (u, k)
    assert torch.equal(a._engine.params, b._engine.params)


def test_rng_draw_order_matches_reference():
    """Without the hook, update() consumes the global generator like the reference."""
    cfg = CASES["small_h64_b6"]
    with open(os.path.join(G, "interface.json")) as f:
        trace_ref = json.load(f)["rng_trace"]
    ag = make_agent(cfg)
    ag._engine.fused_rng = False          # the path that issues torch's own four calls (the fused launch: next test)
    calls = []
    import drqv2
    real_randint, real_sn = torch.randint, drqv2._standard_normal

    def t_randint(*a, **k):
        calls.append(["randint", list(k.get("size", ()))])
        return real_randint(*a, **k)

    def t_sn(shape, dtype, device):
        calls.append(["standard_normal", list(shape)])
        return real_sn(shape, dtype=dtype, device=device)

    torch.randint, drqv2._standard_normal = t_randint, t_sn
    try:
        batch = synth.make_batch(4, 3, 9, seed=0)
        ag.update(iter([tuple(x.numpy() for x in batch)]), 0)
    finally:
        torch.randint, drqv2._standard_normal = real_randint, real_sn
    want = [[k, [4 if d == 4 else d for d in s]] for k, s in trace_ref]
    got = [[k, s] for k, s in calls]
    assert [c[0] for c in got] == [c[0] for c in want]
    assert got[0][1] == [4, 1, 1, 2] and got[2][1] == [4, 3]
    # same seed -> same shifts as the reference call on the same device
    torch.manual_seed(5)
    a = ag.aug.draw(16, "cuda")
    torch.manual_seed(5)
    b = torch.randint(0, 9, size=(16, 1, 1, 2), device="cuda", dtype=torch.float32)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n,A", [(4, 3), (32, 1), (256, 6), (512, 12), (256, 21), (2048, 21), (100, 7)])
def test_fused_rng_launch_equals_the_four_torch_calls(n, A):
    """drq_rng_draws (one launch for the draws it can reproduce: all four, or the two integer shift draws when this
    torch build rounds logf / sincosf differently from the local device libraries) against the reference's own calls
    on the same generator state: the same shifts and noises bit for bit, the same generator offset afterwards, the
    same next random number -- a run with the fused launch consumes torch's RNG stream exactly like the reference
    (SURVEY App. C)."""
    import drqv2
    from torch.distributions.utils import _standard_normal
    ag = make_agent(CASES["small_h64_b6"])
    eng = ag._engine
    gen = torch.cuda.default_generators[eng.device.index]
    torch.manual_seed(1234 + n + A)
    torch.rand(3, device="cuda")                       # a generator that is not at offset 0
    st = gen.get_state()
    ref = [torch.randint(0, 9, size=(n, 1, 1, 2), device="cuda", dtype=torch.float32) for _ in range(2)]
    ref += [_standard_normal((n, A), dtype=torch.float32, device="cuda") for _ in range(2)]
    after_ref = torch.rand(5, device="cuda")
    gen.set_state(st)
    got = eng.rng_draws(n, A, 9)
    assert got is not None and eng._rng_ok in ("all", "shifts")
    got = [t.clone() for t in got]
    after_got = torch.rand(5, device="cuda")
    for a, b in zip(ref, got):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(after_ref, after_got)
    assert float(ref[0].min()) >= 0 and float(ref[0].max()) <= 8


def test_update_is_identical_with_and_without_the_fused_rng_launch():
    cfg = CASES["small_h64_b6"]
    outs = []
    for fused in (True, False):
        ag = make_agent(cfg)
        ag._engine.fused_rng = fused
        torch.manual_seed(77)
        ms = []
        for u in range(3):
            batch = synth.make_batch(cfg["B"], cfg["A"], cfg["C"], seed=u)
            ms.append(ag.update(iter([tuple(x.numpy() for x in batch)]), 2 * u))
        outs.append((ms, ag._engine.params.clone(), torch.rand(4, device="cuda")))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_act_matches_oracle():
    cfg = CASES["cheetah_b8"]
    ag = make_agent(cfg)
    o64 = make_oracle(cfg, torch.float64)
    obs = synth.make_batch(2, cfg["A"], 9, seed=5)[0]
    a = ag.act(obs[0].numpy(), 5000, True)
    assert a.dtype == np.float32 and a.shape == (cfg["A"],)
    ref = o64.act_mean(obs[0]).numpy()
    assert np.abs(a - ref).max() <= 2e-6
    torch.manual_seed(3)
    a1 = ag.act(obs[1].numpy(), 5000, False)
    assert np.abs(a1).max() <= 1.0 and not np.allclose(a1, ag.act(obs[1].numpy(), 5000, True))
    a2 = ag.act(obs[1].numpy(), 10, False)          # uniform exploration branch
    assert a2.shape == (cfg["A"],) and np.abs(a2).max() <= 1.0


def test_module_forwards_and_soft_update():
    import utils
    cfg = CASES["small_h64_b6"]
    ag = make_agent(cfg)
    o64 = make_oracle(cfg, torch.float64)
    from oracle import drq_oracle as O
    obs = synth.make_batch(3, cfg["A"], 9, seed=6)[0]
    feat = ag.encoder(obs.cuda())
    assert nerr(feat, O.encoder_forward(o64.enc, obs.double())) <= 2e-6
    dist = ag.actor(feat, 0.3)
    assert nerr(dist.mean, O.actor_mu(o64.actor, feat.double().cpu())) <= 5e-6
    act = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (3, cfg["A"])).astype(np.float32))
    q1, q2 = ag.critic(feat, act.cuda())
    r1, r2 = O.critic_q(o64.critic, feat.double().cpu(), act.double())
    assert nerr(q1, r1) <= 1e-5 and nerr(q2, r2) <= 1e-5
    # utils.soft_update_params on arena-backed modules == per-tensor formula
    with torch.no_grad():
        for p in ag.critic.parameters():
            p.add_(0.01)
    want = [O_t.clone() for O_t in (t.detach().cpu() for t in ag.critic_target.parameters())]
    for w, p in zip(want, ag.critic.parameters()):
        O.polyak(p.detach().cpu(), w, 0.05)
    utils.soft_update_params(ag.critic, ag.critic_target, 0.05)
    for w, t in zip(want, ag.critic_target.parameters()):
        assert torch.equal(t.detach().cpu(), w)


def test_pickle_roundtrip_on_gpu():
    import io
    cfg = CASES["small_h64_b6"]
    ag = make_agent(cfg)
    run_hip(ag, cfg, 0)
    buf = io.BytesIO()
    torch.save({"agent": ag}, buf)
    buf.seek(0)
    ag2 = torch.load(buf, weights_only=False)["agent"]
    for a, b in zip(ag.critic.parameters(), ag2.critic.parameters()):
        assert torch.equal(a, b)
    assert ag2.critic_opt.t == 1 and torch.equal(ag2._engine.adam_m, ag._engine.adam_m)
    m1, _, _ = run_hip(ag, cfg, 1)
    m2, _, _ = run_hip(ag2, cfg, 1)
    assert m1 == m2          # bit-stable and state fully restored


def test_run_to_run_bit_stable():
    cfg = CASES["small_h64_b6"]
    a, b = make_agent(cfg), make_agent(cfg)
    for u in range(2):
        ma, _, _ = run_hip(a, cfg, u)
        mb, _, _ = run_hip(b, cfg, u)
        assert ma == mb
    assert torch.equal(a._engine.params, b._engine.params)


def test_overlapped_data_parallel_schedule_equals_single_pass():
    """The data-parallel schedule (phases 3,4,5,1 with the all-reduces between them and Adam(actor) deferred
    into the next update) on a ONE-rank RCCL group must be bit-identical to the single drq_update_phase(-1)
    call: every SUM over one rank is the identity, so only the phase split, the stream hand-offs to RCCL's
    stream and the deferred step are exercised.  (N>1 ranks need N GPUs: the driver's scaling run.)"""
    import socket
    import torch.distributed as dist
    cfg = CASES["small_h64_b6"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        a, b, g = make_agent(cfg), make_agent(cfg), make_agent(cfg)
        b.enable_data_parallel(batch_is_global=False)
        g.enable_data_parallel(batch_is_global=False, global_metrics=True)     # + the metric-sums exchange
        assert b._engine.pg is not None and b._engine.world == 1
        for u in range(3):
            ma, _, _ = run_hip(a, cfg, u)
            mb, _, _ = run_hip(b, cfg, u)
            mg, _, _ = run_hip(g, cfg, u)
            assert ma == mb and ma == mg, u
            assert b._engine._pending is not None            # Adam(actor) of this update is still deferred
        g.flush()
        torch.cuda.synchronize()
        assert torch.equal(a._engine.params, g._engine.params)
        # act() must see the stepped actor: it flushes the deferred step first
        obs = synth.make_batch(1, cfg["A"], cfg["C"], seed=5)[0][0].numpy()
        xa = a.act(obs, 10 ** 6, True)
        xb = b.act(obs, 10 ** 6, True)
        assert b._engine._pending is None
        assert np.array_equal(xa, xb)
        torch.cuda.synchronize()
        assert torch.equal(a._engine.params, b._engine.params)
        assert torch.equal(a._engine.adam_m, b._engine.adam_m) and torch.equal(a._engine.adam_v, b._engine.adam_v)
    finally:
        dist.destroy_process_group()


def test_soak_300_updates_finite_and_reproducible():
    """300 consecutive updates on changing batches (device replay feeding the step): every metric stays finite,
    the parameters stay finite, and a second run from the same seeds ends bit-identical (no atomics, no
    order-dependent reductions anywhere on the path)."""
    import drqv2
    from drqv2_amd.replay import DeviceReplay

    def run():
        torch.manual_seed(11)
        ag = drqv2.DrQV2Agent((9, 84, 84), (6,), "cuda", 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,1000)", 0.3, True)
        rp = DeviceReplay(600, (9, 84, 84), 6, 3, 0.99, "cuda", seed=5)
        r = np.random.RandomState(7)
        for e in range(4):
            T1 = 101
            rp.add_episode({"observation": r.randint(0, 256, (T1, 9, 84, 84)).astype(np.uint8),
                            "action": r.uniform(-1, 1, (T1, 6)).astype(np.float32),
                            "reward": r.rand(T1, 1).astype(np.float32), "discount": np.ones((T1, 1), np.float32)})
        rp.batch_size = 32
        it = iter(rp)
        last = None
        for u in range(300):
            last = ag.update(it, 2 * u)
            assert all(np.isfinite(v) for v in last.values()), (u, last)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(ag._engine.params).all())
        return last, ag._engine.params.clone(), ag._engine.adam_v.clone()

    m1, p1, v1 = run()
    m2, p2, v2 = run()
    assert m1 == m2 and torch.equal(p1, p2) and torch.equal(v1, v2)
    assert m1["critic_loss"] >= 0.0 and abs(m1["critic_q1"]) < 1e3


def _dp2_worker(rank, world, port, cfg, ret, backend="gloo", exchange="allreduce"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                                             # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)                                      # gloo moves CUDA tensors through the host:
        dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks can share the one GPU of the box
    try:
        ag = make_agent(cfg)
        ag.enable_data_parallel(batch_is_global=True, global_metrics=True, exchange=exchange)
        ref = make_agent(cfg)                                         # same weights, single-process full batch
        out = {}
        for u in range(2):
            m_dp, _, _ = run_hip(ag, cfg, u)
            m_1, _, _ = run_hip(ref, cfg, u)
            ag.flush()
            torch.cuda.synchronize()
            g_dp, g_1 = ag._engine.grads, ref._engine.grads
            lay = ag._engine.layout["seg"]
            def summed(n):      # a bucket that took the sharded step ("auto" may pick it) keeps this rank's share in the
                b, e = lay[n]   # arena: the sum only ever existed inside the exchange, so it is formed here
                g = g_dp[b:e]
                if ag._engine.exchange is not None and ag._engine.exchange.sharded(e - b):
                    g = g.clone()
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                return g
            errs = {n: nerr(summed(n), g_1[lay[n][0]:lay[n][1]]) for n in ("enc", "critic", "actor")}
            out[u] = (m_dp, m_1, errs)
        # parameters: Adam's first steps are sign-like, so compare through the update direction's agreement
        d_dp = ag._engine.params - make_agent(cfg)._engine.params
        d_1 = ref._engine.params - make_agent(cfg)._engine.params
        out["cos"] = float(torch.nn.functional.cosine_similarity(d_dp, d_1, dim=0))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _direct_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from drqv2_amd.engine import GradExchange
        n = 512 * 64
        g = torch.Generator().manual_seed(7 + rank)
        x = torch.randn(n, generator=g).cuda()
        want = x.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        ex = GradExchange(dist.group.WORLD, world, torch.device("cuda", 0), "direct")
        y = x.clone()
        try:
            ex.start(y).wait()
        except RuntimeError as e:                   # gloo has no all-to-all for device tensors on this build
            ret[rank] = "unsupported: " + str(e)[:80]
            return
        torch.cuda.synchronize()
        ret[rank] = "ok" if torch.equal(y, want) else "mismatch"
    finally:
        dist.destroy_process_group()


def test_direct_exchange_on_device_tensors_side_stream():
    """GradExchange 'direct' on GPU tensors: the all-to-all / rank-order sum / all-gather sequence runs on a side
    stream and the handle's wait() orders the current stream behind it.  Two ranks share the box's GPU through
    gloo; where gloo cannot move device tensors through an all-to-all the test says so and skips (the RCCL run on a
    multi-GPU node is the driver's)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_direct_worker, args=(2, port, ret), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    if any(str(v).startswith("unsupported") for v in ret.values()):
        pytest.skip(str(ret[0]))
    assert dict(ret) == {0: "ok", 1: "ok"}


def _run_two_ranks(backend, exchange="allreduce"):
    import socket
    import torch.multiprocessing as mp
    cfg = dict(CASES["small_h64_b6"]); cfg["B"] = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp2_worker, args=(2, port, cfg, ret, backend, exchange), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for rank in (0, 1):
        out = ret[rank]
        for u in (0, 1):
            m_dp, m_1, errs = out[u]
            for k in m_1:
                assert m_dp[k] == pytest.approx(m_1[k], rel=1e-5, abs=1e-6), (rank, u, k)
            tol = 2e-5 if u == 0 else 5e-3       # update 2 inherits Adam's sign-like first step (SURVEY finding 3)
            assert max(errs.values()) <= tol, (rank, u, errs)
        assert out["cos"] > 0.999


def test_two_rank_data_parallel_update_equals_full_batch_on_gpu():
    """World size 2 on the real kernels: each rank takes half of the same global batch (global shifts/noise sliced),
    gradients are SUM-all-reduced (gloo carries the CUDA tensors, so both ranks can sit on the box's single GPU)
    through the overlapped schedule, and the result must equal the one-process full-batch update: gradients to
    summation-order rounding, metrics to 1e-5."""
    _run_two_ranks("gloo")


@pytest.mark.parametrize("exchange", ["direct", "auto"])
def test_two_rank_update_with_direct_exchange(exchange):
    """The same equivalence with the gradient buckets summed by GradExchange's all-to-all / rank-order sum /
    all-gather on a side stream ("direct"), and with the mode picked by measurement ("auto": what bench.py uses):
    the whole overlapped schedule (deferred Adam steps waiting on side-stream handles) on the real kernels."""
    _run_two_ranks("gloo", exchange)


def _zero1_gpu_worker(rank, world, port, cfg, ret, backend):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rep, zer = make_agent(cfg), make_agent(cfg)
        try:
            rep.enable_data_parallel(batch_is_global=True, exchange="direct")
            zer.enable_data_parallel(batch_is_global=True, exchange="zero1")
            ms = []
            for u in range(3):
                m_r, _, _ = run_hip(rep, cfg, u)
                m_z, _, _ = run_hip(zer, cfg, u)
                ms.append((m_r, m_z))
            rep.flush(); zer.flush()
            torch.cuda.synchronize()
        except RuntimeError as e:                    # a gloo build without all-to-all for device tensors
            ret[rank] = "unsupported: " + str(e)[:80]
            return
        er, ez = rep._engine, zer._engine
        ok = torch.equal(er.params, ez.params)                          # weights AND the Polyak target, every rank
        lay = er.layout["seg"]
        own_ok = True
        for net in ("enc", "critic", "actor"):
            b, e = lay[net]
            ns = (e - b) // world
            sl = slice(b + rank * ns, b + (rank + 1) * ns)
            own_ok &= torch.equal(er.adam_m[sl], ez.adam_m[sl]) and torch.equal(er.adam_v[sl], ez.adam_v[sl])
        ez.gather_optimizer_state()                                     # what a snapshot does first
        torch.cuda.synchronize()
        full_ok = all(torch.equal(er.adam_m[lay[n][0]:lay[n][1]], ez.adam_m[lay[n][0]:lay[n][1]]) and
                      torch.equal(er.adam_v[lay[n][0]:lay[n][1]], ez.adam_v[lay[n][0]:lay[n][1]])
                      for n in ("enc", "critic", "actor"))
        same_metrics = all(a == b for a, b in ms)
        ret[rank] = "ok" if (ok and own_ok and full_ok and same_metrics) else f"mismatch {ok} {own_ok} {full_ok} {same_metrics}"
    finally:
        dist.destroy_process_group()


def _run_zero1(backend):
    import socket
    import torch.multiprocessing as mp
    cfg = dict(CASES["small_h64_b6"]); cfg["B"] = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_zero1_gpu_worker, args=(2, port, cfg, ret, backend), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    if any(str(v).startswith("unsupported") for v in ret.values()):
        pytest.skip(str(ret[0]))
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_two_rank_zero1_equals_replicated_adam_bit_for_bit():
    """ZeRO-1 (exchange="zero1": gradient slices by all-to-all, rank-order sum + Adam on the owned slice in one kernel,
    stepped parameters by all-gather, Polyak on the gathered critic) against the replicated path with the same
    rank-order sum ("direct"): after three updates of the overlapped schedule every parameter and the target are
    bit-identical on every rank, each rank's own slice of the Adam moments too, and gather_optimizer_state() makes the
    moments complete.  Two ranks share the box's GPU through gloo."""
    _run_zero1("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_rank_rccl_zero1_equals_replicated_adam():
    _run_zero1("nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("exchange", ["allreduce", "direct"])
def test_two_rank_rccl_update_equals_full_batch(exchange):
    """The same equivalence with one GPU per rank and the gradient exchanges over RCCL (backend nccl): the path
    bench.py --gpus N takes.  Runs wherever the box shows two or more GPUs."""
    _run_two_ranks("nccl", exchange)


@pytest.mark.parametrize("B", [256, 37])
def test_production_store_setting_equals_the_test_setting(B):
    """Every parity test above asks conv1_aug_kernel to keep BOTH views' encoder input (store_aug_next=True) so that it
    can look at them; production keeps the obs view only (n_store = B).  The two settings must be the same update: bit
    for bit the same parameters, gradients and Adam moments after two updates, and the same metrics."""
    cfg = dict(C=9, A=6, F=50, H=1024, B=B, lr=1e-4, sched="linear(1.0,0.1,500000)", wseed=0, bseed=7, updates=2,
               step0=0, smooth=True)
    outs = []
    for store in (True, False):
        ag = make_agent(cfg)
        ag._engine.store_aug_next = store
        ms = [run_hip(ag, cfg, u)[0] for u in range(cfg["updates"])]
        torch.cuda.synchronize()
        eng = ag._engine
        outs.append((ms, eng.params.clone(), eng.grads.clone(), eng.adam_m.clone(), eng.adam_v.clone()))
    (m0, *a0), (m1, *a1) = outs
    assert m0 == m1
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)


@pytest.mark.parametrize("name", ["cheetah_b8", "cartpole_b32"])
def test_update_critic_update_actor_as_separate_calls_equal_update(name):
    """The reference's public pieces (drqv2.py:177-228): encode -> update_critic -> update_actor ->
    utils.soft_update_params, issued by the caller, are the same update as update(): same consumption of the global
    generator (two shift draws, two noise draws, in that order), bit-identical parameters, target, Adam moments and
    metrics.  (update() fuses these pieces into one call; the pieces cut phase 6 at the method boundaries.)"""
    import utils
    cfg = CASES[name]
    batches = [synth.make_batch(cfg["B"], cfg["A"], cfg["C"], seed=cfg["bseed"] + u, smooth=cfg["smooth"]) for u in range(2)]
    outs = []
    for manual in (False, True):
        ag = make_agent(cfg)
        ag._engine.fused_rng = False                 # torch's own four calls in both runs (the pieces draw one by one)
        torch.manual_seed(123)
        torch.cuda.manual_seed_all(123)
        ms = []
        for u, batch in enumerate(batches):
            step = cfg["step0"] + 2 * u
            if not manual:
                ms.append(ag.update(iter([tuple(x.numpy() for x in batch)]), step))
                continue
            obs, action, reward, discount, next_obs = utils.to_torch(tuple(x.numpy() for x in batch), ag.device)
            m = {"batch_reward": float(reward.float().mean())}
            f_obs, f_next = ag.encode(obs, next_obs, step)
            m.update(ag.update_critic(f_obs, action, reward, discount, f_next, step))
            m.update(ag.update_actor(f_obs.detach(), step))
            utils.soft_update_params(ag.critic, ag.critic_target, ag.critic_target_tau)
            ms.append(m)
        torch.cuda.synchronize()
        eng = ag._engine
        outs.append((ms, eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), torch.cuda.get_rng_state()))
    (m0, p0, am0, av0, r0), (m1, p1, am1, av1, r1) = outs
    assert torch.equal(r0, r1)                       # the generator moved by the same amount
    assert torch.equal(p0, p1) and torch.equal(am0, am1) and torch.equal(av0, av1)
    for a, b in zip(m0, m1):
        assert set(a) == set(b)
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-7), k     # batch_reward: two ways of taking a mean
    # out of order: loud errors, nothing silently skipped
    from drqv2_amd._lib import DrqError
    ag = make_agent(cfg)
    with pytest.raises(DrqError):
        ag.update_actor(torch.zeros(cfg["B"], 39200, device="cuda"), 0)
