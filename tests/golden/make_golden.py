#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

How the reference is loaded: /root/reference/drqv2.py and utils.py import two
packages that are not installed here and that neither file uses on this path
(`import hydra` drqv2.py:5, `from omegaconf import OmegaConf` utils.py:13).
They are satisfied with two EMPTY placeholder modules in sys.modules; every
line that executes is the reference's own.  Nothing from
the reference is written into the repo -- only inputs-by-seed and outputs.

Random draws are injected so the run is replayable without the torch RNG:
  * aug shifts: torch.randint is intercepted for the aug call signature
    (drqv2.py:34-38) and returns the fixture's integer shifts as float32;
  * action noise: utils._standard_normal (utils.py:119) returns fixture noise.
"""
import contextlib
import hashlib
import inspect
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from drqv2_amd import synth  # noqa: E402

REF = "/root/reference"


def load_reference():
    sys.modules.setdefault("hydra", types.ModuleType("hydra"))
    om = types.ModuleType("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", om)
    # the repo root also has drqv2.py / utils.py (the drop-in); load the
    # reference's files explicitly by path under private names.
    import importlib.util

    def load(name, alias):
        spec = importlib.util.spec_from_file_location(alias, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        return spec, mod

    spec_u, ref_utils = load("utils", "utils")
    saved = sys.modules.get("utils")
    sys.modules["utils"] = ref_utils           # drqv2.py does `import utils`
    spec_u.loader.exec_module(ref_utils)
    spec_d, ref_drq = load("drqv2", "ref_drqv2")
    spec_d.loader.exec_module(ref_drq)
    if saved is not None:
        sys.modules["utils"] = saved
    else:
        del sys.modules["utils"]
    spec_r, ref_rb = load("replay_buffer", "ref_replay_buffer")
    spec_r.loader.exec_module(ref_rb)
    return ref_drq, ref_utils, ref_rb


@contextlib.contextmanager
def injected_draws(ref_utils, shifts, noises):
    """shifts: list of int tensors [B,2]; noises: list of float tensors [B,A]."""
    shifts, noises = list(shifts), list(noises)
    real_randint = torch.randint
    real_sn = ref_utils._standard_normal

    def fake_randint(*a, **k):
        size = k.get("size")
        if size is not None and len(size) == 4 and size[1:] == (1, 1, 2) and shifts:
            s = shifts.pop(0)
            return s.to(k.get("dtype", torch.float32)).view(*size)
        return real_randint(*a, **k)

    def fake_sn(shape, dtype, device):
        n = noises.pop(0)
        assert tuple(n.shape) == tuple(shape)
        return n.to(dtype).clone()

    torch.randint = fake_randint
    ref_utils._standard_normal = fake_sn
    try:
        yield
    finally:
        torch.randint = real_randint
        ref_utils._standard_normal = real_sn


def summary(t, k=8):
    """norm, sum and k evenly spaced elements (indices = linspace(0,numel-1,k).long())."""
    t = t.detach().double().reshape(-1)
    idx = torch.linspace(0, t.numel() - 1, min(k, t.numel())).long()
    return {"numel": int(t.numel()), "l2": float(t.norm()), "sum": float(t.sum()),
            "val": t[idx].tolist()}


def crop_aug(dtype):
    import oracle.drq_oracle as O

    class Crop:
        """stands in for agent.aug: exact integer crop with the SAME shifts the
        intercepted randint would have produced (consumes one shift tensor)."""

        def __call__(self, x):
            n = x.shape[0]
            sh = torch.randint(0, 9, size=(n, 1, 1, 2), dtype=torch.float32).view(n, 2).long()
            return O.aug_integer_crop(x.to(dtype), sh)
    return Crop()


def build_agent(ref_drq, C, A, Fd, H, lr, sched, seed, dtype):
    torch.manual_seed(1)
    ag = ref_drq.DrQV2Agent((C, 84, 84), (A,), "cpu", lr, Fd, H, 0.01, 2000, 2, sched, 0.3, True)
    enc, actor, critic = synth.make_weights(C, A, Fd, H, seed)
    ag.encoder.load_state_dict(enc)
    ag.actor.load_state_dict(actor)
    ag.critic.load_state_dict(critic)
    ag.critic_target.load_state_dict(critic)
    if dtype == torch.float64:
        for m in (ag.encoder, ag.actor, ag.critic, ag.critic_target):
            m.double()
    return ag


def run_steps(ref_drq, ref_utils, cfg, dtype, aug_mode, threads):
    torch.set_num_threads(threads)
    C, A, Fd, H, B = cfg["C"], cfg["A"], cfg["F"], cfg["H"], cfg["B"]
    ag = build_agent(ref_drq, C, A, Fd, H, cfg["lr"], cfg["sched"], cfg["wseed"], dtype)
    if aug_mode == "crop":
        ag.aug = crop_aug(dtype)
    out = []
    for u in range(cfg["updates"]):
        step = cfg["step0"] + 2 * u
        batch = synth.make_batch(B, A, C, seed=cfg["bseed"] + u, smooth=cfg["smooth"])
        if dtype == torch.float64:
            batch = tuple(t if t.dtype == torch.uint8 else t.double() for t in batch)
        sh_o, sh_n, n_c, n_a = synth.make_draws(B, A, seed=cfg["bseed"] + u)
        grabbed = {}
        real_step = ag.critic_opt.step

        def grab_then_step(*a, **k):
            grabbed["critic"] = [p.grad.detach().clone() for p in ag.critic.parameters()]
            grabbed["enc"] = [p.grad.detach().clone() for p in ag.encoder.parameters()]
            return real_step(*a, **k)

        ag.critic_opt.step = grab_then_step
        with injected_draws(ref_utils, [sh_o, sh_n], [n_c, n_a]):
            metrics = ag.update(iter([tuple(x.numpy() for x in batch)]), step)
        ag.critic_opt.step = real_step
        rec = {"step": step, "metrics": metrics}
        rec["g_enc"] = [summary(g) for g in grabbed["enc"]]
        rec["g_critic"] = [summary(g) for g in grabbed["critic"]]
        rec["g_actor"] = [summary(p.grad) for p in ag.actor.parameters()]
        rec["p_enc"] = [summary(p) for p in ag.encoder.parameters()]
        rec["p_critic"] = [summary(p) for p in ag.critic.parameters()]
        rec["p_actor"] = [summary(p) for p in ag.actor.parameters()]
        rec["p_target"] = [summary(p) for p in ag.critic_target.parameters()]
        out.append(rec)
    return out


def gen_steps(ref_drq, ref_utils):
    cfgs = {
        "cheetah_b8": dict(C=9, A=6, F=50, H=1024, B=8, lr=1e-4, sched="linear(1.0,0.1,500000)",
                           wseed=0, bseed=0, updates=3, step0=0, smooth=True),
        "humanoid_b4": dict(C=9, A=21, F=100, H=1024, B=4, lr=8e-5, sched="linear(1.0,0.1,2000000)",
                            wseed=1, bseed=10, updates=2, step0=1000, smooth=True),
        "cartpole_b32": dict(C=9, A=1, F=50, H=1024, B=32, lr=1e-4, sched="linear(1.0,0.1,100000)",
                             wseed=2, bseed=20, updates=2, step0=50000, smooth=False),
        "small_h64_b6": dict(C=9, A=3, F=20, H=64, B=6, lr=1e-3, sched="0.2",
                             wseed=3, bseed=30, updates=3, step0=0, smooth=True),
    }
    res = {}
    for name, cfg in cfgs.items():
        res[name] = {"cfg": cfg,
                     "ref_fp32_crop": run_steps(ref_drq, ref_utils, cfg, torch.float32, "crop", 8),
                     "ref_fp64_crop": run_steps(ref_drq, ref_utils, cfg, torch.float64, "crop", 8),
                     "ref_fp32_aug": run_steps(ref_drq, ref_utils, cfg, torch.float32, "real", 8)}
        print("steps", name, res[name]["ref_fp32_crop"][0]["metrics"])
    import gzip
    with gzip.open(os.path.join(HERE, "steps.json.gz"), "wt") as f:
        json.dump(res, f)


def gen_aug(ref_drq, ref_utils):
    obs_s = synth.make_batch(4, 1, 9, seed=3, smooth=True)[0]
    obs_n = synth.make_batch(4, 1, 9, seed=4, smooth=False)[0]
    shifts = torch.tensor([[0, 0], [8, 8], [4, 4], [0, 8]], dtype=torch.int32)
    shifts2 = torch.tensor([[8, 0], [3, 5], [7, 1], [1, 6]], dtype=torch.int32)
    aug = ref_drq.RandomShiftsAug(pad=4)
    rec = {}
    for nm, obs, sh in (("smooth", obs_s, shifts), ("noise", obs_n, shifts2)):
        with injected_draws(ref_utils, [sh], []):
            out = aug(obs.float())
        sub = out[:, ::4, ::5, ::3].contiguous()
        rec[nm] = {"shifts": sh.tolist(), "sub": sub.numpy().astype(np.float32),
                   "sha256_u8": hashlib.sha256(obs.numpy().tobytes()).hexdigest()}
    base = torch.linspace(-1.0 + 1.0 / 92, 1.0 - 1.0 / 92, 92, dtype=torch.float32)[:84]
    # G2: seed -> shifts through the reference call signature
    seed_shift = {}
    for s in (0, 1, 12345):
        torch.manual_seed(s)
        sh = torch.randint(0, 9, size=(256, 1, 1, 2), dtype=torch.float32)
        seed_shift[str(s)] = sh.view(256, 2).to(torch.int32).numpy()
    np.savez_compressed(os.path.join(HERE, "aug.npz"),
                        smooth_sub=rec["smooth"]["sub"], noise_sub=rec["noise"]["sub"],
                        smooth_shifts=np.array(rec["smooth"]["shifts"], dtype=np.int32),
                        noise_shifts=np.array(rec["noise"]["shifts"], dtype=np.int32),
                        base_grid=base.numpy(),
                        **{f"seed_{k}": v for k, v in seed_shift.items()})
    print("aug ok", rec["smooth"]["sub"].shape)


def gen_elementwise(ref_drq, ref_utils):
    """G6: Adam (torch.optim.Adam, CPU) and Polyak on small tensors, t=1..3."""
    rs = np.random.RandomState(11)
    n = 1031
    p0 = rs.standard_normal(n).astype(np.float32)
    gs = [(rs.standard_normal(n) * 10.0 ** rs.uniform(-9, 1, n)).astype(np.float32) for _ in range(3)]
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p], lr=1e-4)
    ps, ms, vs = [], [], []
    for g in gs:
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        st = opt.state[p]
        ps.append(p.detach().numpy().copy())
        ms.append(st["exp_avg"].numpy().copy())
        vs.append(st["exp_avg_sq"].numpy().copy())
    net = torch.nn.Linear(64, 65)
    tgt = torch.nn.Linear(64, 65)
    with torch.no_grad():
        net.weight.copy_(torch.from_numpy(rs.standard_normal((65, 64)).astype(np.float32)))
        tgt.weight.copy_(torch.from_numpy(rs.standard_normal((65, 64)).astype(np.float32)))
    w_net, w_tgt0 = net.weight.detach().numpy().copy(), tgt.weight.detach().numpy().copy()
    ref_utils.soft_update_params(net, tgt, 0.01)
    # TruncatedNormal sample / log_prob / entropy
    mu = torch.from_numpy(np.tanh(rs.standard_normal((16, 6)) * 2).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((16, 6)).astype(np.float32))
    std = 0.37
    dist = ref_utils.TruncatedNormal(mu, torch.ones_like(mu) * std)
    with injected_draws(ref_utils, [], [noise]):
        a = dist.sample(clip=0.3)
    lp = dist.log_prob(a).sum(-1, keepdim=True)
    ent = dist.entropy().sum(-1)
    sched = {s: [ref_utils.schedule(s, st) for st in (0, 1, 999, 50000, 100000, 3000000)]
             for s in ("linear(1.0,0.1,100000)", "linear(1.0,0.1,500000)", "linear(1.0,0.1,2000000)",
                       "0.2", "step_linear(1.0,0.5,1000,0.1,50000)")}
    np.savez_compressed(os.path.join(HERE, "elementwise.npz"),
                        adam_p0=p0, adam_g=np.stack(gs), adam_p=np.stack(ps), adam_m=np.stack(ms),
                        adam_v=np.stack(vs), ema_net=w_net, ema_tgt0=w_tgt0,
                        ema_tgt1=tgt.weight.detach().numpy(), tn_mu=mu.numpy(), tn_noise=noise.numpy(),
                        tn_a=a.numpy(), tn_logp=lp.numpy(), tn_ent=ent.numpy())
    with open(os.path.join(HERE, "schedule.json"), "w") as f:
        json.dump(sched, f)
    print("elementwise ok")


def gen_interface(ref_drq, ref_utils, ref_rb):
    torch.manual_seed(1)
    ag = ref_drq.DrQV2Agent((9, 84, 84), (6,), "cpu", 1e-4, 50, 1024, 0.01, 2000, 2,
                            "linear(1.0,0.1,500000)", 0.3, True)

    def sig(f):
        return [[n, None if p.default is inspect._empty else repr(p.default)]
                for n, p in inspect.signature(f).parameters.items()]

    iface = {
        "signatures": {
            "DrQV2Agent.__init__": sig(ref_drq.DrQV2Agent.__init__),
            "DrQV2Agent.act": sig(ref_drq.DrQV2Agent.act),
            "DrQV2Agent.update": sig(ref_drq.DrQV2Agent.update),
            "DrQV2Agent.update_critic": sig(ref_drq.DrQV2Agent.update_critic),
            "DrQV2Agent.update_actor": sig(ref_drq.DrQV2Agent.update_actor),
            "DrQV2Agent.train": sig(ref_drq.DrQV2Agent.train),
            "RandomShiftsAug.__init__": sig(ref_drq.RandomShiftsAug.__init__),
            "Encoder.__init__": sig(ref_drq.Encoder.__init__),
            "Actor.__init__": sig(ref_drq.Actor.__init__),
            "Actor.forward": sig(ref_drq.Actor.forward),
            "Critic.__init__": sig(ref_drq.Critic.__init__),
            "Critic.forward": sig(ref_drq.Critic.forward),
            "utils.soft_update_params": sig(ref_utils.soft_update_params),
            "utils.to_torch": sig(ref_utils.to_torch),
            "utils.schedule": sig(ref_utils.schedule),
            "utils.weight_init": sig(ref_utils.weight_init),
            "utils.TruncatedNormal.__init__": sig(ref_utils.TruncatedNormal.__init__),
            "utils.TruncatedNormal.sample": sig(ref_utils.TruncatedNormal.sample),
            "utils.set_seed_everywhere": sig(ref_utils.set_seed_everywhere),
            "utils.Until.__init__": sig(ref_utils.Until.__init__),
            "utils.Every.__init__": sig(ref_utils.Every.__init__),
        },
        "utils_names": sorted(n for n in dir(ref_utils) if not n.startswith("_") and
                              getattr(getattr(ref_utils, n), "__module__", None) == ref_utils.__name__),
        "state_dict": {nm: [[k, list(v.shape)] for k, v in getattr(ag, nm).state_dict().items()]
                       for nm in ("encoder", "actor", "critic", "critic_target")},
        "agent_attrs": sorted(k for k in vars(ag)),
        "repr_dim": ag.encoder.repr_dim,
    }
    # constructor RNG parity: weights produced under manual_seed(1) (summaries only)
    iface["init_seed1"] = {nm: [summary(p, 4) for p in getattr(ag, nm).parameters()]
                           for nm in ("encoder", "actor", "critic")}
    # metric keys + RNG call order of one update
    trace = []
    real_randint, real_sn = torch.randint, ref_utils._standard_normal

    def t_randint(*a, **k):
        trace.append(["randint", list(k.get("size", ()))])
        return real_randint(*a, **k)

    def t_sn(shape, dtype, device):
        trace.append(["standard_normal", list(shape)])
        return real_sn(shape, dtype=dtype, device=device)

    torch.randint, ref_utils._standard_normal = t_randint, t_sn
    batch = synth.make_batch(4, 6, 9, seed=0)
    m = ag.update(iter([tuple(x.numpy() for x in batch)]), 0)
    m_gated = ag.update(iter([]), 1)
    torch.randint, ref_utils._standard_normal = real_randint, real_sn
    iface["metric_keys"] = list(m.keys())
    iface["gated_metrics"] = m_gated
    iface["rng_trace"] = trace
    # act(): shapes/dtype
    with torch.no_grad():
        a = ag.act(batch[0][0].numpy(), 5000, True)
    iface["act"] = {"dtype": str(a.dtype), "shape": list(a.shape)}
    with open(os.path.join(HERE, "interface.json"), "w") as f:
        json.dump(iface, f, indent=1)
    print("interface ok", iface["metric_keys"])


def gen_cfg_surface():
    import yaml
    cdir = os.path.join(REF, "cfgs")
    with open(os.path.join(cdir, "config.yaml")) as f:
        base = yaml.safe_load(f)
    base.pop("hydra", None)
    base.pop("defaults", None)
    tasks = {}
    for fn in sorted(os.listdir(os.path.join(cdir, "task"))):
        with open(os.path.join(cdir, "task", fn)) as f:
            tasks[fn[:-5]] = yaml.safe_load(f)

    def resolve(name):
        t = dict(tasks[name])
        out = {}
        for d in t.pop("defaults", []):
            if isinstance(d, str) and d != "_self_":
                out.update(resolve(d))
        out.update(t)
        return out

    surface = {"base": base, "tasks": {}}
    for name in tasks:
        if name in ("easy", "medium", "hard"):
            continue
        cfg = dict(base)
        cfg.update(resolve(name))
        surface["tasks"][name] = cfg
    with open(os.path.join(HERE, "cfg_surface.json"), "w") as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    print("cfg ok", len(surface["tasks"]))


def gen_nstep(ref_rb):
    """G8: n-step reward/discount of replay_buffer.py:142-160 on synthetic episodes."""
    rs = np.random.RandomState(3)
    out = []
    for nstep, gamma in ((3, 0.99), (1, 0.99), (5, 0.9)):
        T = 12
        ep = {"observation": np.zeros((T + 1, 1), np.uint8), "action": np.zeros((T + 1, 1), np.float32),
              "reward": rs.standard_normal((T + 1, 1)).astype(np.float32),
              "discount": np.where(rs.uniform(size=(T + 1, 1)) < 0.1, 0.0, 1.0).astype(np.float32)}
        for idx in range(1, T - nstep + 2):
            reward = np.zeros_like(ep["reward"][idx])
            discount = np.ones_like(ep["discount"][idx])
            # drive the reference's own loop by fixing its two random choices
            rb = ref_rb.ReplayBuffer.__new__(ref_rb.ReplayBuffer)
            rb._nstep, rb._discount = nstep, gamma
            rb._samples_since_last_fetch = 0
            rb._try_fetch = lambda: None
            rb._sample_episode = lambda ep=ep: ep
            real = np.random.randint
            np.random.randint = lambda lo, hi=None, idx=idx: idx - 1
            try:
                o, a, r, d, no = rb._sample()
            finally:
                np.random.randint = real
            out.append({"nstep": nstep, "gamma": gamma, "idx": idx, "reward_in": ep["reward"][:, 0].tolist(),
                        "discount_in": ep["discount"][:, 0].tolist(), "reward": float(r[0]), "discount": float(d[0])})
    with open(os.path.join(HERE, "nstep.json"), "w") as f:
        json.dump(out, f)
    print("nstep ok", len(out))


def gen_ref_state(ref_drq, ref_utils):
    """A small REFERENCE agent after two of its own updates, saved as plain tensors (module state_dicts and
    torch.optim.Adam state_dicts; loads with weights_only=True): what tests/test_cpu_interface.py imports into this
    repo's agent.  The reverse direction is checked right here, in the only place where the reference runs: the
    export of this repo's agent is loaded by the reference's own objects with their own load_state_dict."""
    C, A, Fd, H, B = 9, 2, 2, 8, 2
    torch.manual_seed(3)
    ag = ref_drq.DrQV2Agent((C, 84, 84), (A,), "cpu", 1e-3, Fd, H, 0.01, 2000, 2, "0.2", 0.3, True)
    for u in range(2):
        batch = synth.make_batch(B, A, C, seed=200 + u, smooth=True)
        ag.update(iter([tuple(x.numpy() for x in batch)]), 2 * u)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = {"format": "drqv2-reference-state-v1", "dims": {"C": C, "A": A, "F": Fd, "H": H},
           "encoder": sd(ag.encoder), "actor": sd(ag.actor), "critic": sd(ag.critic),
           "critic_target": sd(ag.critic_target),
           "encoder_opt": ag.encoder_opt.state_dict(), "actor_opt": ag.actor_opt.state_dict(),
           "critic_opt": ag.critic_opt.state_dict()}
    torch.save(out, os.path.join(HERE, "ref_state.pt"))
    # reverse direction: this repo's agent -> export -> the reference's objects
    import importlib
    sys.modules.pop("utils", None)
    mine_mod = importlib.import_module("drqv2")            # the drop-in at the repo root (with its own utils)
    mine = mine_mod.DrQV2Agent((C, 84, 84), (A,), "cpu", 1e-3, Fd, H, 0.01, 2000, 2, "0.2", 0.3, True)
    mine.import_reference_state(torch.load(os.path.join(HERE, "ref_state.pt"), weights_only=True))
    exp = mine.export_reference_state()
    torch.manual_seed(4)
    other = ref_drq.DrQV2Agent((C, 84, 84), (A,), "cpu", 1e-3, Fd, H, 0.01, 2000, 2, "0.2", 0.3, True)
    for n in ("encoder", "actor", "critic", "critic_target"):
        getattr(other, n).load_state_dict(exp[n])
    for n in ("encoder_opt", "actor_opt", "critic_opt"):
        getattr(other, n).load_state_dict(exp[n])
    for n in ("encoder", "actor", "critic", "critic_target"):
        for (k, a), (_, b) in zip(getattr(ag, n).state_dict().items(), getattr(other, n).state_dict().items()):
            assert torch.equal(a, b), (n, k)
    for n in ("encoder_opt", "actor_opt", "critic_opt"):
        sa, sb = getattr(ag, n).state_dict()["state"], getattr(other, n).state_dict()["state"]
        assert sa.keys() == sb.keys()
        for i in sa:
            assert float(sa[i]["step"]) == float(sb[i]["step"])
            assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"])
    # and the reference keeps training from it
    batch = synth.make_batch(B, A, C, seed=202, smooth=True)
    m = other.update(iter([tuple(x.numpy() for x in batch)]), 4)
    assert all(np.isfinite(v) for v in m.values())
    print("ref_state.pt written; the reference's agent loaded this repo's export and took an update:", m["critic_loss"])


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    ref_drq, ref_utils, ref_rb = load_reference()
    which = sys.argv[1:] or ["aug", "elementwise", "interface", "cfg", "nstep", "steps", "ref_state"]
    if "aug" in which:
        gen_aug(ref_drq, ref_utils)
    if "elementwise" in which:
        gen_elementwise(ref_drq, ref_utils)
    if "interface" in which:
        gen_interface(ref_drq, ref_utils, ref_rb)
    if "cfg" in which:
        gen_cfg_surface()
    if "nstep" in which:
        gen_nstep(ref_rb)
    if "steps" in which:
        gen_steps(ref_drq, ref_utils)
    if "ref_state" in which:
        gen_ref_state(ref_drq, ref_utils)
