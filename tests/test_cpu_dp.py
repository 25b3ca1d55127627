"""Data-parallel sharding of the update step, world_size 2 on the gloo backend (CPU).

The reference has no distributed code; the sharding contract is new: the global
batch is split into equal contiguous shards, every rank draws the GLOBAL shifts/noise and keeps its slice,
local gradients are scaled by 1/global_B and SUM-all-reduced in two buckets of the flat gradient arena
(encoder+critic, then actor -- the actor step needs the already-updated critic).  Here the per-shard
gradients come from the CPU oracle (fp64), the buckets and slicing from the product host code
(drqv2_amd.engine), and the result must equal the full-batch oracle step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from drqv2_amd import synth

CFG = dict(C=9, A=3, F=20, H=64, B=8, lr=1e-3, sched="0.2")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_batch_reference():
    from oracle import drq_oracle as O
    enc, actor, critic = synth.make_weights(CFG["C"], CFG["A"], CFG["F"], CFG["H"], 3)
    batch = synth.make_batch(CFG["B"], CFG["A"], CFG["C"], seed=30)
    draws = synth.make_draws(CFG["B"], CFG["A"], seed=30)
    ag = O.OracleAgent(enc, actor, critic, CFG["lr"], stddev_schedule=CFG["sched"], dtype=torch.float64)
    m = ag.update(batch, 0, *draws, keep=True)
    return ag, m, (enc, actor, critic), batch, draws


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import drq_oracle as O
        from drqv2_amd import _lib
        from drqv2_amd.engine import grad_buckets, grad_buckets_overlap, shard_bounds
        full, m_full, (enc, actor, critic), batch, draws = _full_batch_reference()
        lay = _lib.param_layout(CFG["C"], CFG["A"], CFG["F"], CFG["H"])
        lo, hi, n_global = shard_bounds(CFG["B"], world, rank, True)
        assert n_global == CFG["B"] and hi - lo == CFG["B"] // world
        sh = lambda t: t[lo:hi]
        # phase 0 on the shard: critic loss grads w.r.t. encoder+critic, means over the LOCAL rows
        ag = O.OracleAgent(enc, actor, critic, CFG["lr"], stddev_schedule=CFG["sched"], dtype=torch.float64)
        shard = tuple(sh(t) for t in batch)
        sdraws = tuple(sh(t) for t in draws)
        ag.update(shard, 0, *sdraws, keep=True)
        local = ag.last
        grads = torch.zeros(lay["total"], dtype=torch.float64)
        scale = (hi - lo) / n_global               # the kernels scale by 1/global_B instead of 1/local_B

        def put(net, gd):
            for off, g in zip(lay[net], gd.values()):
                grads[off:off + g.numel()] = g.reshape(-1) * scale
        put("enc", local["g_enc"])
        put("critic", local["g_critic"])
        b1, b2 = grad_buckets(lay)
        assert b1 == (lay["seg"]["enc"][0], lay["seg"]["critic"][1]) and b1[1] <= b2[0]
        cb, eb, ab = grad_buckets_overlap(lay)      # the overlapped schedule splits bucket 1 at the enc/critic seam
        assert (eb[0], cb[1]) == b1 and eb[1] <= cb[0] and ab == b2
        dist.all_reduce(grads[b1[0]:b1[1]], op=dist.ReduceOp.SUM)
        for net, key in (("enc", "g_enc"), ("critic", "g_critic")):
            for off, g in zip(lay[net], full.last[key].values()):
                got = grads[off:off + g.numel()]
                assert torch.allclose(got, g.reshape(-1), rtol=1e-9, atol=1e-12), (net, rank)
        # metric partial sums reduce the same way
        s = torch.tensor([local["target_q"].sum(), local["q1"].sum()], dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        assert float(s[0]) / n_global == pytest.approx(m_full["critic_target_q"], rel=1e-12)
        assert float(s[1]) / n_global == pytest.approx(m_full["critic_q1"], rel=1e-12)
        # phase 1 needs the critic stepped with the REDUCED gradient: rebuild it from the full-batch result.
        # The shard oracle above stepped its critic with its local gradient, so redo the actor part with the
        # globally updated critic/encoder state.
        ag2 = O.OracleAgent(enc, actor, critic, CFG["lr"], stddev_schedule=CFG["sched"], dtype=torch.float64)
        ag2.critic = {k: v.clone() for k, v in full.critic.items()}       # post-Adam critic of the global step
        featd = local["feat"]
        req = {k: v.detach().clone().requires_grad_(True) for k, v in ag2.actor.items()}
        mu = O.actor_mu(req, featd)
        a = O.trunc_normal_sample(mu, sdraws[3].double(), O.schedule(CFG["sched"], 0), 0.3)
        # NB the global run applies Polyak AFTER the actor step; full.critic is post-Adam (Polyak touches the target)
        q1, q2 = O.critic_q(ag2.critic, featd, a)
        loss = -torch.minimum(q1, q2).mean()
        ga = torch.autograd.grad(loss, list(req.values()))
        for off, g in zip(lay["actor"], ga):
            grads[off:off + g.numel()] = g.reshape(-1) * scale
        dist.all_reduce(grads[b2[0]:b2[1]], op=dist.ReduceOp.SUM)
        for off, g in zip(lay["actor"], full.last["g_actor"].values()):
            assert torch.allclose(grads[off:off + g.numel()], g.reshape(-1), rtol=1e-8, atol=1e-12), rank
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_update_equals_full_batch():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_overlapped_schedule_order_and_deferred_actor_step():
    """Host logic of the overlapped data-parallel schedule (drqv2_amd.engine.StepEngine.update): phase order,
    which arena range each exchange covers, and that Adam(encoder) / Adam(actor) (phases 8 / 9) of update k
    run in update k+1 just before their weights are first read, or at flush()/act()/snapshot time.  No GPU: phases and collectives are recorded."""
    import drqv2
    from drqv2_amd.engine import grad_buckets_overlap
    ag = drqv2.DrQV2Agent((9, 84, 84), (3,), "cpu", 1e-3, 20, 64, 0.01, 2000, 2, "0.2", 0.3, False)
    eng = ag._engine
    log = []

    class Work:
        def __init__(self, name):
            self.name = name

        def wait(self):
            log.append(("wait", self.name))

    def fake_async(t):
        off = (t.data_ptr() - eng.grads.data_ptr()) // 4 if t.data_ptr() != eng.sums.data_ptr() else -1
        name = "sums" if off < 0 else (off, off + t.numel())
        log.append(("allreduce", name))
        return Work(name)

    eng.pg, eng.world, eng.rank = object(), 2, 0
    eng._allreduce_async = fake_async
    eng._phase = lambda d, k: log.append(("phase", k, int(d.step_actor)))
    # the real descriptor builder needs a GPU: a minimal one stands in
    import types
    from drqv2_amd._lib import DrqStep

    def make_desc(self, B_local, B_global, std, clip, tau, steps):
        d = DrqStep()
        d.B, d.global_B = B_local, B_global
        d.step_critic, d.step_enc, d.step_actor = steps
        return d
    eng.make_desc = types.MethodType(make_desc, eng)
    batch = synth.make_batch(4, 3)
    draws = synth.make_draws(8, 3, seed=1)
    args = list(batch) + [t[:4] for t in draws]
    crit, enc, act = grad_buckets_overlap(eng.layout)
    eng.update(*args, 0.2, 0.3, 0.01, B_global=8)
    # default: shard-local metrics -> two hand-overs to the collective library per update
    assert log == [("phase", 3, 1), ("phase", 4, 1), ("allreduce", crit), ("phase", 5, 1), ("wait", crit),
                   ("phase", 6, 1), ("phase", 7, 1), ("allreduce", enc), ("allreduce", act)]
    assert eng._pending is not None and eng._pending_enc is not None
    del log[:]
    eng.global_metrics = True
    eng.update(*args, 0.2, 0.3, 0.01, B_global=8)
    # the deferred steps of update 1 (step numbers 1): Adam(encoder) before phase 3, Adam(actor) before phase 4
    assert log[:6] == [("wait", enc), ("phase", 8, 1), ("phase", 3, 2), ("wait", act), ("phase", 9, 1),
                       ("phase", 4, 2)]
    # global metrics: the sums are exchanged between phases 6 and 7
    assert log[6:] == [("allreduce", crit), ("phase", 5, 2), ("wait", crit), ("phase", 6, 2), ("allreduce", "sums"),
                       ("wait", "sums"), ("phase", 7, 2), ("allreduce", enc), ("allreduce", act)]
    del log[:]
    ag.flush()
    assert log == [("wait", enc), ("phase", 8, 2), ("wait", act), ("phase", 9, 2)]
    assert eng._pending is None and eng._pending_enc is None
    ag.flush()
    assert len(log) == 4                                      # idempotent
    # the three ranges tile [enc_beg, actor_end) of the arena exactly
    seg = eng.layout["seg"]
    assert enc == tuple(seg["enc"]) and crit == tuple(seg["critic"]) and act == tuple(seg["actor"])
    assert enc[1] <= crit[0] and crit[1] <= act[0]


def _exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from drqv2_amd import _lib
        from drqv2_amd.engine import GradExchange, grad_buckets_overlap
        lay = _lib.param_layout(9, 6, 50, 1024)
        out = {}
        for (b, e) in grad_buckets_overlap(lay):
            n = e - b
            assert n % 512 == 0                      # every bucket splits evenly over 2, 4, 8 ranks
            g = torch.Generator().manual_seed(100 + rank)
            x = torch.randn(n, generator=g)
            want = x.clone()
            dist.all_reduce(want, op=dist.ReduceOp.SUM)
            ex = GradExchange(dist.group.WORLD, world, "cpu", "direct")
            assert ex.direct_ok(n)
            y = x.clone()
            ex.start(y).wait()
            # two ranks: a + b in either order is the same float; the direct result is also identical on all ranks
            assert torch.equal(y, want), (rank, n)
            both = [torch.empty_like(y) for _ in range(world)]
            dist.all_gather(both, y)
            assert torch.equal(both[0], both[1])
            ex2 = GradExchange(dist.group.WORLD, world, "cpu", "auto")
            choice = ex2.calibrate([n], iters=2, warmup=1)
            assert choice[n] in ("allreduce", "direct") and set(ex2.timings_us[n]) == {"allreduce", "direct"}
            z = x.clone()
            ex2.start(z).wait()
            assert torch.equal(z, want)
            out[n] = choice[n]
        # a length that does not split falls back to the all-reduce
        ex = GradExchange(dist.group.WORLD, world, "cpu", "direct")
        odd = torch.ones(8)
        assert not ex.direct_ok(8)
        ex.start(odd).wait()
        assert torch.equal(odd, torch.full((8,), float(world)))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_direct_exchange_equals_allreduce_world2():
    """GradExchange 'direct' (all-to-all of slices, rank-order sum, all-gather) on the real bucket sizes of the
    cheetah layout equals the all-reduce, is identical on every rank, and 'auto' picks the same mode everywhere."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1] and ret[0] == ret[1]


def _zero1_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import drq_oracle as O
        from drqv2_amd import _lib
        from drqv2_amd.engine import GradExchange, grad_buckets_overlap
        lay = _lib.param_layout(9, 3, 20, 64)
        lr = 1e-3
        for bi, (b, e) in enumerate(grad_buckets_overlap(lay)):
            n = e - b
            gen = lambda seed: torch.Generator().manual_seed(seed)
            g_local = torch.randn(n, generator=gen(200 + 10 * bi + rank)) * 1e-2
            p0 = torch.randn(n, generator=gen(7 + bi))                   # same on every rank
            m0 = torch.randn(n, generator=gen(8 + bi)) * 1e-3
            v0 = torch.rand(n, generator=gen(9 + bi)) * 1e-4
            # replicated path: SUM of the gradients (direct exchange: rank-order sum), then Adam on the whole segment
            ex_d = GradExchange(dist.group.WORLD, world, "cpu", "direct")
            g_sum = g_local.clone()
            ex_d.start(g_sum).wait()
            p_ref, m_ref, v_ref = p0.clone(), m0.clone(), v0.clone()
            O.adam_step(p_ref, g_sum, m_ref, v_ref, 3, lr)
            # sharded path: the product's slicing and collectives, the oracle's Adam as the per-slice step
            ex = GradExchange(dist.group.WORLD, world, "cpu", "zero1")
            assert ex.sharded(n) and ex._pick(n) == "zero1"
            p, m, v = p0.clone(), m0.clone(), v0.clone()

            def step_fn(ps, copies, ms, vs):
                gs = copies[0].clone()
                for r in range(1, copies.shape[0]):
                    gs += copies[r]                                          # rank order
                O.adam_step(ps, gs, ms, vs, 3, lr)
            ex.start_zero1(g_local.clone(), p, m, v, lr, 3, step_fn=step_fn).wait()
            assert torch.equal(p, p_ref), (rank, bi)                         # every parameter, on every rank
            ns = n // world
            own = slice(rank * ns, (rank + 1) * ns)
            assert torch.equal(m[own], m_ref[own]) and torch.equal(v[own], v_ref[own])
            other = slice((1 - rank) * ns, (2 - rank) * ns)
            assert torch.equal(m[other], m0[other])                          # moments of the other rank's slice: untouched
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_zero1_sharded_adam_equals_replicated_world2():
    """GradExchange 'zero1' (gradient slices in by all-to-all, rank-order sum + Adam on the owned slice, stepped
    parameters out by all-gather) on the three buckets of a small layout: parameters bit-identical to the replicated
    path (direct exchange + Adam on the whole segment) on every rank; each rank maintains the moments of its slice only."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_zero1_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_zero1_schedule_order():
    """Host logic with the sharded optimiser: the optimiser steps travel with the exchanges (no phase 6 / 8 / 9 on the
    compute stream: Polyak (12) and the actor loss (11) instead of 6; the deferred hand-overs only wait)."""
    import types
    import drqv2
    from drqv2_amd._lib import DrqStep
    from drqv2_amd.engine import grad_buckets_overlap
    ag = drqv2.DrQV2Agent((9, 84, 84), (3,), "cpu", 1e-3, 20, 64, 0.01, 2000, 2, "0.2", 0.3, False)
    eng = ag._engine
    log = []

    class Work:
        def __init__(self, name):
            self.name = name

        def wait(self):
            log.append(("wait", self.name))

    class FakeExchange:
        mode = "zero1"

        def sharded(self, n):
            return True

        def start_zero1(self, g, p, m, v, lr, step, gscale=1.0, step_fn=None):
            off = (g.data_ptr() - eng.grads.data_ptr()) // 4
            assert (p.data_ptr() - eng.params.data_ptr()) // 4 == off and p.numel() == g.numel()
            assert (m.data_ptr() - eng.adam_m.data_ptr()) // 4 == off and (v.data_ptr() - eng.adam_v.data_ptr()) // 4 == off
            log.append(("zero1", (off, off + g.numel()), step))
            return Work((off, off + g.numel()))

    eng.pg, eng.world, eng.rank, eng.exchange = object(), 2, 0, FakeExchange()
    eng._phase = lambda d, k: log.append(("phase", k, int(d.step_actor)))

    def make_desc(self, B_local, B_global, std, clip, tau, steps):
        d = DrqStep()
        d.B, d.global_B = B_local, B_global
        d.step_critic, d.step_enc, d.step_actor = steps
        d.lr = 1e-3
        return d
    eng.make_desc = types.MethodType(make_desc, eng)
    batch = synth.make_batch(4, 3)
    draws = synth.make_draws(8, 3, seed=1)
    args = list(batch) + [t[:4] for t in draws]
    crit, enc, act = grad_buckets_overlap(eng.layout)
    eng.update(*args, 0.2, 0.3, 0.01, B_global=8)
    assert log == [("phase", 3, 1), ("phase", 4, 1), ("zero1", crit, 1), ("phase", 5, 1), ("wait", crit),
                   ("phase", 12, 1), ("phase", 11, 1), ("phase", 7, 1), ("zero1", enc, 1), ("zero1", act, 1)]
    del log[:]
    eng.update(*args, 0.2, 0.3, 0.01, B_global=8)
    assert log[:4] == [("wait", enc), ("phase", 3, 2), ("wait", act), ("phase", 4, 2)]     # no phase 8 / 9
    del log[:]
    ag.flush()
    assert log == [("wait", enc), ("wait", act)]


def test_shard_bounds():
    from drqv2_amd.engine import shard_bounds
    assert shard_bounds(256, 1, 0, True) == (0, 256, 256)
    assert shard_bounds(256, 8, 3, True) == (96, 128, 256)
    assert shard_bounds(256, 8, 3, False) == (768, 1024, 2048)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0, True)


def test_bench_self_launch_starts_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent starts N child ranks with the torch.distributed
    environment (127.0.0.1 rendezvous), relays rank 0's line, returns the first non-zero exit code -- and has not
    initialised the GPU itself (no torch.cuda call is reachable before the hand-off: torch is imported later)."""
    import importlib
    import sys
    bench = importlib.import_module("bench")
    started = []

    import io

    class FakeProc:
        """rank 1 fails at once; the others would run on (poll() -> None) until the parent stops them"""
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env, stdout))
            self.rank = int(env["RANK"])
            self.returncode = 3 if self.rank == 1 else None
            self.stdout = io.BytesIO(b'{"metric": "x"}\n') if self.rank == 0 else None
            self.terminated = False

        def poll(self):
            return self.returncode

        def terminate(self):
            self.terminated = True
            self.returncode = -15

        def kill(self):
            self.returncode = -9

        def wait(self, timeout=None):
            return self.returncode

    procs_made = []
    orig_init = FakeProc.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        procs_made.append(self)

    FakeProc.__init__ = init
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse_args(["--gpus", "4", "--steps", "3"])
    rc = bench.launch_ranks(args)
    assert rc == 3 and len(started) == 4
    # the failing rank took the survivors with it (they would otherwise wait for the collective's timeout)
    assert [p.terminated for p in procs_made] == [True, False, True, True]
    ports = set()
    for r, (cmd, env, out) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
        assert (out == bench.subprocess.PIPE) == (r == 0)
    assert len(ports) == 1
    # the module itself imports without torch.cuda work: torch is imported inside main(), after the hand-off
    src = open(bench.__file__).read()
    assert src.index("return launch_ranks(args)") < src.index("import torch\n\n    world = int(")
    # workload defaults: config 2 on one GPU, config 4 (humanoid, ONE global batch of 256) on several
    assert bench.TASKS["humanoid_run"][:2] == (21, 100) and bench.TASKS["cheetah_run"][:2] == (6, 50)
    a1, a8 = bench.parse_args([]), bench.parse_args(["--gpus", "8"])
    assert a1.workload == a8.workload == "auto" and a8.batch == 256 and not a8.weak and a8.exchange == "auto"
