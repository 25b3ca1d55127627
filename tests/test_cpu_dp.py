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
        from drqv2_amd.engine import grad_buckets, shard_bounds
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


def test_shard_bounds():
    from drqv2_amd.engine import shard_bounds
    assert shard_bounds(256, 1, 0, True) == (0, 256, 256)
    assert shard_bounds(256, 8, 3, True) == (96, 128, 256)
    assert shard_bounds(256, 8, 3, False) == (768, 1024, 2048)
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0, True)
