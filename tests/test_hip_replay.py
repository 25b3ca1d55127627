"""Device replay batch assembly (drq_nstep_gather through drqv2_amd.replay / replay_buffer.py) against the oracle's
restatement of `_sample` (oracle.nstep_sample, pinned by tests/golden/nstep.json): bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
OBS = (9, 84, 84)


def episode(T, A, seed):
    r = np.random.RandomState(seed)
    d = np.ones((T + 1, 1), np.float32)
    d[-1] = 0.0 if seed % 2 else 1.0                 # a terminal step with discount 0 in some episodes
    return {"observation": r.randint(0, 256, (T + 1,) + OBS).astype(np.uint8),
            "action": r.uniform(-1, 1, (T + 1, A)).astype(np.float32),
            "reward": r.randn(T + 1, 1).astype(np.float32), "discount": d}


@pytest.mark.parametrize("nstep,gamma,A", [(3, 0.99, 6), (1, 0.99, 1), (5, 0.9, 21)])
def test_batches_match_oracle_bit_for_bit(nstep, gamma, A):
    from drqv2_amd.replay import DeviceReplay
    from oracle import drq_oracle as O
    rp = DeviceReplay(200, OBS, A, nstep, gamma, "cuda", seed=3)
    eps = {}
    for i, T in enumerate((5, 17, 9, 30, 12)):
        e = episode(T, A, seed=10 + i)
        eps[rp.add_episode(e)] = e
    B = 96
    pos = rp.draw_positions(B)
    obs, act, rew, disc, nxt = (t.cpu().numpy() for t in rp.gather(pos))
    assert obs.shape == (B,) + OBS and rew.shape == (B, 1) and act.shape == (B, A)
    for b, p in enumerate(pos.tolist()):
        start = max(s for s in eps if s <= p)
        o, a, r, d, n = O.nstep_sample(eps[start], p - start, nstep, gamma)
        assert np.array_equal(obs[b], o) and np.array_equal(nxt[b], n) and np.array_equal(act[b], a)
        assert rew[b, 0] == r[0] and disc[b, 0] == d[0], (b, rew[b, 0], r[0])


def test_update_consumes_device_batches(tmp_path):
    """train.py's wiring: storage.add(time_step) ... agent.update(iter(loader), step) with no host copy of the batch"""
    import drqv2
    import replay_buffer as rb

    class Spec:
        def __init__(self, name, shape, dtype):
            self.name, self.shape, self.dtype = name, shape, dtype

    class Step(dict):
        def last(self):
            return self["_last"]

    A = 3
    specs = (Spec("observation", OBS, np.uint8), Spec("action", (A,), np.float32), Spec("reward", (1,), np.float32),
             Spec("discount", (1,), np.float32))
    st = rb.ReplayBufferStorage(specs, tmp_path / "buffer")
    loader = rb.make_replay_loader(tmp_path / "buffer", 500, 16, 4, False, 3, 0.99, seed=1)
    for e in range(3):
        ep = episode(20, A, seed=e)
        for t in range(21):
            st.add(Step(observation=ep["observation"][t], action=ep["action"][t], reward=ep["reward"][t],
                        discount=ep["discount"][t], _last=(t == 20)))
    assert len(st) == 60
    ag = drqv2.DrQV2Agent(OBS, (A,), "cuda", 1e-3, 20, 64, 0.01, 2000, 2, "0.2", 0.3, True)
    it = iter(loader)
    for step in (0, 2, 4):
        m = ag.update(it, step)
        assert all(np.isfinite(v) for v in m.values()) and set(m) >= {"critic_loss", "actor_loss", "batch_reward"}
    batch = next(it)
    assert all(t.is_cuda for t in batch) and batch[0].dtype == torch.uint8 and batch[0].shape == (16,) + OBS


def test_fused_aug_conv1_gathers_from_the_store_bit_for_bit():
    """drq_conv1_aug_fwd_indexed (the two views given as rows of a frame store) against drq_conv1_aug_fwd on the
    gathered frames: identical layer output and identical stored encoder input, every element."""
    from drqv2_amd import ops
    r = np.random.RandomState(0)
    slots, n = 300, 37
    frames = torch.from_numpy(r.randint(0, 256, (slots, 9 * 84 * 84)).astype(np.uint8)).cuda()
    idx = torch.from_numpy(r.randint(0, slots, n).astype(np.int64)).cuda()
    idx1 = torch.from_numpy(r.randint(0, slots, n).astype(np.int64)).cuda()
    sh = torch.from_numpy(r.randint(0, 9, (n, 2)).astype(np.float32)).cuda()
    sh1 = torch.from_numpy(r.randint(0, 9, (n, 2)).astype(np.float32)).cuda()
    w = torch.from_numpy((r.randn(32, 9, 3, 3) * 0.2).astype(np.float32)).cuda()
    b = torch.from_numpy((r.randn(32) * 0.1).astype(np.float32)).cuda()
    obs = frames[idx].view(n, 9, 84, 84).contiguous()
    obs1 = frames[idx1].view(n, 9, 84, 84).contiguous()
    y0, x0 = ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=2 * n)
    y1, x1 = ops.conv1_aug_fwd_indexed(frames, idx, sh, frames, idx1, sh1, w, b, n_store=2 * n)
    assert torch.equal(y0, y1) and torch.equal(x0, x1)


def test_update_from_indexed_batches_equals_materialised_batches():
    """IndexedBatch (frames stay in the store; obs / next_obs travel as indices) through DrQV2Agent.update(): the same
    update as with the batch drq_nstep_gather materialises -- parameters, Adam moments and metrics bit for bit, three
    updates, same index draws."""
    import drqv2
    from drqv2_amd.replay import DeviceReplay, IndexedBatch
    A, B = 6, 64
    outs = []
    for indexed in (False, True):
        rp = DeviceReplay(400, OBS, A, 3, 0.99, "cuda", seed=5, indexed=indexed)
        for i, T in enumerate((40, 25, 60)):
            rp.add_episode(episode(T, A, seed=20 + i))
        rp.batch_size = B
        torch.manual_seed(3)
        ag = drqv2.DrQV2Agent(OBS, (A,), "cuda", 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,500000)", 0.3, True)
        torch.manual_seed(11); torch.cuda.manual_seed_all(11)
        it = iter(rp)
        ms = [ag.update(it, 2 * u) for u in range(3)]
        torch.cuda.synchronize()
        b = next(it)
        assert isinstance(b, IndexedBatch) == indexed
        if indexed:
            obs, act, rew, disc, nxt = b.materialize(OBS)
            assert obs.shape == (B,) + OBS and obs.dtype == torch.uint8 and rew.shape == (B, 1)
        eng = ag._engine
        outs.append((ms, eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()))
    (m0, *a0), (m1, *a1) = outs
    assert m0 == m1
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)


def test_indexed_batches_without_metrics_reads_host_running_ahead():
    """use_tb=False: update() returns without reading the metrics, so nothing but StepEngine._throttle keeps the host
    from drawing batch after batch while the GPU is still on the first -- and the device replay stages its index lists
    in four pinned host sets that it re-uses.  Forty updates, back to back, must leave exactly the state of the same
    forty updates issued with the metrics read every time (bit for bit), and the engine never holds more than two
    updates behind the running one."""
    import drqv2
    from drqv2_amd.replay import DeviceReplay
    A, B, N = 6, 64, 40
    outs = []
    for use_tb in (True, False):
        rp = DeviceReplay(400, OBS, A, 3, 0.99, "cuda", seed=5, indexed=True)
        for i, T in enumerate((40, 25, 60)):
            rp.add_episode(episode(T, A, seed=20 + i))
        rp.batch_size = B
        torch.manual_seed(3)
        ag = drqv2.DrQV2Agent(OBS, (A,), "cuda", 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,500000)", 0.3, use_tb)
        torch.manual_seed(11); torch.cuda.manual_seed_all(11)
        it = iter(rp)
        depth = 0
        for u in range(N):
            m = ag.update(it, 2 * u)
            assert (m == {}) == (not use_tb)
            eng = ag._engine
            pub = int(eng._sums_seq) & 0xFFFFFFFF
            last = eng._enqueued[-1]
            depth = max(depth, (last - pub) & 0xFFFFFFFF if pub else 0)
        torch.cuda.synchronize()
        assert depth <= 3, depth          # the one running + two queued
        outs.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
