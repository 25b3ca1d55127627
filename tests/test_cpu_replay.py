"""Host logic of the device-resident replay (drqv2_amd/replay.py, replay_buffer.py drop-in names): placement and
whole-episode eviction, the sampling ranges of replay_buffer.py:145-150, the storage's episode assembly, and the
loud failure of batch assembly without a GPU."""
import numpy as np
import pytest
import torch

from drqv2_amd.replay import DeviceReplay
from drqv2_amd._lib import DrqError

OBS = (1, 4, 4)      # 16-byte frames


def episode(T, A=2, seed=0):
    r = np.random.RandomState(seed)
    return {"observation": r.randint(0, 256, (T + 1,) + OBS).astype(np.uint8),
            "action": r.uniform(-1, 1, (T + 1, A)).astype(np.float32),
            "reward": r.randn(T + 1, 1).astype(np.float32),
            "discount": np.ones((T + 1, 1), np.float32)}


def test_placement_wraps_and_evicts_whole_oldest_episodes():
    rp = DeviceReplay(20, OBS, 2, nstep=3, discount=0.99, device="cpu", seed=0)
    assert rp.add_episode(episode(5, seed=1)) == 0          # 6 steps: slots 0..5
    assert rp.add_episode(episode(7, seed=2)) == 6          # 8 steps: 6..13
    assert len(rp) == 12 and rp.episodes == [[0, 6], [6, 8]]
    assert rp.add_episode(episode(4, seed=3)) == 14         # 5 steps: 14..18
    # 6 steps do not fit behind slot 19: wrap to 0, the oldest episode (slots 0..5) goes
    assert rp.add_episode(episode(5, seed=4)) == 0
    assert rp.episodes == [[6, 8], [14, 5], [0, 6]]
    # the next one overlaps the second-oldest only partly: it is dropped whole
    assert rp.add_episode(episode(2, seed=5)) == 6
    assert rp.episodes == [[14, 5], [0, 6], [6, 3]]
    assert len(rp) == 4 + 5 + 2
    ep = episode(5, seed=4)
    assert torch.equal(rp.frames[0:6], torch.from_numpy(ep["observation"].reshape(6, 16)))
    assert torch.equal(rp.reward[0:6], torch.from_numpy(ep["reward"].reshape(6)))
    with pytest.raises(ValueError):
        rp.add_episode(episode(30))


def test_sampling_ranges_match_reference_rule():
    rp = DeviceReplay(64, OBS, 2, nstep=3, discount=0.99, device="cpu", seed=7)
    rp.add_episode(episode(2, seed=1))       # shorter than nstep: never sampled (the reference would raise on it)
    s1 = rp.add_episode(episode(3, seed=2))  # exactly nstep: idx can only be 1
    s2 = rp.add_episode(episode(10, seed=3))
    pos = rp.draw_positions(4000)
    in1 = pos[(pos >= s1) & (pos < s1 + 4)] - s1
    in2 = pos[(pos >= s2) & (pos < s2 + 11)] - s2
    assert len(in1) + len(in2) == 4000
    assert set(in1.tolist()) == {1}
    assert set(in2.tolist()) == set(range(1, 10 - 3 + 2))       # idx in [1, len - nstep + 1]
    assert abs(len(in1) / 4000 - 0.5) < 0.05                    # episodes uniform, whatever their length
    with pytest.raises(DrqError):
        rp.gather(pos[:8])                                      # batch assembly is a HIP kernel: no CPU fallback
    empty = DeviceReplay(8, OBS, 2, nstep=3, discount=0.99, device="cpu")
    with pytest.raises(DrqError):
        empty.draw_positions(4)


class _Spec:
    def __init__(self, name, shape, dtype):
        self.name, self.shape, self.dtype = name, shape, dtype


class _Step(dict):
    def __init__(self, last, **kw):
        super().__init__(**kw)
        self._last = last

    def last(self):
        return self._last


def test_storage_and_loader_dropin_names(tmp_path):
    import replay_buffer as rb
    specs = (_Spec("observation", OBS, np.uint8), _Spec("action", (2,), np.float32),
             _Spec("reward", (1,), np.float32), _Spec("discount", (1,), np.float32))
    st = rb.ReplayBufferStorage(specs, tmp_path / "buffer")
    ep = episode(4, seed=9)
    for t in range(5):
        st.add(_Step(t == 4, observation=ep["observation"][t], action=ep["action"][t],
                     reward=float(ep["reward"][t, 0]), discount=1.0))       # scalars are broadcast to the spec shape
    assert len(st) == 4
    loader = rb.make_replay_loader(tmp_path / "buffer", 100, 8, 4, False, 3, 0.99, device="cpu")
    store = rb._REGISTRY[str(tmp_path / "buffer")]["store"]
    assert store.nstep == 3 and store.A == 2 and store.obs_shape == OBS
    it = iter(loader)
    with pytest.raises(DrqError):            # the pending episode is flushed into the store, then the gather needs the GPU
        next(it)
    assert store.episodes == [[0, 5]]
    assert torch.equal(store.frames[:5], torch.from_numpy(ep["observation"].reshape(5, 16)))
    assert torch.equal(store.action[:5], torch.from_numpy(ep["action"]))


def test_resume_reloads_reference_named_episode_files(tmp_path):
    """save_snapshot=True: finished episodes are written as <ts>_<idx>_<len>.npz like the reference's storage does,
    and a NEW storage + loader on the same directory (a resumed run, train.py:199-204) starts with them in the device
    store: the first update() after load_snapshot() has data.  Only the newest episodes that fit max_size load."""
    import replay_buffer as rb
    d = tmp_path / "buffer"
    specs = (_Spec("observation", OBS, np.uint8), _Spec("action", (2,), np.float32),
             _Spec("reward", (1,), np.float32), _Spec("discount", (1,), np.float32))
    st = rb.ReplayBufferStorage(specs, d)
    rb.make_replay_loader(d, 100, 8, 4, True, 3, 0.99, device="cpu")
    eps = [episode(4, seed=11), episode(6, seed=12), episode(5, seed=13)]
    for ep in eps:
        T1 = ep["observation"].shape[0]
        for t in range(T1):
            st.add(_Step(t == T1 - 1, observation=ep["observation"][t], action=ep["action"][t],
                         reward=ep["reward"][t], discount=ep["discount"][t]))
    files = sorted(d.glob("*.npz"))
    assert [f.stem.split("_")[1:] for f in files] == [["0", "4"], ["1", "6"], ["2", "5"]]
    back = rb.load_episode(files[1])
    assert all(np.array_equal(back[k], eps[1][k]) for k in eps[1])
    # "restart": forget the in-memory registry, build storage + loader again on the same directory
    rb._REGISTRY.clear()
    st2 = rb.ReplayBufferStorage(specs, d)
    assert len(st2) == 15 and st2._num_episodes == 3                 # _preload (replay_buffer.py:62-67)
    rb.make_replay_loader(d, 100, 8, 4, True, 3, 0.99, device="cpu")
    store = rb._REGISTRY[str(d)]["store"]
    assert store.episodes == [[0, 5], [5, 7], [12, 6]] and len(store) == 15
    assert torch.equal(store.frames[5:12], torch.from_numpy(eps[1]["observation"].reshape(7, 16)))
    # max_size smaller than what is on disk: the newest episodes win
    rb._REGISTRY.clear()
    rb.ReplayBufferStorage(specs, d)
    rb.make_replay_loader(d, 11, 8, 4, True, 3, 0.99, device="cpu")
    store = rb._REGISTRY[str(d)]["store"]
    assert len(store) == 11 and torch.equal(store.frames[:7], torch.from_numpy(eps[1]["observation"].reshape(7, 16)))
    # without save_snapshot nothing is written
    d2 = tmp_path / "buffer2"
    st3 = rb.ReplayBufferStorage(specs, d2)
    rb.make_replay_loader(d2, 100, 8, 4, False, 3, 0.99, device="cpu")
    ep = eps[0]
    for t in range(5):
        st3.add(_Step(t == 4, observation=ep["observation"][t], action=ep["action"][t], reward=ep["reward"][t],
                      discount=ep["discount"][t]))
    assert not list(d2.glob("*.npz"))


def test_batch_iterator_look_ahead_keeps_the_order_of_the_draws():
    """drqv2_amd.replay.BatchIterator: prefetch() draws the next batch early (DrQV2Agent.update() calls it once its
    kernels are queued); the sequence handed out is that of plain next() calls, and at most one batch is held."""
    from drqv2_amd.replay import BatchIterator
    n = [0]

    def draw():
        n[0] += 1
        return n[0]
    it = BatchIterator(draw)
    assert iter(it) is it
    assert next(it) == 1
    it.prefetch()
    it.prefetch()                      # a second call does not draw again
    assert n[0] == 2
    assert next(it) == 2 and next(it) == 3
    it.prefetch()
    assert n[0] == 4 and next(it) == 4
