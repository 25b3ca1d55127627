"""Drop-in for the reference's `replay_buffer.py` (names used by `train.py:22,65-72,135,159,187`):
`ReplayBufferStorage(data_specs, replay_dir)` with `.add(time_step)` / `len()`, and
`make_replay_loader(replay_dir, max_size, batch_size, num_workers, save_snapshot, nstep, discount)`.

The reference writes every finished episode to an .npz file which DataLoader workers pick up, and ships each
sampled batch host -> device.  Here a finished episode goes straight into a device-resident store
(`drqv2_amd.replay.DeviceReplay`) and the loader's iterator assembles batches on the GPU with one HIP launch
(`drq_nstep_gather`): what `DrQV2Agent.update` receives has the same shapes, dtypes and n-step arithmetic
(`replay_buffer.py:142-160`), already in HBM.  `num_workers` is accepted and unused (no worker processes).

Resume (train.py:192-204 + `save_snapshot: true`): the reference keeps every episode as
`<replay_dir>/<timestamp>_<idx>_<len>.npz` when save_snapshot is set (its workers delete the files otherwise,
replay_buffer.py:114-115) and reloads them after a restart (`_preload`, `_try_fetch`).  The same here: with
save_snapshot the storage writes the reference's files, and `make_replay_loader` loads the newest episodes found in
`replay_dir` (up to `max_size` transitions, oldest first) into the device store, so `agent.update()` has data right
after `load_snapshot()` -- the reference's own files load too.
"""
import datetime
import io
import pathlib
from collections import defaultdict

import numpy as np
import torch

from drqv2_amd.replay import DeviceReplay

_REGISTRY = {}      # str(replay_dir) -> {"store", "pending": [episodes], "specs", "save_snapshot"}


def episode_len(episode):
    # subtract -1 because the dummy first transition (replay_buffer.py:15-17)
    return next(iter(episode.values())).shape[0] - 1


def _entry(replay_dir):
    return _REGISTRY.setdefault(str(replay_dir), {"store": None, "pending": [], "specs": None, "save_snapshot": False})


def save_episode(episode, fn):
    """replay_buffer.py:22-27: one compressed npz per episode"""
    with io.BytesIO() as bs:
        np.savez_compressed(bs, **episode)
        bs.seek(0)
        with open(fn, "wb") as f:
            f.write(bs.read())


def load_episode(fn):
    """replay_buffer.py:30-34 (np.load default: no pickles)"""
    with open(fn, "rb") as f:
        ep = np.load(f)
        return {k: ep[k] for k in ep.keys()}


def _episode_files(replay_dir):
    """[(path, idx, len)] of the reference-named files in replay_dir, oldest first (names sort by time stamp)"""
    d = pathlib.Path(replay_dir)
    out = []
    if d.is_dir():
        for fn in sorted(d.glob("*.npz")):
            parts = fn.stem.split("_")
            if len(parts) == 3 and parts[1].isdigit() and parts[2].isdigit():
                out.append((fn, int(parts[1]), int(parts[2])))
    return out


class ReplayBufferStorage:
    """replay_buffer.py:34-73: collects the steps of the running episode; a finished episode is handed to the
    device store of the same replay_dir (or kept until make_replay_loader creates it)."""

    def __init__(self, data_specs, replay_dir):
        self._data_specs = data_specs
        self._replay_dir = replay_dir
        self._current_episode = defaultdict(list)
        try:
            pathlib.Path(replay_dir).mkdir(parents=True, exist_ok=True)      # replay_buffer.py:38
        except OSError:
            pass
        self._preload()
        _entry(replay_dir)["specs"] = data_specs

    def _preload(self):
        """replay_buffer.py:62-67: episodes already on disk count (a resumed run skips the seed phase)"""
        files = _episode_files(self._replay_dir)
        self._num_episodes = len(files)
        self._num_transitions = sum(n for _, _, n in files)

    def __len__(self):
        return self._num_transitions

    def add(self, time_step):
        step = self._current_episode
        for spec in self._data_specs:
            v = time_step[spec.name]
            if np.isscalar(v):                       # dm_env hands reward / discount over as python scalars
                v = np.full(spec.shape, v, spec.dtype)
            if v.shape != spec.shape or v.dtype != spec.dtype:
                raise AssertionError(f"{spec.name}: got {v.shape} {v.dtype}, the spec says {spec.shape} {spec.dtype}")
            step[spec.name].append(v)
        if not time_step.last():
            return
        done = {spec.name: np.array(step[spec.name], spec.dtype) for spec in self._data_specs}
        self._current_episode = defaultdict(list)
        self._store_episode(done)

    def _store_episode(self, episode):
        eps_idx, eps_len = self._num_episodes, episode_len(episode)
        self._num_episodes += 1
        self._num_transitions += eps_len
        ent = _entry(self._replay_dir)
        if ent["save_snapshot"]:                      # replay_buffer.py:69-78 (same file name, same content)
            ts = datetime.datetime.now().strftime("%Y%m%dT%H%M%S")
            save_episode(episode, pathlib.Path(self._replay_dir) / f"{ts}_{eps_idx}_{eps_len}.npz")
        if ent["store"] is not None:
            ent["store"].add_episode(episode)
        else:
            ent["pending"].append(episode)


class _Loader:
    """What `iter(make_replay_loader(...))` yields from: device batches of `batch_size` rows."""

    def __init__(self, replay_dir, store, batch_size):
        self._replay_dir = replay_dir
        self._store = store
        self._batch_size = batch_size

    def __iter__(self):
        from drqv2_amd.replay import BatchIterator
        ent = _entry(self._replay_dir)

        def draw():
            while ent["pending"]:
                self._store.add_episode(ent["pending"].pop(0))
            return self._store.sample(self._batch_size)
        return BatchIterator(draw)


def make_replay_loader(replay_dir, max_size, batch_size, num_workers, save_snapshot, nstep, discount, device=None,
                       obs_shape=None, action_dim=None, seed=None, indexed=False):
    """Same positional signature as the reference (replay_buffer.py:173-190).  The observation / action shapes
    come from the data_specs the storage of the same replay_dir was built with (or from the keyword arguments)."""
    ent = _entry(replay_dir)
    if ent["specs"] is not None:
        by_name = {s.name: s for s in ent["specs"]}
        obs_shape = obs_shape or tuple(by_name["observation"].shape)
        action_dim = action_dim or int(np.prod(by_name["action"].shape))
    if obs_shape is None or action_dim is None:
        raise ValueError("make_replay_loader: create the ReplayBufferStorage first or pass obs_shape/action_dim")
    device = torch.device(device if device is not None else "cuda")
    # capacity in steps: the reference counts transitions; every episode also stores its dummy first step
    capacity = int(max_size) + int(max_size) // 100 + 1024
    if seed is None:
        seed = int(np.random.get_state()[1][0])          # what the reference's _worker_init_fn seeds from
    # indexed=True: the iterator yields drqv2_amd.replay.IndexedBatch objects (frames stay in the store, the update's first
    # kernel gathers them): what DrQV2Agent.update() consumes fastest; False: plain device tensors like the reference's
    ent["store"] = DeviceReplay(capacity, obs_shape, action_dim, nstep, discount, device, seed=seed, indexed=indexed)
    ent["save_snapshot"] = bool(save_snapshot)
    # resume: the newest episodes on disk that fit max_size (replay_buffer.py:120-140 walks them newest first),
    # added oldest first so that eviction order stays chronological
    keep, size = [], 0
    for fn, _, n in reversed(_episode_files(replay_dir)):
        if size + n > int(max_size):
            break
        keep.append(fn)
        size += n
    for fn in reversed(keep):
        try:
            ent["store"].add_episode(load_episode(fn))
        except (OSError, ValueError, KeyError):
            continue                                  # an unreadable file is skipped, as the reference does (:104-107)
    return _Loader(replay_dir, ent["store"], batch_size)
