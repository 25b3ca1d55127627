"""Drop-in for the reference's `replay_buffer.py` (names used by `train.py:22,65-72,135,159,187`):
`ReplayBufferStorage(data_specs, replay_dir)` with `.add(time_step)` / `len()`, and
`make_replay_loader(replay_dir, max_size, batch_size, num_workers, save_snapshot, nstep, discount)`.

The reference writes every finished episode to an .npz file which DataLoader workers pick up, and ships each
sampled batch host -> device.  Here a finished episode goes straight into a device-resident store
(`drqv2_amd.replay.DeviceReplay`) and the loader's iterator assembles batches on the GPU with one HIP launch
(`drq_nstep_gather`): what `DrQV2Agent.update` receives has the same shapes, dtypes and n-step arithmetic
(`replay_buffer.py:142-160`), already in HBM.  `num_workers` and `save_snapshot` are accepted and unused: there are
no worker processes and nothing is written to disk.
"""
from collections import defaultdict

import numpy as np
import torch

from drqv2_amd.replay import DeviceReplay

_REGISTRY = {}      # str(replay_dir) -> {"store": DeviceReplay | None, "pending": [episodes], "specs": data_specs}


def episode_len(episode):
    # subtract -1 because the dummy first transition (replay_buffer.py:15-17)
    return next(iter(episode.values())).shape[0] - 1


def _entry(replay_dir):
    return _REGISTRY.setdefault(str(replay_dir), {"store": None, "pending": [], "specs": None})


class ReplayBufferStorage:
    """replay_buffer.py:34-73: collects the steps of the running episode; a finished episode is handed to the
    device store of the same replay_dir (or kept until make_replay_loader creates it)."""

    def __init__(self, data_specs, replay_dir):
        self._data_specs = data_specs
        self._replay_dir = replay_dir
        self._current_episode = defaultdict(list)
        self._num_episodes = 0
        self._num_transitions = 0
        _entry(replay_dir)["specs"] = data_specs

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
        self._num_episodes += 1
        self._num_transitions += episode_len(episode)
        ent = _entry(self._replay_dir)
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
        ent = _entry(self._replay_dir)
        while True:
            while ent["pending"]:
                self._store.add_episode(ent["pending"].pop(0))
            yield self._store.sample(self._batch_size)


def make_replay_loader(replay_dir, max_size, batch_size, num_workers, save_snapshot, nstep, discount, device=None,
                       obs_shape=None, action_dim=None, seed=None):
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
    ent["store"] = DeviceReplay(capacity, obs_shape, action_dim, nstep, discount, device, seed=seed)
    return _Loader(replay_dir, ent["store"], batch_size)
