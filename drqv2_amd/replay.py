"""Device-resident replay store and batch assembly.

The reference keeps episodes as numpy dicts in DataLoader workers and ships every batch host -> pinned ->
device (`replay_buffer.py:76-190`): 32.5 MB per update at batch 256, which caps training at the PCIe rate.
Here the steps live in HBM (uint8 frames: 63,504 B per step, 1 M steps = 63.5 GB of the 288 GB) and
`drq_nstep_gather` (one launch) assembles the batch in place: two frame gathers, the action rows and the
n-step reward/discount accumulation of `_sample` (`replay_buffer.py:142-160`) in the reference's float32 order.

Semantics kept from the reference: episodes are stored whole, each with its dummy first transition; eviction
drops the OLDEST whole episodes until the new one fits (`_store_episode`, :100-118); a sample picks an episode
uniformly, then `idx` uniformly in [1, len - nstep + 1] (:145-150).  Not kept: the on-disk npz files and the
worker processes (their per-worker RNG streams are not reproducible in the reference either); index draws come
from one numpy RandomState, vectorised over the batch.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


class IndexedBatch(tuple):
    """A replay batch whose frames stay in the store: the 5-tuple (obs, action, reward, discount, next_obs) of the
    reference's loader with obs / next_obs replaced by int64 index tensors [B] into `frames` (the store, [slots, frame
    bytes] uint8).  DrQV2Agent.update() hands store + indices to the fused aug+conv1 launch, which gathers its source
    rows itself: the 32.5 MB batch copy (and its re-read) of the materialised form never happens."""

    def __new__(cls, frames, obs_idx, action, reward, discount, next_idx):
        self = super().__new__(cls, (obs_idx, action, reward, discount, next_idx))
        self.frames = frames
        return self

    def materialize(self, obs_shape):
        """The batch as the reference's loader would hand it over: (obs, action, reward, discount, next_obs) tensors."""
        obs_idx, action, reward, discount, next_idx = self
        shp = (obs_idx.numel(),) + tuple(obs_shape)
        return (self.frames[obs_idx].view(shp), action, reward, discount, self.frames[next_idx].view(shp))


class BatchIterator:
    """Endless iterator over device batches with one batch of look-ahead: DrQV2Agent.update() calls prefetch() right
    after it has queued its kernels, so the next batch's host work (index draw, upload, gather launch) happens while the
    GPU is busy instead of in front of the next update's first launch (the reference's DataLoader workers run further
    ahead still, replay_buffer.py:173-190).  The draws and their order are those of plain next() calls."""

    def __init__(self, draw):
        self._draw = draw
        self._ahead = None

    def __iter__(self):
        return self

    def __next__(self):
        b, self._ahead = self._ahead, None
        return b if b is not None else self._draw()

    def prefetch(self):
        if self._ahead is None:
            self._ahead = self._draw()


class DeviceReplay:
    def __init__(self, capacity_steps, obs_shape, action_dim, nstep, discount, device, seed=None, indexed=False):
        self.device = torch.device(device)
        self.obs_shape = tuple(int(s) for s in obs_shape)
        self.frame_bytes = int(np.prod(self.obs_shape))
        if self.frame_bytes % 16:
            raise ValueError("frame size must be a multiple of 16 bytes")
        self.A = int(action_dim)
        self.nstep = int(nstep)
        self.gamma = float(discount)
        self.capacity = int(capacity_steps)
        dev = self.device
        self.frames = torch.empty((self.capacity, self.frame_bytes), dtype=torch.uint8, device=dev)
        self.action = torch.zeros((self.capacity, self.A), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((self.capacity,), dtype=torch.float32, device=dev)
        self.discount = torch.ones((self.capacity,), dtype=torch.float32, device=dev)
        self.episodes = []          # [start_slot, steps (= T+1)], oldest first; each contiguous in the store
        self._head = 0              # next free slot
        self.rng = np.random.RandomState(seed)
        self._out = {}
        # True: batches are IndexedBatch objects (frames stay in the store); False: materialised tensors like the
        # reference's loader yields
        self.indexed = bool(indexed)
        self._ibufs = {}

    # ---- storage -------------------------------------------------------------------------
    def __len__(self):
        """transitions stored, as the reference counts them (episode_len = steps - 1)"""
        return sum(n - 1 for _, n in self.episodes)

    def _place(self, n):
        """slot range for an episode of n steps: contiguous, wrapping to slot 0 when the tail is too short;
        evicts every stored episode the range overlaps (they are the oldest ones)."""
        if n > self.capacity:
            raise ValueError(f"episode of {n} steps exceeds the store ({self.capacity})")
        start = self._head if self._head + n <= self.capacity else 0
        end = start + n
        self.episodes = [e for e in self.episodes if e[0] + e[1] <= start or e[0] >= end]
        self._head = end
        return start

    def add_episode(self, episode):
        """episode: dict of numpy arrays like the reference's npz (observation [T+1,...] uint8, action [T+1,A],
        reward [T+1,1] or [T+1], discount likewise); index 0 is the dummy reset transition."""
        obs = np.ascontiguousarray(episode["observation"])
        n = obs.shape[0]
        if obs.dtype != np.uint8 or int(np.prod(obs.shape[1:])) != self.frame_bytes:
            raise ValueError("observation must be uint8 frames of the configured shape")
        start = self._place(n)
        sl = slice(start, start + n)
        dev = self.device
        self.frames[sl].copy_(torch.from_numpy(obs.reshape(n, self.frame_bytes)), non_blocking=False)
        self.action[sl].copy_(torch.from_numpy(np.asarray(episode["action"], np.float32).reshape(n, self.A)))
        self.reward[sl].copy_(torch.from_numpy(np.asarray(episode["reward"], np.float32).reshape(n)))
        self.discount[sl].copy_(torch.from_numpy(np.asarray(episode["discount"], np.float32).reshape(n)))
        self.episodes.append([start, n])
        return start

    # ---- sampling ------------------------------------------------------------------------
    def draw_positions(self, batch_size):
        """store indices of `idx` for a batch: episode uniform, idx uniform in [1, len - nstep + 1]
        (replay_buffer.py:147-148).  Episodes shorter than nstep cannot be sampled (the reference would raise)."""
        ok = [(s, n) for s, n in self.episodes if n - 1 >= self.nstep]
        if not ok:
            raise _lib.DrqError("replay: no stored episode is at least nstep long")
        starts = np.array([s for s, _ in ok], np.int64)
        lens = np.array([n - 1 for _, n in ok], np.int64)
        e = self.rng.randint(0, len(ok), size=batch_size)
        idx = (self.rng.random_sample(batch_size) * (lens[e] - self.nstep + 1)).astype(np.int64) + 1
        return starts[e] + idx

    def gather(self, pos):
        """pos: int64 store indices [B] (host array or device tensor) -> (obs, action, reward, discount, next_obs) on
        the device, shaped like the reference's batch ([B,*obs], [B,A], [B,1], [B,1], [B,*obs])."""
        if self.device.type != "cuda":
            raise _lib.DrqError("replay batch assembly runs on the GPU: the HIP path has no CPU fallback")
        lib = _lib.load()
        if not torch.is_tensor(pos):
            pos = torch.from_numpy(np.ascontiguousarray(pos, np.int64))
        pos = pos.to(self.device, non_blocking=True)
        B = pos.numel()
        out = self._out.get(B)
        if out is None:
            dev = self.device
            out = (torch.empty((B, self.frame_bytes), dtype=torch.uint8, device=dev),
                   torch.empty((B, self.A), dtype=torch.float32, device=dev),
                   torch.empty((B, 1), dtype=torch.float32, device=dev),
                   torch.empty((B, 1), dtype=torch.float32, device=dev),
                   torch.empty((B, self.frame_bytes), dtype=torch.uint8, device=dev))
            self._out = {B: out}            # one batch size at a time; the buffers are reused every call
        obs, act, rew, disc, nxt = out
        check(lib.drq_nstep_gather(ptr(self.frames), ptr(self.action), ptr(self.reward), ptr(self.discount), ptr(pos), B,
                                   self.A, self.frame_bytes, self.nstep, self.gamma, ptr(obs), ptr(act), ptr(rew),
                                   ptr(disc), ptr(nxt), torch.cuda.current_stream().cuda_stream), "drq_nstep_gather")
        shp = (B,) + self.obs_shape
        return obs.view(shp), act, rew, disc, nxt.view(shp)

    def gather_indexed(self, pos):
        """pos: host int64 store indices [B] -> IndexedBatch: action rows and n-step reward / discount assembled by the
        same kernel (its frame copies skipped), obs = frame pos-1, next_obs = frame pos+nstep-1 as indices."""
        if self.device.type != "cuda":
            raise _lib.DrqError("replay batch assembly runs on the GPU: the HIP path has no CPU fallback")
        lib = _lib.load()
        pos = np.ascontiguousarray(pos, np.int64)
        B = pos.size
        bufs = self._ibufs.get(B)
        if bufs is None:
            dev = self.device
            # four sets, used in turn.  The device tensors are written in stream order (no hazard), but the pinned host
            # staging of set s is overwritten by the HOST when batch k+4 is drawn, and its upload must have run by then:
            # StepEngine.update keeps at most two updates queued behind the running one (_throttle), so three batches
            # can be pending at most.  A consumer of its own must bound its run-ahead likewise (or use gather()).
            mk = lambda: (torch.empty((3, B), dtype=torch.int64, device=dev), torch.empty((B, self.A), dtype=torch.float32, device=dev),
                          torch.empty((B, 1), dtype=torch.float32, device=dev), torch.empty((B, 1), dtype=torch.float32, device=dev),
                          torch.empty((3, B), dtype=torch.int64).pin_memory())
            bufs = [mk(), mk(), mk(), mk(), 0]
            self._ibufs = {B: bufs}
        idx, act, rew, disc, host = bufs[bufs[4]]
        bufs[4] = (bufs[4] + 1) & 3
        h = host.numpy()
        h[0], h[1], h[2] = pos - 1, pos + self.nstep - 1, pos
        idx.copy_(host, non_blocking=True)
        check(lib.drq_nstep_gather(ptr(self.frames), ptr(self.action), ptr(self.reward), ptr(self.discount), ptr(idx[2]), B,
                                   self.A, self.frame_bytes, self.nstep, self.gamma, None, ptr(act), ptr(rew), ptr(disc),
                                   None, torch.cuda.current_stream().cuda_stream), "drq_nstep_gather")
        return IndexedBatch(self.frames, idx[0], act, rew, disc, idx[1])

    def sample(self, batch_size):
        pos = self.draw_positions(batch_size)
        return self.gather_indexed(pos) if self.indexed else self.gather(pos)

    def __iter__(self):
        return BatchIterator(lambda: self.sample(self.batch_size))

    batch_size = 256
