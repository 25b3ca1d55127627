"""Drop-in `utils` module for the reference training loop (train.py:20 does `import utils`).

Same public names and call semantics as /root/reference/utils.py; the tensor work behind
`soft_update_params` runs in the HIP library (drqv2_amd/csrc/elementwise.hip), everything else here is
host-side control flow.  No CPU fallback for the device ops: they raise when handed CPU tensors.
"""
import random
import re
import time

import numpy as np
import torch
import torch.nn as nn
from torch import distributions as pyd
from torch.distributions.utils import _standard_normal

from drqv2_amd import _lib


class eval_mode:
    """Context manager: put models in eval mode, restore the previous flags on exit (utils.py:18-31)."""

    def __init__(self, *models):
        self.models = models
        self.prev_states = []

    def __enter__(self):
        self.prev_states = [m.training for m in self.models]
        for m in self.models:
            m.train(False)

    def __exit__(self, *exc):
        for m, was_training in zip(self.models, self.prev_states):
            m.train(was_training)
        return False


def set_seed_everywhere(seed):
    """utils.py:34-39."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def soft_update_params(net, target_net, tau):
    """target <- tau*net + (1-tau)*target over zipped parameters (utils.py:42-45), on the GPU.

    Arena-backed modules (everything DrQV2Agent builds) are updated with one launch over the whole
    parameter segment; other CUDA modules go tensor by tensor through the same kernel."""
    lib = _lib.load()
    a, b = getattr(net, "_drq_segment", None), getattr(target_net, "_drq_segment", None)
    stream = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
    if a is not None and b is not None and a.numel() == b.numel() and a.is_cuda:
        _lib.check(lib.drq_ema_flat(a.data_ptr(), b.data_ptr(), a.numel(), float(tau), stream), "drq_ema_flat")
        return
    for p, t in zip(net.parameters(), target_net.parameters()):
        if not (p.is_cuda and t.is_cuda):
            raise _lib.DrqError("soft_update_params: parameters must live on the GPU (no CPU path)")
        if not (p.data.is_contiguous() and t.data.is_contiguous() and p.dtype == torch.float32):
            raise _lib.DrqError("soft_update_params: contiguous fp32 parameters required")
        _lib.check(lib.drq_ema_flat(p.data.data_ptr(), t.data.data_ptr(), p.numel(), float(tau), stream),
                   "drq_ema_flat")


def to_torch(xs, device):
    """utils.py:48-49."""
    return tuple(torch.as_tensor(x, device=device) for x in xs)


def weight_init(m):
    """Orthogonal init, zero bias; gain sqrt(2) for conv layers (utils.py:52-61)."""
    if isinstance(m, nn.Linear):
        gain = 1.0
    elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        gain = nn.init.calculate_gain("relu")
    else:
        return
    nn.init.orthogonal_(m.weight.data, gain)
    if hasattr(m.bias, "data"):
        m.bias.data.fill_(0.0)


class Until:
    """True while step < until // action_repeat (utils.py:64-73)."""

    def __init__(self, until, action_repeat=1):
        self._until = until
        self._action_repeat = action_repeat

    def __call__(self, step):
        if self._until is None:
            return True
        return step < self._until // self._action_repeat


class Every:
    """True on every (every // action_repeat)-th step (utils.py:76-87)."""

    def __init__(self, every, action_repeat=1):
        self._every = every
        self._action_repeat = action_repeat

    def __call__(self, step):
        if self._every is None:
            return False
        return step % (self._every // self._action_repeat) == 0


class Timer:
    """Wall-clock helper (utils.py:90-102)."""

    def __init__(self):
        now = time.time()
        self._start_time = now
        self._last_time = now

    def reset(self):
        now = time.time()
        elapsed = now - self._last_time
        self._last_time = now
        return elapsed, now - self._start_time

    def total_time(self):
        return time.time() - self._start_time


class TruncatedNormal(pyd.Normal):
    """Normal whose samples are clamped to [low+eps, high-eps] with a straight-through gradient
    (utils.py:105-126).  Used by DrQV2Agent.act; update() samples inside the HIP step."""

    def __init__(self, loc, scale, low=-1.0, high=1.0, eps=1e-6):
        super().__init__(loc, scale, validate_args=False)
        self.low = low
        self.high = high
        self.eps = eps

    def _clamp(self, x):
        inside = torch.clamp(x, self.low + self.eps, self.high - self.eps)
        return x - x.detach() + inside.detach()

    def sample(self, clip=None, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        noise = _standard_normal(shape, dtype=self.loc.dtype, device=self.loc.device)
        noise *= self.scale
        if clip is not None:
            noise = torch.clamp(noise, -clip, clip)
        return self._clamp(self.loc + noise)


_LINEAR = re.compile(r"linear\((.+),(.+),(.+)\)")
_STEP_LINEAR = re.compile(r"step_linear\((.+),(.+),(.+),(.+),(.+)\)")


def schedule(schdl, step):
    """Scalar schedule: a float, 'linear(a,b,T)' or 'step_linear(a,b1,T1,b2,T2)' (utils.py:129-149)."""
    try:
        return float(schdl)
    except ValueError:
        pass
    m = _LINEAR.match(schdl)
    if m:
        init, final, duration = (float(g) for g in m.groups())
        mix = np.clip(step / duration, 0.0, 1.0)
        return (1.0 - mix) * init + mix * final
    m = _STEP_LINEAR.match(schdl)
    if m:
        init, final1, duration1, final2, duration2 = (float(g) for g in m.groups())
        if step <= duration1:
            mix = np.clip(step / duration1, 0.0, 1.0)
            return (1.0 - mix) * init + mix * final1
        mix = np.clip((step - duration1) / duration2, 0.0, 1.0)
        return (1.0 - mix) * final1 + mix * final2
    raise NotImplementedError(schdl)
