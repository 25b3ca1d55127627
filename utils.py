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
        self._mods = tuple(models)
        self._flags = ()

    def __enter__(self):
        self._flags = tuple(m.training for m in self._mods)
        for m in self._mods:
            m.train(False)

    def __exit__(self, *exc):
        for m, flag in zip(self._mods, self._flags):
            m.train(flag)
        return False


def set_seed_everywhere(seed):
    """Seeds python, numpy and torch (all devices) from one integer (utils.py:34-39)."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


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
    """True while step < until // action_repeat; always True without a limit (utils.py:64-73)."""

    def __init__(self, until, action_repeat=1):
        self._limit = None if until is None else until // action_repeat

    def __call__(self, step):
        return self._limit is None or step < self._limit


class Every:
    """True on every (every // action_repeat)-th step; never without a period (utils.py:76-87)."""

    def __init__(self, every, action_repeat=1):
        self._period = None if every is None else every // action_repeat

    def __call__(self, step):
        return self._period is not None and step % self._period == 0


class Timer:
    """Wall-clock helper: reset() -> (seconds since the last reset, seconds since construction) (utils.py:90-102)."""

    def __init__(self):
        self._t0 = self._lap = time.time()

    def reset(self):
        now = time.time()
        lap, self._lap = now - self._lap, now
        return lap, now - self._t0

    def total_time(self):
        return time.time() - self._t0


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


_NUM = r"\s*([^,()]+?)\s*"
_FORMS = ((re.compile(r"linear\(" + ",".join([_NUM] * 3) + r"\)"), 3),
          (re.compile(r"step_linear\(" + ",".join([_NUM] * 5) + r"\)"), 5))


def _ramp(a, b, t):
    """a -> b as t goes 0 -> 1, in the reference's floating-point form"""
    mix = np.clip(t, 0.0, 1.0)
    return (1.0 - mix) * a + mix * b


def schedule(schdl, step):
    """Scalar schedule: a float, 'linear(a,b,T)' or 'step_linear(a,b1,T1,b2,T2)' (utils.py:129-149)."""
    try:
        return float(schdl)
    except ValueError:
        pass
    for rx, n in _FORMS:
        m = rx.match(schdl)
        if m is None:
            continue
        v = [float(g) for g in m.groups()]
        if n == 3:
            return _ramp(v[0], v[1], step / v[2])
        if step <= v[2]:
            return _ramp(v[0], v[1], step / v[2])
        return _ramp(v[1], v[3], (step - v[2]) / v[4])
    raise NotImplementedError(schdl)
