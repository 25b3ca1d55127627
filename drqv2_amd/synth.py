"""Deterministic synthetic replay batches and weights (numpy RandomState only, so
the same bytes come out on every machine).  Used by bench.py, the tests and the
golden-fixture generator; shapes follow the batch contract of the reference
(replay_buffer.py:142-160 -> drqv2.py:236-238) and SURVEY.md section 8(d)."""
import math
from collections import OrderedDict

import numpy as np
import torch

REPR_DIM = 32 * 35 * 35


def make_batch(B, A, C=9, seed=0, smooth=True, nstep=3, gamma=0.99):
    """Returns (obs u8 [B,C,84,84], action f32 [B,A], reward f32 [B,1],
    discount f32 [B,1], next_obs u8) as CPU torch tensors."""
    rs = np.random.RandomState(seed)

    def frames():
        if not smooth:
            return rs.randint(0, 256, (B, C, 84, 84)).astype(np.uint8)
        yy, xx = np.meshgrid(np.arange(84.0), np.arange(84.0), indexing="ij")
        ph = rs.uniform(0, 2 * np.pi, (B, C, 1, 1))
        ps = rs.uniform(0, 2 * np.pi, (B, C, 1, 1))
        img = 127 + 100 * np.sin(xx / 9 + ph) * np.cos(yy / 7 + ps) + 5 * rs.standard_normal((B, C, 84, 84))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    obs = frames()
    next_obs = frames()
    action = rs.uniform(-1, 1, (B, A)).astype(np.float32)
    reward = rs.standard_normal((B, 1)).astype(np.float32)
    discount = np.full((B, 1), gamma ** nstep, dtype=np.float32)
    t = torch.from_numpy
    return t(obs), t(action), t(reward), t(discount), t(next_obs)


def make_draws(B, A, seed=0, pad=4):
    """The four random draws of one update as explicit tensors:
    shifts are integer pairs (x,y) in [0,2*pad], noises are standard normal."""
    rs = np.random.RandomState(1000 + seed)
    sh_o = torch.from_numpy(rs.randint(0, 2 * pad + 1, (B, 2)).astype(np.int32))
    sh_n = torch.from_numpy(rs.randint(0, 2 * pad + 1, (B, 2)).astype(np.int32))
    n_c = torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32))
    n_a = torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32))
    return sh_o, sh_n, n_c, n_a


def make_weights(C, A, F, H, seed=0):
    """(encoder, actor, critic) OrderedDicts with the reference state_dict keys
    (drqv2.py:55-59,74-81,100-111).  Gaussian entries scaled like an orthogonal
    init (utils.py:52-61), small non-zero biases so bias paths are exercised."""
    rs = np.random.RandomState(7000 + seed)

    def mat(shape, gain=1.0):
        rows = shape[0]
        cols = int(np.prod(shape[1:]))
        w = rs.standard_normal(shape) * (gain / math.sqrt(max(rows, cols)))
        return torch.from_numpy(w.astype(np.float32))

    def vec(n, scale=0.01):
        return torch.from_numpy((rs.standard_normal(n) * scale).astype(np.float32))

    enc = OrderedDict()
    cin = C
    for i in (0, 2, 4, 6):
        enc[f"convnet.{i}.weight"] = mat((32, cin, 3, 3), math.sqrt(2.0))
        enc[f"convnet.{i}.bias"] = vec(32)
        cin = 32

    def head(prefix_dims):
        d = OrderedDict()
        d["trunk.0.weight"] = mat((F, REPR_DIM))
        d["trunk.0.bias"] = vec(F)
        d["trunk.1.weight"] = torch.from_numpy((1.0 + 0.05 * rs.standard_normal(F)).astype(np.float32))
        d["trunk.1.bias"] = vec(F)
        for prefix, dims in prefix_dims:
            for i, (o, k) in zip((0, 2, 4), dims):
                d[f"{prefix}.{i}.weight"] = mat((o, k))
                d[f"{prefix}.{i}.bias"] = vec(o)
        return d

    actor = head([("policy", [(H, F), (H, H), (A, H)])])
    critic = head([("Q1", [(H, F + A), (H, H), (1, H)]),
                   ("Q2", [(H, F + A), (H, H), (1, H)])])
    return enc, actor, critic
