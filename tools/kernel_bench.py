#!/usr/bin/env python3
"""Per-kernel timings (events on the launch stream) at the bench shapes.  Dev tool, GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops  # noqa: E402

B = int(os.environ.get("KB_BATCH", "256"))
H = [84, 41, 39, 37, 35]


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = "cuda"
    res = {}
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    # conv forward (2B frames), dgrad (B), wgrad (B)
    for l in range(4):
        cin = 9 if l == 0 else 32
        hin, hout = H[l], H[l + 1]
        stride = 2 if l == 0 else 1
        x2 = rn(2 * B, cin, hin, hin)
        w, b = rn(32, cin, 3, 3) * 0.1, rn(32) * 0.1
        fl = 2 * 32 * cin * 9 * hout * hout
        t = timeit(lambda: ops.conv3x3_fwd(x2, w, b, stride))
        res[f"conv{l+1}_fwd_2B"] = (t * 1e6, 2 * B * fl / t / 1e12)
        x = x2[:B].contiguous()
        dyp = torch.zeros(B, 32, hout + 4, hout + 4, device=dev)
        dyp[:, :, 2:-2, 2:-2] = rn(B, 32, hout, hout)
        t = timeit(lambda: ops.conv3x3_wgrad(x, dyp[:, :, 2:-2, 2:-2], stride))
        res[f"conv{l+1}_wgrad_B"] = (t * 1e6, B * fl / t / 1e12)
        if l >= 1:
            mask = x
            t = timeit(lambda: ops.conv3x3_dgrad(dyp, w, mask))
            res[f"conv{l+1}_dgrad_B"] = (t * 1e6, B * fl / t / 1e12)
    # aug
    obs = torch.randint(0, 256, (B, 9, 84, 84), device=dev, dtype=torch.uint8)
    sh = torch.randint(0, 9, (B, 2), device=dev).float()
    t = timeit(lambda: ops.random_shifts_aug(obs, sh, 4, fuse_norm=True))
    res["aug_B"] = (t * 1e6, B * 9 * 7056 * 5 / t / 1e12)     # TB/s in the second slot
    # linear layers
    R = 39200
    for name, (M, N, K) in {"trunk_fwd": (B, 50, R), "mlp_fwd": (B, 1024, 1024), "q1_fwd": (B, 1024, 56)}.items():
        x, w, b = rn(M, K), rn(N, K) * K ** -0.5, rn(N)
        t = timeit(lambda: ops.linear_fwd(x, w, b, relu=True))
        res[name] = (t * 1e6, 2 * M * N * K / t / 1e12)
    dy, w = rn(B, 1024), rn(1024, 1024) / 32
    t = timeit(lambda: ops.linear_dgrad(dy, w, dy))
    res["mlp_dgrad"] = (t * 1e6, 2 * B * 1024 * 1024 / t / 1e12)
    t = timeit(lambda: ops.linear_wgrad(dy, dy))
    res["mlp_wgrad(+colsum)"] = (t * 1e6, 2 * B * 1024 * 1024 / t / 1e12)
    dz, feat, wt = rn(B, 50), rn(B, R), rn(50, R) * R ** -0.5
    t = timeit(lambda: ops.linear_wgrad(dz, feat))
    res["trunk_wgrad(+colsum)"] = (t * 1e6, 2 * B * 50 * R / t / 1e12)
    t = timeit(lambda: ops.linear_dgrad(dz, wt, feat))
    res["trunk_dgrad"] = (t * 1e6, 2 * B * 50 * R / t / 1e12)
    for k, (us, tf) in res.items():
        print(f"{k:24s} {us:9.1f} us   {tf:7.2f} TFLOP/s (aug: TB/s)")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
