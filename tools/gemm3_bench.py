#!/usr/bin/env python3
"""gemm3 (LDS-DMA ring) against the kernels it replaces, at the update's hidden-layer shapes.  Dev tool, GPU box.
Raw ctypes calls on preallocated buffers (the wrappers' allocations would dominate 10-us kernels)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib  # noqa: E402
from drqv2_amd.ops import _ptr_array, _stream  # noqa: E402


def timeit(fn, reps=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(ts)


def main():
    lib = _lib.load(dev=bool(os.environ.get('DRQ_G3_DBG')))
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    H = 1024
    st = _stream()
    wsb = torch.empty(16 * 1024 * 1024, device=dev)
    shapes = ((4, 256), (2, 256), (1, 512), (4, 512), (2, 64), (4, 64), (2, 32))
    if os.environ.get('DRQ_G3_DBG'):
        shapes = ((4, 256), (4, 64), (2, 256))
    for n, M in shapes:
        xs = [rn(M, H) for _ in range(n)]
        ws = [rn(H, H) / 32 for _ in range(n)]
        bs = [rn(H) for _ in range(n)]
        qw = [rn(H) / 32 for _ in range(n)]
        ys = [torch.empty(M, H, device=dev) for _ in range(n)]
        qp = [torch.empty(M, 32, device=dev) for _ in range(n)]
        dws = [torch.empty(H, H, device=dev) for _ in range(n)]
        dbs = [torch.empty(H, device=dev) for _ in range(n)]
        X, W, Bv, Y, QW, QP, DW, DB = map(_ptr_array, (xs, ws, bs, ys, qw, qp, dws, dbs))
        nq = ctypes.c_int(0)
        fl = 2.0 * n * M * H * H
        old_f = lambda: lib.drq_gemm_batched_f32(n, X, H, 1, W, H, 1, Y, H, M, H, H, Bv, 1, None, 0, None, 0, 0, 0,
                                                 wsb.data_ptr(), wsb.numel() * 4, st)
        old_d = lambda: lib.drq_gemm_batched_f32(n, X, H, 1, W, H, 0, Y, H, M, H, H, None, 0, X, H, None, 0, 0, 0,
                                                 wsb.data_ptr(), wsb.numel() * 4, st)
        old_w = lambda: lib.drq_gemm_batched_f32(n, X, H, 0, X, H, 0, DW, H, H, H, M, None, 0, None, 0, DB, 0, 0, 0,
                                                 wsb.data_ptr(), wsb.numel() * 4, st)
        res = {}
        if M % 64 == 0:
            res["fwd3"] = timeit(lambda: lib.drq_mlp_fwd(n, X, H, W, H, Y, H, M, H, H, Bv, 1, None, None, ctypes.byref(nq), st))
            res["fwd3q"] = timeit(lambda: lib.drq_mlp_fwd(n, X, H, W, H, Y, H, M, H, H, Bv, 1, QW, QP, ctypes.byref(nq), st))
            res["dgrad3"] = timeit(lambda: lib.drq_mlp_dgrad(n, X, H, W, H, Y, H, M, H, H, X, H, st))
            res["pair3"] = timeit(lambda: lib.drq_mlp_wgrad_dgrad(n, X, H, X, H, DW, DB, W, H, Y, H, X, H, M, H, H, st))
        res["fwd_old"] = timeit(old_f)
        res["dgrad_old"] = timeit(old_d)
        res["wgrad_old"] = timeit(old_w)
        print(f"n={n} M={M}: " + "  ".join(f"{k} {v:6.1f}us({fl/v/1e6:5.1f}TF)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
