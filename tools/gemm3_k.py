#!/usr/bin/env python3
"""gemm3 forward time against K (slope = per-k-tile cost, intercept = fixed cost of a launch).  Dev tool."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib
from drqv2_amd.ops import _ptr_array, _stream
from gemm3_bench import timeit

lib = _lib.load(dev=bool(os.environ.get('DRQ_G3_DBG')))
dev = "cuda"
st = _stream()
n, M, N = 4, 256, 1024
for K in (64, 256, 512, 1024, 2048, 4096):
    xs = [torch.randn(M, K, device=dev) for _ in range(n)]
    ws = [torch.randn(N, K, device=dev) / 32 for _ in range(n)]
    ys = [torch.empty(M, N, device=dev) for _ in range(n)]
    X, W, Y = map(_ptr_array, (xs, ws, ys))
    nq = ctypes.c_int(0)
    t = timeit(lambda: lib.drq_mlp_fwd(n, X, K, W, K, Y, N, M, N, K, None, 1, None, None, ctypes.byref(nq), st))
    wsb = torch.empty(16 * 1024 * 1024, device=dev)
    t2 = timeit(lambda: lib.drq_gemm_batched_f32(n, X, K, 1, W, K, 1, Y, N, M, N, K, None, 1, None, 0, None, 0, 0, 0, wsb.data_ptr(), wsb.numel() * 4, st))
    print(f"K={K}: gemm3 {t:7.2f} us  ({t/(K/32)*1000:6.1f} ns per k-tile incl. fixed)   old {t2:7.2f} us", flush=True)
