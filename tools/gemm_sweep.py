#!/usr/bin/env python3
"""(tile, splitk) sweep of drq_gemm_f32 at the shapes of the update step.  Dev tool, GPU box only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
from drqv2_amd._lib import ptr, check

B, H, R = 256, 1024, 39200
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
lib = _lib.load()
ws = torch.empty(16 * 1024 * 1024, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream


def run(A, lda, akc, Bm, ldb, bkc, C, ldc, M, N, K, nb, abs_, bbs, cbs, tile, sk):
    return lib.drq_gemm_f32(ptr(A), lda, akc, ptr(Bm), ldb, bkc, ptr(C), ldc, M, N, K, nb, abs_, bbs, cbs, None, 0, 0,
                            None, 0, 0, 0, tile, sk, ptr(ws), ws.numel() * 4, st())


def timeit(fn, reps=10):
    for _ in range(3):
        rc = fn()
        if rc != 0:
            return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = {
    # name: (A, lda, a_kc, B, ldb, b_kc, M, N, K, nbatch, a_bs, b_bs)
    "mlp_fwd_x4   M256 N1024 K1024": (rn(4, B, H), H, 1, rn(4, H, H), H, 1, B, H, H, 4, B * H, H * H),
    "mlp_fwd_x2   M256 N1024 K1024": (rn(2, B, H), H, 1, rn(2, H, H), H, 1, B, H, H, 2, B * H, H * H),
    "pol_fwd_x1   M512 N1024 K1024": (rn(2 * B, H), H, 1, rn(H, H), H, 1, 2 * B, H, H, 1, 0, 0),
    "pol_wgrad_x1 M1024 N1024 K256": (rn(B, H), H, 0, rn(B, H), H, 0, H, H, B, 1, 0, 0),
    "pol_dgrad_x1 M256 N1024 K1024": (rn(B, H), H, 1, rn(H, H), H, 0, B, H, H, 1, 0, 0),
    "mlp_fwd_x1   M256 N1024 K1024": (rn(B, H), H, 1, rn(H, H), H, 1, B, H, H, 1, 0, 0),
    "mlp_dgrad_x2 M256 N1024 K1024": (rn(2, B, H), H, 1, rn(2, H, H), H, 0, B, H, H, 2, B * H, H * H),
    "mlp_wgrad_x2 M1024 N1024 K256": (rn(2, B, H), H, 0, rn(2, B, H), H, 0, H, H, B, 2, B * H, B * H),

    "trunk_fwd_x1 M256 N50 K39200": (rn(B, R), R, 1, rn(50, R), R, 1, B, 50, R, 1, 0, 0),
    "trunk_wgrad  M50 N39200 K256": (rn(B, 50), 50, 0, rn(B, R), R, 0, 50, R, B, 1, 0, 0),
    "q1_fwd_x4    M256 N1024 K56": (rn(B, 56), 56, 1, rn(4, H, 56), 56, 1, B, H, 56, 4, 0, H * 56),
    "q1_wgrad_x2  M1024 N56 K256": (rn(2, B, H), H, 0, rn(B, 56), 56, 0, H, 56, B, 2, B * H, 0),
    "q1_dgrad_x2  M256 N56 K1024": (rn(2, B, H), H, 1, rn(2, H, 56), 56, 0, B, 56, H, 2, B * H, H * 56),
}
for name, (A, lda, akc, Bm, ldb, bkc, M, N, K, nb, abs_, bbs) in cases.items():
    C = torch.empty(nb, M, N, device=dev)
    out = []
    for tile in (1, 2, 5, 6):
        for sk in (1, 2, 4, 8, 16, 24, 48, 96):
            if sk > 1 and K // sk < 64:
                continue
            t = timeit(lambda: run(A, lda, akc, Bm, ldb, bkc, C, N, M, N, K, nb, abs_, bbs, M * N, tile, sk))
            if t is not None:
                out.append((t, tile, sk))
    t0 = timeit(lambda: run(A, lda, akc, Bm, ldb, bkc, C, N, M, N, K, nb, abs_, bbs, M * N, 0, 0))
    out.sort()
    fl = 2.0 * M * N * K * nb
    print(f"{name:34s} auto {t0:7.1f} us | best " + "  ".join(f"t{tl}/k{sk}:{t:6.1f}" for t, tl, sk in out[:5])
          + f" | ideal {fl / 157.3e12 * 1e6:5.1f} us")
