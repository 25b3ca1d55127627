"""Runs the batched MLP GEMMs (fwd NT tile1, wgrad TN tile2, dgrad NN) a few times: target for rocprofv3 --pmc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib
from drqv2_amd._lib import ptr
lib = _lib.load()
B, H = 256, 1024
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, w, dy = rn(2, B, H), rn(2, H, H), rn(2, B, H)
C = torch.empty(2, B, H, device="cuda"); W = torch.empty(2, H, H, device="cuda")
ws = torch.empty(16 * 1024 * 1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def gemm(A, lda, akc, Bm, ldb, bkc, Cm, ldc, M, N, K, abs_, bbs, cbs, tile, sk):
    rc = lib.drq_gemm_f32(ptr(A), lda, akc, ptr(Bm), ldb, bkc, ptr(Cm), ldc, M, N, K, 2, abs_, bbs, cbs, None, 0, 0, None, 0, 0, 0, tile, sk, ptr(ws), ws.numel() * 4, st)
    assert rc == 0
for _ in range(5):
    gemm(x, H, 1, w, H, 1, C, H, B, H, H, B * H, H * H, B * H, 1, 1)      # fwd  (tile 32)
    gemm(x, H, 1, w, H, 1, C, H, B, H, H, B * H, H * H, B * H, 2, 1)      # fwd  (tile 64)
    gemm(dy, H, 0, x, H, 0, W, H, H, H, B, B * H, B * H, H * H, 2, 1)     # wgrad (tile 64)
    gemm(dy, H, 1, w, H, 0, C, H, B, H, H, B * H, H * H, B * H, 1, 1)     # dgrad (tile 32)
torch.cuda.synchronize()
