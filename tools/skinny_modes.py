"""Trunk dgrad (skinny kernel): padded-scatter output vs dense output, with and without the ReLU mask (dev tool).
(gemm_batched allocates its 64 MB workspace per call; the numbers are for comparing the modes.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
B, F, hw = 256, 50, 35
N = 32 * hw * hw
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
dz, w, feat = rn(B, F), rn(F, N) * 0.1, rn(B, N)
pad = torch.zeros(B, 32, hw + 4, hw + 4, device="cuda")
dense = torch.zeros(B, N, device="cuda")


def timeit(f, n=100):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


a = ([dz], True, [w], False, B, N, F, F, N)
print(f"scatter + mask : {timeit(lambda: ops.gemm_batched(*a, auxs=[feat], scatter_hw=hw, Cs=[pad])):6.1f} us")
print(f"dense   + mask : {timeit(lambda: ops.gemm_batched(*a, auxs=[feat], Cs=[dense])):6.1f} us")
print(f"scatter no mask: {timeit(lambda: ops.gemm_batched(*a, scatter_hw=hw, Cs=[pad])):6.1f} us")
print(f"dense   no mask: {timeit(lambda: ops.gemm_batched(*a, Cs=[dense])):6.1f} us")
