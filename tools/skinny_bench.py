"""Trunk backward GEMMs: automatic dispatch (dedicated dgrad kernel) vs the generic tiled path (dev tool)."""
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


def timeit(f, n=200):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, kw in (("skinny", {}), ("generic", dict(tile=2, splitk=1))):
    t = timeit(lambda: ops.gemm_batched([dz], True, [w], False, B, N, F, F, N, auxs=[feat], scatter_hw=hw, Cs=[pad], **kw))
    print(f"trunk dgrad {name:8s} {t:7.1f} us  ({(2*B*N*4 + F*N*4)/t/1e6:5.2f} TB/s algorithmic)", flush=True)
for name, kw in (("skinny", {}), ("generic", dict(tile=2))):
    t = timeit(lambda: ops.gemm_batched([dz], False, [feat], False, F, N, B, F, N, rowsum=True, **kw))
    print(f"trunk wgrad {name:8s} {t:7.1f} us  ({(B*N*4 + F*N*4)/t/1e6:5.2f} TB/s algorithmic)", flush=True)
