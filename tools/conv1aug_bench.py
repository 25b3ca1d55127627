"""Fused aug+conv1 against the two-kernel path, in isolation (B frames per view)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drqv2_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(0)
obs = torch.randint(0, 256, (B, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
obs1 = torch.randint(0, 256, (B, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
sh = torch.randint(0, 9, (B, 2), generator=g).float().cuda()
sh1 = torch.randint(0, 9, (B, 2), generator=g).float().cuda()
w = (torch.randn(32, 9, 3, 3, generator=g) * 0.2).cuda()
b = (torch.randn(32, generator=g) * 0.1).cuda()
both = torch.cat([obs, obs1]); shb = torch.cat([sh, sh1])

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

t_f = timeit(lambda: ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b))
t_f0 = timeit(lambda: ops.conv1_aug_fwd(obs, sh, obs1, sh1, w, b, n_store=0))
def two():
    x = ops.random_shifts_aug(both, shb, 4, fuse_norm=True)
    return ops.conv3x3_fwd(x, w, b, 2)
t_2 = timeit(two)
print(f"B={B}: fused {t_f:.1f} us (incl. allocs), fused no-store {t_f0:.1f} us, two kernels {t_2:.1f} us", flush=True)
