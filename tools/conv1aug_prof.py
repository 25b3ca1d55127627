"""Runs the fused aug+conv1 kernel a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drqv2_amd import _lib, ops
from drqv2_amd._lib import ptr
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(0)
obs = torch.randint(0, 256, (B, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
obs1 = torch.randint(0, 256, (B, 9, 84, 84), generator=g, dtype=torch.uint8).cuda()
sh = torch.randint(0, 9, (B, 2), generator=g).float().cuda()
sh1 = torch.randint(0, 9, (B, 2), generator=g).float().cuda()
w = (torch.randn(32, 9, 3, 3, generator=g) * 0.2).cuda()
b = (torch.randn(32, generator=g) * 0.1).cuda()
base = ops.aug_base_grid(84, 4, "cuda")
y = torch.empty((2 * B, 32, 41, 41), device="cuda")
xaug = torch.empty((2 * B, 9, 84, 84), device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    assert lib.drq_conv1_aug_fwd(ptr(obs), ptr(sh), ptr(obs1), ptr(sh1), ptr(base), ptr(w), ptr(b), ptr(xaug), ptr(y), B, B, st) == 0
torch.cuda.synchronize()
