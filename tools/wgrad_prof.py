"""conv2 wgrad a few times: target for rocprofv3 --pmc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
B = 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = rn(B, 32, 41, 41)
dyp = torch.zeros(B, 32, 43, 43, device="cuda"); dyp[:, :, 2:-2, 2:-2] = rn(B, 32, 39, 39)
for _ in range(5):
    ops.conv3x3_wgrad(x, dyp[:, :, 2:-2, 2:-2], 1)
torch.cuda.synchronize()
