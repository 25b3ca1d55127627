"""Runs conv2 fwd (2B), conv3 dgrad (B) and conv2 wgrad (B) a few times: target for rocprofv3 --pmc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
B = 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
x2 = rn(2 * B, 32, 41, 41); w = rn(32, 32, 3, 3) * 0.1; b = rn(32) * 0.1
x = x2[:B].contiguous()
dyp = torch.zeros(B, 32, 43, 43, device="cuda"); dyp[:, :, 2:-2, 2:-2] = rn(B, 32, 39, 39)
dyp41 = torch.zeros(B, 32, 41, 41, device="cuda"); dyp41[:, :, 2:-2, 2:-2] = rn(B, 32, 37, 37)
mask = rn(B, 32, 39, 39)
for _ in range(5):
    ops.conv3x3_fwd(x2, w, b, 1)
    ops.conv3x3_dgrad(dyp41, w, mask)
    ops.conv3x3_wgrad(x, dyp[:, :, 2:-2, 2:-2], 1)
torch.cuda.synchronize()
