"""aug kernel alone, for rocprofv3 (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, synth
obs = synth.make_batch(256, 6, 9, seed=0, smooth=True)[0].cuda()
sh = torch.randint(0, 9, (256, 2), device="cuda").float()
for _ in range(30):
    y = ops.random_shifts_aug(obs, sh, fuse_norm=True)
torch.cuda.synchronize()
