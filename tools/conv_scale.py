"""conv2 forward at growing batch: separates fixed per-launch cost from the per-tile rate (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1
b = torch.randn(32, device="cuda", generator=g) * 0.1
for nb in (128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
    for _ in range(3):
        ops.conv3x3_fwd(x, w, b, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv3x3_fwd(x, w, b, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = nb * 39 * 39 * 32 * 32 * 9 * 2
    print(f"variant={os.environ.get('DRQ_CONV_VARIANT','0')} nb={nb:5d} {us:8.1f} us {fl/us/1e6:7.1f} TFLOP/s", flush=True)
    del x
