"""Same-process A/B of the conv forward variants under sustained load (dev tool): each variant runs for
~0.4 s back to back, three rounds, so device-to-device and thermal differences cancel."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
lib = _lib.load(dev=True)   # -DDRQ_DEV build: python -m drqv2_amd.build --dev
lib.drq_dev_conv_variant.argtypes = [ctypes.c_int]
lib.drq_dev_conv_variant.restype = None
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 6, 7, 8]
layer = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # 1 = conv1 (9 -> 32, 84x84, stride 2), 2 = conv2
g = torch.Generator(device="cuda").manual_seed(0)
cin, hin, stride, hout = (9, 84, 2, 41) if layer == 1 else (32, 41, 1, 39)
w = torch.randn(32, cin, 3, 3, device="cuda", generator=g) * 0.1
b = torch.randn(32, device="cuda", generator=g) * 0.1
x = torch.randn(nb, cin, hin, hin, device="cuda", generator=g)
fl = nb * hout * hout * 32 * cin * 9 * 2
n = max(20, int(0.4e6 / (nb * 0.27)))
for rnd in range(3):
    for v in variants:
        lib.drq_dev_conv_variant(v)
        for _ in range(3):
            ops.conv3x3_fwd(x, w, b, stride)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.conv3x3_fwd(x, w, b, stride)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"round {rnd} variant {v:2d} nb={nb}: {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s", flush=True)
