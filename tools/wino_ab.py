"""Timing ablations of conv3x3_wino_kernel (development build): which part of the unit time is the input transform,
the output transform, the patch loads.  Usage: python tools/wino_ab.py [B]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib
lib = _lib.load(dev=True)
from drqv2_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
w, b = rn(32, 32, 3, 3) * 0.1, rn(32) * 0.1
x = rn(2 * B, 32, 41, 41).clamp_min(0)


def timeit(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


names = {0: "full", 1: "no input transform", 2: "no output transform", 3: "no transforms", 4: "no patch loads",
         7: "MFMA + LDS operand reads only"}
lib.drq_dev_wino_variant.argtypes = [ctypes.c_int]
for rep in range(2):
    for v in (0, 1, 2, 3, 4, 7):
        lib.drq_dev_wino_variant(v)
        print(f"variant {v} ({names[v]:30s}): {timeit(lambda: ops.conv3x3_fwd(x, w, b, 1, wino=True)):7.1f} us", flush=True)
lib.drq_dev_wino_variant(0)
lib.drq_dev_wino_stagger.argtypes = [ctypes.c_int]
dyp = torch.zeros(B, 32, 41, 41, device="cuda")
dyp[:, :, 2:-2, 2:-2] = rn(B, 32, 37, 37)
mask = rn(B, 32, 39, 39)
for rep in range(2):
    for sg in (0, 1, 2, 4, 6, 9, 12):
        lib.drq_dev_wino_stagger(sg)
        tf = timeit(lambda: ops.conv3x3_fwd(x, w, b, 1, wino=True))
        td = timeit(lambda: ops.conv3x3_dgrad(dyp, w, mask, wino=True))
        print(f"stagger {sg:2d} x 1024 clocks: fwd hin 41 {tf:7.1f} us   dgrad hout 37 {td:7.1f} us", flush=True)

# ---- the weight-gradient kernel (conv3x3_wgrad_wino_kernel), conv2's shape
lib.drq_dev_wgrad_wino_variant.argtypes = [ctypes.c_int]
xw = rn(B, 32, 41, 41).clamp_min(0)
dyw = torch.zeros(B, 32, 43, 43, device="cuda")
dyw[:, :, 2:-2, 2:-2] = rn(B, 32, 39, 39)
dyv = dyw[:, :, 2:-2, 2:-2]
wn = {0: "full", 1: "no patch loads", 2: "no transforms", 3: "neither (MFMAs + epilogue)"}
for rep in range(2):
    for v in (0, 1, 2, 3):
        lib.drq_dev_wgrad_wino_variant(v)
        print(f"wgrad variant {v} ({wn[v]:28s}): {timeit(lambda: ops.conv3x3_wgrad(xw, dyv, 1, wino=True)):7.1f} us (incl. the ~8 us record reduction)", flush=True)
lib.drq_dev_wgrad_wino_variant(0)
