"""Timing ablations of the fused aug+conv1 kernel (development build)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drqv2_amd import _lib
lib = _lib.load(dev=True)
from drqv2_amd import ops
from drqv2_amd._lib import ptr
lib.drq_dev_conv1aug_variant.argtypes = [ctypes.c_int]
lib.drq_dev_conv1aug_variant.restype = None
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
def run(n_store):
    rc = lib.drq_conv1_aug_fwd(ptr(obs), ptr(sh), ptr(obs1), ptr(sh1), ptr(base), ptr(w), ptr(b), ptr(xaug), ptr(y), B, n_store, st)
    assert rc == 0, rc
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
lib.drq_dev_conv1aug_stagger.argtypes = [ctypes.c_int]
lib.drq_dev_conv1aug_stagger.restype = None
for sg in (0, 1, 2):
    lib.drq_dev_conv1aug_stagger(sg)
    lib.drq_dev_conv1aug_variant(0)
    print(f"B={B} stagger {sg}: full, store obs view {timeit(lambda: run(B)):7.1f} us   no store {timeit(lambda: run(0)):7.1f} us", flush=True)
lib.drq_dev_conv1aug_stagger(1)
names = {0: "full", 1: "no tiles (stage 3)", 2: "no aug (stage 2)", 3: "DMA + barriers only", 4: "no global stores in stage 2", 10: "no DMA, no aug: tiles only"}
for v in (0, 4, 1, 2, 3, 10):
    lib.drq_dev_conv1aug_variant(v)
    print(f"B={B} variant {v:2d} {names[v]:32s} store obs view {timeit(lambda: run(B)):7.1f} us   no store {timeit(lambda: run(0)):7.1f} us", flush=True)
