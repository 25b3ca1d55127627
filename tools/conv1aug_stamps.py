"""Per-workgroup start/end stamps and CU placement of the fused aug+conv1 kernel (development build)."""
import os, sys, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drqv2_amd import _lib
lib = _lib.load(dev=True)
from drqv2_amd import ops
from drqv2_amd._lib import ptr
for n, t in (("drq_dev_conv1aug_variant", ctypes.c_int), ("drq_dev_conv1aug_stagger", ctypes.c_int), ("drq_dev_conv1aug_stamps", ctypes.c_void_p)):
    getattr(lib, n).argtypes = [t]; getattr(lib, n).restype = None
B = 256
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
stamps = torch.zeros((512, 4), dtype=torch.int64, device="cuda")
def run():
    assert lib.drq_conv1_aug_fwd(ptr(obs), ptr(sh), ptr(obs1), ptr(sh1), ptr(base), ptr(w), ptr(b), ptr(xaug), ptr(y), B, B, st) == 0
for sg in (0, 1):
    lib.drq_dev_conv1aug_stagger(sg)
    lib.drq_dev_conv1aug_stamps(None)
    for _ in range(5): run()
    lib.drq_dev_conv1aug_stamps(ctypes.c_void_p(stamps.data_ptr()))
    run(); torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    t0 = s[:, 0].min()
    start, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0     # s_memrealtime: 100 MHz -> us
    hw = s[:, 2]
    cu = ((s[:, 3] & 0xF) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    per = collections.defaultdict(list)
    for i in range(512):
        per[int(cu[i])].append((float(start[i]), float(end[i]), int(hw[i] & 0xF), i))
    nb = collections.Counter(len(v) for v in per.values())
    print(f"stagger {sg}: {len(per)} distinct CUs; workgroups per CU histogram {dict(nb)}; kernel span {end.max():.1f} us")
    ov = 0; tot = 0
    for k, v in list(per.items()):
        v.sort()
        for a_, b_ in zip(v[:-1], v[1:]):
            tot += 1
            if b_[0] < a_[1] - 1.0: ov += 1
    print(f"   pairs on one CU that overlap in time: {ov} of {tot}")
    for k in list(per)[:6]:
        print("   CU", hex(k), [(round(a_, 1), round(b_, 1), f"slot{c_}", f"wg{d_}") for a_, b_, c_, d_ in per[k]])
