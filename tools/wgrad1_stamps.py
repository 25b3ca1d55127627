"""Per-wave time stamps of one conv1 weight-gradient launch (conv3x3_wgrad_kernel<9,84,2>, dev tool)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
lib = _lib.load(dev=True)   # -DDRQ_DEV build: python -m drqv2_amd.build --dev
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(nb, 9, 84, 84, device="cuda", generator=g)
dy = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nblk = 2 * ncu
st = torch.zeros(nblk * 4 * 32, dtype=torch.int64, device="cuda")
lib.drq_dev_wgrad1_stamps.argtypes = [ctypes.c_void_p]
lib.drq_dev_wgrad1_stamps.restype = None
for _ in range(5):
    ops.conv3x3_wgrad(x, dy, 2)
torch.cuda.synchronize()
lib.drq_dev_wgrad1_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv3x3_wgrad(x, dy, 2)
torch.cuda.synchronize()
lib.drq_dev_wgrad1_stamps(None)
s = st.cpu().numpy().reshape(nblk, 4, 32)
t0 = s[:, :, 30].min()
start = (s[:, :, 30] - t0) / 100.0
end = (s[:, :, 29] - t0) / 100.0
print(f"nb={nb}: first wave start -> last wave end {end.max():.1f} us; wave start median {np.median(start):.2f} max {start.max():.2f};"
      f" wave end median {np.median(end):.1f} min {end.min():.1f}")
n = s[:, :, 28]
c = s[:, :, :28].astype(np.float64)
dur = (s[:, :, 29] - s[:, :, 30]) / 100.0
for k in sorted(set(((n - 5) // 2).flatten().tolist())):
    sel = ((n - 5) // 2) == k
    cs = c[sel]
    pro = cs[:, 1] - cs[:, 0]
    ustart = cs[:, 2:2 + 2 * k:2]                     # unit start
    staged = cs[:, 3:3 + 2 * k:2]                     # after write_lds (+ lgkmcnt(0))
    uend = np.concatenate([ustart[:, 1:], cs[:, 2 + 2 * k:3 + 2 * k]], axis=1)
    stage = staged - ustart
    mf = uend - staged
    bar = cs[:, 3 + 2 * k] - cs[:, 2 + 2 * k]
    red = cs[:, 4 + 2 * k] - cs[:, 3 + 2 * k]
    tot = cs[:, 4 + 2 * k] - cs[:, 0]
    clk = tot / dur[sel]
    print(f"waves with {k} rows: {sel.sum()}  clock median {np.median(clk):.0f} MHz")
    print(f"  prologue {np.median(pro):.0f}  stage (wait loads + LDS writes) median {np.median(stage):.0f} p90 {np.percentile(stage, 90):.0f}"
          f"  MFMA loop median {np.median(mf):.0f} p90 {np.percentile(mf, 90):.0f}  wait-at-barrier {np.median(bar):.0f}  reduce {np.median(red):.0f}"
          f"  total {np.median(tot):.0f} cycles")
print("ideal MFMA cycles per row: 21 steps x 3 x 64 =", 21 * 3 * 64, "(two waves share a SIMD)")
