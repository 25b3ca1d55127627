"""Per-wave time stamps of one conv2 weight-gradient launch (rolling-tile kernel, dev tool)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
lib = _lib.load(dev=True)   # -DDRQ_DEV build: python -m drqv2_amd.build --dev
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
dy = torch.randn(nb, 32, 39, 39, device="cuda", generator=g)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.zeros(ncu * 4 * 32, dtype=torch.int64, device="cuda")
lib.drq_dev_wgrad_stamps.argtypes = [ctypes.c_void_p]
lib.drq_dev_wgrad_stamps.restype = None
for _ in range(5):
    ops.conv3x3_wgrad(x, dy, 1)
torch.cuda.synchronize()
lib.drq_dev_wgrad_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv3x3_wgrad(x, dy, 1)
torch.cuda.synchronize()
lib.drq_dev_wgrad_stamps(None)
s = st.cpu().numpy().reshape(ncu, 4, 32)
t0 = s[:, :, 30].min()
start = (s[:, :, 30] - t0) / 100.0
end = (s[:, :, 29] - t0) / 100.0
print(f"nb={nb}: first wave start -> last wave end {end.max():.1f} us; wave start median {np.median(start):.2f} max {start.max():.2f};"
      f" wave end median {np.median(end):.1f} min {end.min():.1f}")
n = s[:, :, 28]
c = s[:, :, :28].astype(np.float64)
dur = (s[:, :, 29] - s[:, :, 30]) / 100.0
for units in sorted(set((n - 5).flatten().tolist())):
    sel = (n - 5) == units
    cs = c[sel]
    k = units
    pro = cs[:, 1] - cs[:, 0]
    rows = np.diff(cs[:, 2:3 + k], axis=1)          # row i = stamp[2+i] -> stamp[3+i] (the last ends at the post-loop mark)
    bar = cs[:, 3 + k] - cs[:, 2 + k]
    red = cs[:, 4 + k] - cs[:, 3 + k]
    tot = cs[:, 4 + k] - cs[:, 0]
    clk = tot / dur[sel]
    print(f"waves with {k} rows: {sel.sum()}  clock median {np.median(clk):.0f} MHz")
    print(f"  prologue {np.median(pro):.0f}  row median {np.median(rows):.0f} p10 {np.percentile(rows, 10):.0f} p90 {np.percentile(rows, 90):.0f} max {rows.max():.0f}"
          f"  wait-at-barrier {np.median(bar):.0f}  reduce {np.median(red):.0f}  total {np.median(tot):.0f} cycles")
    print("  row cycles by position (median):", np.median(rows, axis=0).round(0))
print("ideal MFMA cycles per row: 20 steps x 9 x 64 =", 20 * 9 * 64)
