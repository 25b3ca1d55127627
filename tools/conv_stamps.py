"""Per-wave time stamps of one conv2 forward launch (DRQ_CONV_VARIANT=10, dev tool)."""
import ctypes, os, sys
os.environ.setdefault("DRQ_CONV_VARIANT", "10")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
lib = _lib.load(dev=True)   # -DDRQ_DEV build: python -m drqv2_amd.build --dev
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1
b = torch.randn(32, device="cuda", generator=g) * 0.1
x = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
st = torch.zeros(512 * 8 * 32, dtype=torch.int64, device="cuda")
lib.drq_dev_conv_stamps.argtypes = [ctypes.c_void_p]
lib.drq_dev_conv_stamps.restype = None
for _ in range(5):
    ops.conv3x3_fwd(x, w, b, 1)
torch.cuda.synchronize()
lib.drq_dev_conv_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv3x3_fwd(x, w, b, 1)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(512, 8, 32)
t0r = s[:, :, 30].min(); t1r = s[:, :, 29].max()
print(f"nb={nb}: wall (memrealtime) first wave start -> last wave end: {(t1r - t0r) / 100.0:.1f} us")
start_r = (s[:, :, 30] - t0r) / 100.0
end_r = (s[:, :, 29] - t0r) / 100.0
print(f"wave start us: min {start_r.min():.2f} median {np.median(start_r):.2f} max {start_r.max():.2f}")
print(f"wave end   us: min {end_r.min():.2f} median {np.median(end_r):.2f} max {end_r.max():.2f}")
c = s[:, :, :29].astype(np.float64)
ntile = int((c[0, 0] > 0).sum()) - 3
if os.environ["DRQ_CONV_VARIANT"] != "11":
    dur_r = (s[:, :, 29] - s[:, :, 30]).astype(np.float64) / 100.0
    clk = (c[:, :, 2] - c[:, :, 0]) / dur_r
    ok = dur_r > 0.5 * np.median(dur_r)
    print(f"variant {os.environ['DRQ_CONV_VARIANT']}: in-kernel clock median {np.median(clk[ok]):.0f} MHz (min {clk[ok].min():.0f} max {clk[ok].max():.0f});"
          f" wave duration median {np.median(dur_r):.1f} us; wave cycles median {np.median((c[:, :, 2] - c[:, :, 0])[ok]):.0f}")
    sys.exit(0)
pro = c[:, :, 1] - c[:, :, 0]
print(f"prologue cycles (start -> after barrier): median {np.median(pro):.0f} max {pro.max():.0f}")
tiles = np.diff(c[:, :, 1:2 + ntile], axis=2)
print(f"tiles per wave: {ntile}; per-tile cycles by iteration (median over waves):", np.median(tiles, axis=(0, 1)).round(0))
print(f"per-tile cycles overall: p10 {np.percentile(tiles, 10):.0f} median {np.median(tiles):.0f} p90 {np.percentile(tiles, 90):.0f} max {tiles.max():.0f}")
tot = c[:, :, 2 + ntile] - c[:, :, 0]
print(f"wave total cycles: median {np.median(tot):.0f} max {tot.max():.0f}; clock = {np.median(tot) / (np.median(end_r - start_r)):.0f} cycles/us")
last = 2 + ntile
dur_r = (s[:, :, 29] - s[:, :, 30]).astype(np.float64) / 100.0          # us per wave
full = (c[:, :, last] > 0) & (c[:, :, 1 + ntile] > 0)
clk = (c[:, :, last] - c[:, :, 0])[full] / dur_r[full]
print(f"in-kernel clock per wave (memtime cycles / memrealtime us): median {np.median(clk):.0f} min {clk.min():.0f} max {clk.max():.0f} MHz")
e = end_r.copy()
print("end time by wave id within block (median over blocks):", np.median(e, axis=0).round(1))
print("block end (max over its waves) percentiles 10/50/90/100:", np.percentile(e.max(axis=1), [10, 50, 90, 100]).round(1))
print("block end - block first-wave end (spread inside a block), median:", np.median(e.max(axis=1) - e.min(axis=1)).round(1))
print("blocks 0..15 end:", e.max(axis=1)[:16].round(0))
hw = s[:, :, 31]
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 0x3
print("distinct (se,sh,cu) ids:", len(set(zip(se.flatten().tolist(), sh.flatten().tolist(), cu.flatten().tolist()))))
