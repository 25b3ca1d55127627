"""Per-wave time stamps of one Winograd weight-gradient launch (conv3x3_wgrad_wino_kernel<41>, development build):
where a wave's time goes -- first patches, the step loop, the G transform into LDS, the wait at the barrier, the
four-wave sum + record.  Usage: python tools/wwino_stamps.py [nb]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops, _lib
lib = _lib.load(dev=True)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
dyp = torch.zeros(nb, 32, 43, 43, device="cuda")
dyp[:, :, 2:41, 2:41] = torch.randn(nb, 32, 39, 39, device="cuda", generator=g)
dy = dyp[:, :, 2:41, 2:41]
ncu = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.zeros(ncu * 4 * 8, dtype=torch.int64, device="cuda")
lib.drq_dev_wgrad_wino_stamps.argtypes = [ctypes.c_void_p]
lib.drq_dev_wgrad_wino_stamps.restype = None
for _ in range(5):
    ops.conv3x3_wgrad(x, dy, 1, wino=True)
torch.cuda.synchronize()
lib.drq_dev_wgrad_wino_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv3x3_wgrad(x, dy, 1, wino=True)
torch.cuda.synchronize()
lib.drq_dev_wgrad_wino_stamps(None)
s = st.cpu().numpy().reshape(ncu, 4, 8)
t0 = s[:, :, 6].min()
start, end = (s[:, :, 6] - t0) / 100.0, (s[:, :, 7] - t0) / 100.0
print(f"nb={nb}: first wave start -> last wave end {end.max():.1f} us; wave start median {np.median(start):.2f} max "
      f"{start.max():.2f}; wave end median {np.median(end):.1f} min {end.min():.1f} max {end.max():.1f}")
c = s[:, :, :6].astype(np.float64)
d = np.diff(c, axis=2)
dur = (s[:, :, 7] - s[:, :, 6]) / 100.0
clk = (c[:, :, 5] - c[:, :, 0]) / dur
print(f"in-kernel clock median {np.median(clk):.0f} MHz")
for i, nm in enumerate(["first patches issued", "step loop", "G transform -> LDS", "wait at barrier", "4-wave sum + record"]):
    v = d[:, :, i]
    print(f"  {nm:24s} median {np.median(v):9.0f} cycles = {np.median(v) / np.median(clk):6.2f} us   (p10 {np.percentile(v, 10):.0f}, p90 {np.percentile(v, 90):.0f})")
steps = (nb * 400 + 3) // 4
print(f"steps per wave {steps / (ncu * 4):.2f}; ideal loop cycles at 64 MFMA x 32 = {steps / (ncu * 4) * 2048:.0f}")
