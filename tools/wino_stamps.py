"""Per-wave time stamps of one conv2 forward launch of conv3x3_wino_kernel (development build): prologue, cycles per
unit, in-kernel clock.  Usage: python tools/wino_stamps.py [nb] [variant 8|15]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import _lib
lib = _lib.load(dev=True)
from drqv2_amd import ops
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = [int(v) for v in sys.argv[2:]] or [8, 15]
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(32, 32, 3, 3, device="cuda", generator=g) * 0.1
b = torch.randn(32, device="cuda", generator=g) * 0.1
x = torch.randn(nb, 32, 41, 41, device="cuda", generator=g)
lib.drq_dev_wino_stamps.argtypes = [ctypes.c_void_p]
lib.drq_dev_wino_variant.argtypes = [ctypes.c_int]
for variant in variants:
    st = torch.zeros(512 * 4 * 32, dtype=torch.int64, device="cuda")
    lib.drq_dev_wino_variant(0)
    for _ in range(8):
        ops.conv3x3_fwd(x, w, b, 1, wino=True)
    torch.cuda.synchronize()
    lib.drq_dev_wino_stamps(ctypes.c_void_p(st.data_ptr()))
    lib.drq_dev_wino_variant(variant)
    for _ in range(3):
        ops.conv3x3_fwd(x, w, b, 1, wino=True)
    torch.cuda.synchronize()
    lib.drq_dev_wino_variant(0)
    s = st.cpu().numpy().reshape(512, 4, 32)
    t0r, t1r = s[:, :, 30].min(), s[:, :, 29].max()
    print(f"--- variant {variant}, nb={nb}: wall first start -> last end {(t1r - t0r) / 100.0:.1f} us")
    start_r, end_r = (s[:, :, 30] - t0r) / 100.0, (s[:, :, 29] - t0r) / 100.0
    print(f"wave start us: median {np.median(start_r):.2f} max {start_r.max():.2f}; end us: min {end_r.min():.1f} median {np.median(end_r):.1f} max {end_r.max():.1f}")
    c = s[:, :, :28].astype(np.float64)
    pro = c[:, :, 1] - c[:, :, 0]
    print(f"prologue cycles: median {np.median(pro):.0f} max {pro.max():.0f}")
    nun = (c[:, :, 2:] > 0).sum(axis=2)
    print(f"units per wave: min {nun.min()} max {nun.max()}")
    n = int(nun.min())
    per = np.diff(c[:, :, 1:2 + n], axis=2)
    print("cycles per unit by iteration (median over waves):", np.median(per, axis=(0, 1)).round(0))
    print(f"cycles per unit: p10 {np.percentile(per, 10):.0f} median {np.median(per):.0f} p90 {np.percentile(per, 90):.0f}")
    dur = (s[:, :, 29] - s[:, :, 30]).astype(np.float64) / 100.0
    last = np.take_along_axis(c, (1 + nun)[:, :, None], axis=2)[:, :, 0]
    clk = (last - c[:, :, 0]) / dur
    print(f"in-kernel clock: median {np.median(clk):.0f} MHz (min {clk.min():.0f} max {clk.max():.0f}); wave duration median {np.median(dur):.1f} us")
    hw = s[:, :, 31]
    slot = hw & 0xF
    print("wave slot ids seen:", sorted(set(slot.flatten().tolist())))
    xcc = s[:, 0, 28] & 0xF
    cu = (hw[:, 0] >> 8) & 0xF; sh = (hw[:, 0] >> 12) & 1; se = (hw[:, 0] >> 13) & 0x7
    place = {}
    for blk in range(512):
        place.setdefault((int(xcc[blk]), int(se[blk]), int(sh[blk]), int(cu[blk])), []).append(blk)
    print("CUs used:", len(place), " workgroups per CU:", sorted(set(len(v) for v in place.values())))
    print("first CUs -> workgroups:", [v for k, v in sorted(place.items())[:12]])
    diffs = [v[1] - v[0] for v in place.values() if len(v) == 2]
    print("blockIdx difference of the two workgroups of a CU: ", sorted(set(diffs))[:20])
    print("workgroup -> xcc of blocks 0..23:", xcc[:24].tolist())
    simd = (hw >> 4) & 3
    print("simd of waves 0..3 of block 0..3:", simd[:4].tolist())
    tot = nun.astype(np.float64)
    per_simd = {}
    for blk in range(512):
        key = (int(xcc[blk]), int(se[blk]), int(sh[blk]), int(cu[blk]))
        for wv in range(4):
            per_simd[key + (int(simd[blk, wv]),)] = per_simd.get(key + (int(simd[blk, wv]),), 0) + int(nun[blk, wv])
    vals = np.array(list(per_simd.values()))
    print("units per SIMD: min", vals.min(), "max", vals.max(), "histogram", {int(v): int((vals == v).sum()) for v in sorted(set(vals.tolist()))})
