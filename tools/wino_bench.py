"""Direct (conv3x3_kernel) against Winograd (conv3x3_wino_kernel) forms of the 32->32 layers, in isolation, at the
update's shapes: forward on 2B frames, dgrad on B frames.  Usage: python tools/wino_bench.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


w = rn(32, 32, 3, 3) * 0.1
b = rn(32) * 0.1
for hin in (41, 39, 37):
    x = rn(2 * B, 32, hin, hin).clamp_min(0)
    yd = ops.conv3x3_fwd(x, w, b, 1)
    yw = ops.conv3x3_fwd(x, w, b, 1, wino=True)
    err = float((yd - yw).norm() / yd.norm())
    td = timeit(lambda: ops.conv3x3_fwd(x, w, b, 1))
    tw = timeit(lambda: ops.conv3x3_fwd(x, w, b, 1, wino=True))
    gf = 2 * B * (hin - 2) ** 2 * 18432 / 1e9
    print(f"fwd   hin {hin} nb {2*B}: direct {td:7.1f} us ({gf/td*1e3:6.1f} TF)  wino {tw:7.1f} us ({gf/tw*1e3:6.1f} TF-equiv)  "
          f"x{td/tw:.2f}  diff {err:.1e}", flush=True)
for hout in (35, 37, 39):
    hin = hout + 2
    dyp = torch.zeros(B, 32, hout + 4, hout + 4, device="cuda")
    dyp[:, :, 2:-2, 2:-2] = rn(B, 32, hout, hout)
    mask = rn(B, 32, hin, hin)
    dd = ops.conv3x3_dgrad(dyp, w, mask)
    dw = ops.conv3x3_dgrad(dyp, w, mask, wino=True)
    err = float((dd - dw).norm() / dd.norm())
    td = timeit(lambda: ops.conv3x3_dgrad(dyp, w, mask))
    tw = timeit(lambda: ops.conv3x3_dgrad(dyp, w, mask, wino=True))
    gf = B * hin ** 2 * 18432 / 1e9
    print(f"dgrad hout {hout} nb {B}: direct {td:7.1f} us ({gf/td*1e3:6.1f} TF)  wino {tw:7.1f} us ({gf/tw*1e3:6.1f} TF-equiv)  "
          f"x{td/tw:.2f}  diff {err:.1e}", flush=True)
for hin in (37, 39, 41):
    hout = hin - 2
    x = rn(B, 32, hin, hin).clamp_min(0)
    dyp = torch.zeros(B, 32, hout + 4, hout + 4, device="cuda")
    dyp[:, :, 2:-2, 2:-2] = rn(B, 32, hout, hout)
    dyv = dyp[:, :, 2:-2, 2:-2]
    wd, bd = ops.conv3x3_wgrad(x, dyv, 1)
    ww, bw = ops.conv3x3_wgrad(x, dyv, 1, wino=True)
    err = float((wd - ww).norm() / wd.norm())
    td = timeit(lambda: ops.conv3x3_wgrad(x, dyv, 1))
    tw = timeit(lambda: ops.conv3x3_wgrad(x, dyv, 1, wino=True))
    gf = B * hout ** 2 * 18432 / 1e9
    print(f"wgrad hin {hin} nb {B}: direct {td:7.1f} us ({gf/td*1e3:6.1f} TF)  wino {tw:7.1f} us ({gf/tw*1e3:6.1f} TF-equiv)  "
          f"x{td/tw:.2f}  diff {err:.1e}  (both incl. their record reduction)", flush=True)
