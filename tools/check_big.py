"""Conv forward / dgrad / wgrad at production batch sizes against torch's own GPU convolution in fp64
(structural check of the multi-tile / multi-row paths; dev tool)."""
import os, sys
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
nerr = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for (cin, hin, stride) in ((9, 84, 2), (32, 41, 1), (32, 39, 1), (32, 37, 1)):
    hout = (hin - 3) // stride + 1
    for nb in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (96, 256, 512):
        x = rn(nb, cin, hin, hin); w = rn(32, cin, 3, 3) * 0.1; b = rn(32) * 0.1
        y = ops.conv3x3_fwd(x, w, b, stride)
        ref = torch.relu(Fn.conv2d(x.double(), w.double(), b.double(), stride=stride))
        e_f = nerr(y, ref)
        dy_pad = torch.zeros(nb, 32, hout + 4, hout + 4, device="cuda")
        dy = rn(nb, 32, hout, hout)
        dy_pad[:, :, 2:-2, 2:-2] = dy
        dw, db = ops.conv3x3_wgrad(x, dy_pad[:, :, 2:-2, 2:-2], stride)
        xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
        out = Fn.conv2d(xd, wd, bd, stride=stride)
        gx, gw, gb = torch.autograd.grad(out, (xd, wd, bd), dy.double())
        e_w, e_b = nerr(dw, gw), nerr(db, gb)
        e_d = float("nan")
        if stride == 1:
            mask = rn(nb, 32, hin, hin)
            dx = ops.conv3x3_dgrad(dy_pad, w, mask)
            e_d = nerr(dx, gx * (mask.double() > 0))
        print(f"cin={cin:2d} hin={hin} nb={nb:4d}: fwd {e_f:.2e}  dgrad {e_d:.2e}  wgrad {e_w:.2e}  bgrad {e_b:.2e}", flush=True)
