import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drqv2_amd import ops
from oracle import drq_oracle as O
def rnd(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))
n = 10007
p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=1e-3)
for gs, mult in ((1.0, 1.0), (0.25, 4.0)):
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    ops.adam_flat(p, (g * mult).cuda(), m, v, 8e-5, 1, gscale=gs)
    pr, mr, vr = p0.clone(), torch.zeros(n), torch.zeros(n)
    O.adam_step(pr, g, mr, vr, 1, 8e-5)
    bad = (p.cpu() != pr).nonzero().flatten()
    print("gscale", gs, "mismatch m", (m.cpu() != mr).sum().item(), "v", (v.cpu() != vr).sum().item(), "p", bad.numel())
    for i in bad[:5].tolist():
        print(i, float(g[i]), float(p0[i]), float(p[i].cpu()), float(pr[i]), float(mr[i]), float(vr[i]))
    # recompute pieces with torch GPU ops
    vg = vr.cuda(); mg = mr.cuda()
    den_gpu = vg.sqrt() / torch.tensor((1 - 0.999) ** 0.5, device="cuda") + 1e-8
    den_cpu = vr.sqrt() / torch.tensor((1 - 0.999) ** 0.5) + 1e-8
    print(" denom gpu-vs-cpu mismatches", (den_gpu.cpu() != den_cpu).sum().item())
