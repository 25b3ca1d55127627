"""Where does the B=512 whole-update gradient difference come from?  All gradients against the fp64 oracle (HIP
encoder ReLU decisions injected), and the ReLU decisions of the critic's hidden layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_step import WIDE, make_agent, make_oracle, run_hip, nerr
from oracle import drq_oracle as O

for name in sys.argv[1:] or ["quadruped_b512"]:
    cfg = WIDE[name]
    B, H = cfg["B"], cfg["H"]
    ag = make_agent(cfg)
    o64 = make_oracle(cfg, torch.float64)
    m, batch, (sh_o, sh_n, n_c, n_a) = run_hip(ag, cfg, 0)
    eng = ag._engine
    xin = eng.ws_view("AUG", B, (2 * B, 9, 84, 84)).cpu()
    hs = (41, 39, 37, 35)
    acts = [eng.ws_view(nm, B, (2 * B, 32, h, h))[:B].cpu() > 0 for nm, h in zip(("ACT1", "ACT2", "ACT3"), hs)]
    acts.append(eng.ws_view("FEAT", B, (2 * B, 32, 35, 35))[:B].cpu() > 0)
    # critic hidden decisions, from the oracle's weights BEFORE its update
    cr = {k: v.clone() for k, v in o64.critic.items()}
    m64 = o64.update(batch, cfg["step0"], sh_o, sh_n, n_c, n_a, enc_in_override=(xin[:B], xin[B:]), keep=True,
                     relu_masks=acts)
    for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"), ("actor", ag.actor, "g_actor")):
        for (pn, p), g64 in zip(mod.named_parameters(), o64.last[key].values()):
            print(f"{name} {nm:6s} {pn:22s} err {nerr(p.grad, g64):.3e}", flush=True)
    feat = o64.last["feat"]
    h = O.trunk_forward(cr, feat)
    ha = torch.cat([h, batch[1].double()], -1)
    C1 = eng.ws_view("C1", B, (2, B, H)).cpu()
    C2 = eng.ws_view("C2", B, (2, B, H)).cpu()
    for q, pre in ((0, "Q1"), (1, "Q2")):
        z1 = ha @ cr[f"{pre}.0.weight"].T + cr[f"{pre}.0.bias"]
        a1 = torch.relu(z1)
        z2 = a1 @ cr[f"{pre}.2.weight"].T + cr[f"{pre}.2.bias"]
        for nm2, z, hip in (("l1", z1, C1[q]), ("l2", z2, C2[q])):
            dis = (z > 0) != (hip > 0)
            print(f"{name} {pre} {nm2}: {int(dis.sum())} of {z.numel()} decisions differ; |z| there: "
                  f"{z.abs()[dis].tolist()[:8]}  hip there: {hip[dis].tolist()[:8]}", flush=True)
