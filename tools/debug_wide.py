"""Gradient errors of one update vs the fp64 oracle at several batch sizes (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_step as T
for B in [int(b) for b in sys.argv[1:]] or [32, 64]:
    cfg = dict(T.WIDE["cheetah_b64"]); cfg["B"] = B
    ag = T.make_agent(cfg)
    o64 = T.make_oracle(cfg, torch.float64)
    m, batch, (sh_o, sh_n, n_c, n_a) = T.run_hip(ag, cfg, 0)
    eng = ag._engine
    xin = eng.ws_view("AUG", B, (2 * B, 9, 84, 84)).cpu()
    o64.update(batch, 0, sh_o, sh_n, n_c, n_a, enc_in_override=(xin[:B], xin[B:]), keep=True)
    print("B =", B)
    for nm, mod, key in (("enc", ag.encoder, "g_enc"), ("critic", ag.critic, "g_critic"), ("actor", ag.actor, "g_actor")):
        for (pn, p), g64 in zip(mod.named_parameters(), o64.last[key].values()):
            e = T.nerr(p.grad, g64)
            if e > 5e-6 or nm == "enc":
                print(f"   {nm:6s} {pn:22s} {e:.3e}")
