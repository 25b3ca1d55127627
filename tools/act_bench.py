"""Latency of DrQV2Agent.act (encoder + actor forward on one frame stack, drqv2.py:164-175) -- dev tool."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drqv2
ag = drqv2.DrQV2Agent((9, 84, 84), (6,), "cuda", 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,500000)", 0.3, True)
obs = np.random.RandomState(0).randint(0, 256, (9, 84, 84)).astype(np.uint8)
for mode in (True, False):
    for _ in range(20):
        ag.act(obs, 5000, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        a = ag.act(obs, 5000, mode)      # returns a numpy action: includes the device->host copy and sync
    dt = (time.perf_counter() - t0) / n
    print(f"act(eval_mode={mode}): {1e6*dt:7.1f} us per call (host wall, action returned as numpy)", flush=True)
