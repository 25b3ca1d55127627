"""Host-side cost of DrQV2Agent.update (dev tool): wall per update with and without the metrics sync,
host time spent before the C entry point, and a cProfile of the Python prologue."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drqv2
from drqv2_amd import synth
dev = torch.device("cuda", 0)
torch.manual_seed(1)
B, A = 256, 6


def run(use_tb, n=200):
    agent = drqv2.DrQV2Agent((9, 84, 84), (A,), dev, 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,100000)", 0.3, use_tb)
    batch = tuple(t.to(dev) for t in synth.make_batch(B, A, 9, seed=0, smooth=True))
    it = iter(lambda: batch, None)
    for s in range(10):
        agent.update(it, 2 * s)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for s in range(n):
        h0 = time.perf_counter()
        agent.update(it, 2 * s)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"use_tb={use_tb}: {1e6*dt/n:8.1f} us/update wall, {1e6*host/n:8.1f} us inside update() on the host", flush=True)
    return agent, it


agent, it = run(True)
run(False)
pr = cProfile.Profile()
pr.enable()
for s in range(200):
    agent.update(it, 2 * s)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
