"""Host-side cost of feeding the update from the device replay (dev tool): time inside DeviceReplay.sample() and inside
update() per iteration, and a cProfile of the sampling."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drqv2
from drqv2_amd import synth
from drqv2_amd.replay import DeviceReplay
dev = torch.device("cuda", 0)
torch.manual_seed(1)
B, A = 256, 6
agent = drqv2.DrQV2Agent((9, 84, 84), (A,), dev, 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,100000)", 0.3, True)
batch = synth.make_batch(B, A, 9, seed=0, smooth=True)
store = DeviceReplay(4096, (9, 84, 84), A, 3, 0.99, dev, seed=0, indexed=True)
obs_pool = batch[0].numpy()
r = np.random.RandomState(0)
for e in range(16):
    T1 = 201
    store.add_episode({"observation": obs_pool[r.randint(0, obs_pool.shape[0], T1)],
                       "action": r.uniform(-1, 1, (T1, A)).astype(np.float32),
                       "reward": r.uniform(0, 1, (T1, 1)).astype(np.float32),
                       "discount": np.ones((T1, 1), np.float32)})
store.batch_size = B


class Timed:
    def __init__(self):
        self.t = 0.0
    def __iter__(self):
        return self
    def __next__(self):
        t0 = time.perf_counter()
        b = store.sample(B)
        self.t += time.perf_counter() - t0
        return b


it = Timed()
for s in range(20):
    agent.update(it, 2 * s)
torch.cuda.synchronize()
it.t = 0.0
n = 300
t0 = time.perf_counter(); host = 0.0
for s in range(n):
    h0 = time.perf_counter()
    agent.update(it, 2 * s)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"indexed device replay: {1e6*dt/n:8.1f} us/update wall, {1e6*host/n:8.1f} us inside update(), of which {1e6*it.t/n:6.1f} us in sample()")
pr = cProfile.Profile()
pr.enable()
for s in range(300):
    store.sample(B)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
