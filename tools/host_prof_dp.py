"""Host-side cost of DrQV2Agent.update on the data-parallel schedule (RCCL group of one rank): cProfile (dev tool)."""
import cProfile, os, pstats, socket, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drqv2
from drqv2_amd import synth
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
torch.manual_seed(1)
B, A = 256, 6
agent = drqv2.DrQV2Agent((9, 84, 84), (A,), dev, 1e-4, 50, 1024, 0.01, 2000, 2, "linear(1.0,0.1,100000)", 0.3, True)
agent.enable_data_parallel(batch_is_global=False)
batch = tuple(t.to(dev) for t in synth.make_batch(B, A, 9, seed=0, smooth=True))
it = iter(lambda: batch, None)
for s_ in range(20):
    agent.update(it, 2 * s_)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s_ in range(200):
    agent.update(it, 2 * s_)
torch.cuda.synchronize()
print(f"{1e6 * (time.perf_counter() - t0) / 200:.1f} us per update (wall)")
pr = cProfile.Profile()
pr.enable()
for s_ in range(200):
    agent.update(it, 2 * s_)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
