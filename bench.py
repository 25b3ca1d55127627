#!/usr/bin/env python3
"""Headline benchmark: agent updates/sec of DrQV2Agent.update() (HIP path) on synthetic, device-resident
replay batches.  Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.

  step      = one DrQV2Agent.update(): aug x2, encoder fwd x2, critic loss+backward, Adam(critic, encoder),
              actor loss+backward, Adam(actor), Polyak; metrics fetched (use_tb=True) every update.
  workload  = BASELINE.json configs[1]: cheetah_run, batch_size=256, 9x84x84 uint8 observations, A=6,
              feature_dim=50, hidden_dim=1024, fp32.  N>1: one process per GPU (torch.distributed, RCCL),
              every rank trains on its own 256-sample shard (weak scaling; --strong splits ONE 256 batch)
              with two gradient all-reduces per update; value = batch-256 updates/sec of the whole job
              = global samples/sec / 256.
  roofline  = dominant kernel conv3x3_kernel<32,41,1> (conv2 forward on both views + conv3 dgrad),
              timed live with events on the launch stream; algorithmic FLOPs = 2*32*288 per output pixel.
  cpu_baseline = the CPU oracle (oracle/drq_oracle.py, kind "port") on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TASKS = {   # name: (A, feature_dim, lr, stddev_schedule)  -- cfgs/task/*.yaml of the reference
    "cheetah_run": (6, 50, 1e-4, "linear(1.0,0.1,500000)"),
    "quadruped_walk": (12, 50, 1e-4, "linear(1.0,0.1,500000)"),
    "humanoid_run": (21, 100, 8e-5, "linear(1.0,0.1,2000000)"),
    "cartpole_swingup": (1, 50, 1e-4, "linear(1.0,0.1,100000)"),
}
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 matrix == vector peak


def alg_flops_per_update(B, A, F, H=1024):
    """F_alg = B*[2*E_f + E_b + 8T + 4P + 12Q]."""
    E_f, E_b = 84_561_984, 160_409_664
    T = 2 * 39200 * F
    P = 2 * (F * H + H * H + H * A)
    Q = 2 * ((F + A) * H + H * H + H)
    return B * (2 * E_f + E_b + 8 * T + 4 * P + 12 * Q)


def main():
    # stdout carries exactly one JSON line: anything else that writes to fd 1 (RCCL prints a version banner
    # there at init, libdrm complains about amdgpu.ids) is sent to stderr, the line goes out on a private dup.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak) or global batch (--strong)")
    ap.add_argument("--task", default="cheetah_run", choices=list(TASKS))
    ap.add_argument("--strong", action="store_true", help="split ONE --batch over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--host-batch", action="store_true",
                    help="the replay iterator yields pinned HOST tensors (the reference boundary): every update "
                         "pays the H2D copy of its batch.  Reported for DESIGN.md section 6; never the headline value")
    ap.add_argument("--device-replay", action="store_true",
                    help="batches come from the device-resident replay (drq_nstep_gather, index draws on the host) "
                         "instead of one fixed resident batch: the SURVEY 8f rank-2 path, reported in DESIGN.md")
    ap.add_argument("--dp-schedule", action="store_true",
                    help="development: run the data-parallel schedule on a one-rank RCCL group (N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dp = world > 1 or args.dp_schedule
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import drqv2
    from drqv2_amd import synth

    A, F, lr, sched = TASKS[args.task]
    H = 1024
    B_local = args.batch // world if args.strong else args.batch
    B_global = B_local * world
    torch.manual_seed(1)
    agent = drqv2.DrQV2Agent((9, 84, 84), (A,), dev, lr, F, H, 0.01, 2000, 2, sched, 0.3, True)
    if use_dp:
        agent.enable_data_parallel(batch_is_global=False)
    batch = synth.make_batch(B_local, A, 9, seed=rank, smooth=True)
    batch = tuple(t.pin_memory() for t in batch) if args.host_batch else tuple(t.to(dev) for t in batch)

    def replay():
        while True:
            yield batch

    it = replay()
    if args.device_replay:
        import numpy as np
        from drqv2_amd.replay import DeviceReplay
        store = DeviceReplay(4096, (9, 84, 84), A, 3, 0.99, dev, seed=rank)
        obs_pool = batch[0].cpu().numpy()
        r = np.random.RandomState(rank)
        for e in range(16):                          # 16 episodes of 200 steps drawn from the synthetic frames
            T1 = 201
            store.add_episode({"observation": obs_pool[r.randint(0, obs_pool.shape[0], T1)],
                               "action": r.uniform(-1, 1, (T1, A)).astype(np.float32),
                               "reward": r.randn(T1, 1).astype(np.float32),
                               "discount": np.ones((T1, 1), np.float32)})
        store.batch_size = B_local
        it = iter(store)
    step = 0
    for _ in range(args.warmup):
        agent.update(it, step)
        step += 2
    agent.flush()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    stamps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = agent.update(it, step)
        step += 2
        stamps.append(time.perf_counter())       # update() returns when its metrics have left the GPU
    agent.flush()          # data parallel: the last update's deferred Adam(actor) belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    per = sorted(1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:]))
    pct = (lambda q: per[min(len(per) - 1, int(q * len(per)))]) if per else (lambda q: None)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    value = args.steps * (B_global / 256.0) / dt
    out = {
        "metric": "agent updates/sec (batch=256, 9x84x84 obs)", "value": value, "unit": "updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "ms_per_step_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)],
        "ms_per_step_max": per[-1] if per else None,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" + (" (batch copied host->device every update)" if args.host_batch else "")
                                + (" (batches assembled by the device replay)" if args.device_replay else ""),
        "config": {"workload": f"{args.task} batch_size={B_local}/GPU ({B_global} global) 9x84x84 u8 obs, A={A}, "
                               f"feature_dim={F}, hidden_dim={H}, fp32, use_tb=True",
                   "parallelism": f"dp{world}", "global_batch": B_global},
        "alg_gflop_per_update": alg_flops_per_update(B_global, A, F, H) / 1e9,
        "frac_fp32_peak_whole_step": alg_flops_per_update(B_global, A, F, H) * args.steps / dt /
                                     (PEAK_FP32_TFLOPS * 1e12 * world),
        "last_metrics": metrics,
    }

    if not args.no_roofline:
        # every rank runs it (the updates inside carry the data-parallel collectives); rank 0 reports
        roof = roofline_conv(agent, B_local, it, step)
        agent.flush()
        if rank == 0:
            out["roofline"] = roof
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.task, B_local)
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if use_dp:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


# HBM-side bytes per launch of the two launches of conv3x3_kernel<32,41,1> at B = 256, from separate rocprofv3
# --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (profiles/r01_conv_traffic_pmc.txt, tools/pmc_traffic.sh):
# (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the factor 2 being the guide's gfx950 correction of FETCH_SIZE.
CONV_TRAFFIC_KB_B256 = {"conv2_fwd_2B": (68048.2, 106558.5), "conv3_dgrad_B": (63133.0, 54238.4)}
TRAFFIC_NOTE = ("bytes per launch, mean of the two launches, (2*FETCH_SIZE+WRITE_SIZE)*1024 from separate --pmc passes "
                "at B=256 (profiles/r01_conv_traffic_pmc.txt); algorithmic bytes are 210 MB (fwd) / 155 MB (dgrad); the "
                "x2 on FETCH_SIZE is calibrated for 16-byte lane loads, this kernel uses 12-byte ones: the uncorrected "
                "sum is 179 / 120 MB")


def roofline_conv(agent, B, it, step):
    """conv3x3_kernel<32,41,1>: its two launches per update (conv2 forward on 2B frames, conv3 dgrad on B), timed
    IN the update with HIP events the library records around those two launches on the stream they run on
    (DrqStep.timing_events).  In isolation, back to back, the same launches run ~10 % slower (the chip holds a
    lower clock under an MFMA-only load than inside the update's mix of kernels), and rocprofv3's per-kernel
    average of the bench agrees with the in-update figure, not with the isolated one."""
    eng = agent._engine
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()                     # torch creates the hipEvent_t at the first record
    torch.cuda.synchronize()
    eng.set_timing_events(ev)
    t_f, t_d, n = 0.0, 0.0, 0
    try:
        for u in range(24):
            agent.update(it, step)
            step += 2
            torch.cuda.synchronize()   # the events of THIS update have been reached before they are re-recorded
            if u >= 4:
                t_f += ev[0].elapsed_time(ev[1]) * 1e-3
                t_d += ev[2].elapsed_time(ev[3]) * 1e-3
                n += 1
    finally:
        eng.set_timing_events(None)
    t_f /= n
    t_d /= n
    traffic = None
    if B == 256:
        traffic = sum((2 * f + w) * 1024 for f, w in CONV_TRAFFIC_KB_B256.values()) / len(CONV_TRAFFIC_KB_B256)
    fl_f = 2 * 32 * 288 * (2 * B) * 39 * 39
    fl_d = 2 * 32 * 288 * B * 39 * 39
    ach = (fl_f + fl_d) / (t_f + t_d) / 1e12
    return {"kernel": "conv3x3_kernel<32,41,1>", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "traffic": traffic,
            "traffic_note": TRAFFIC_NOTE if traffic is not None else None,
            "avg_launch_us": 0.5e6 * (t_f + t_d), "launch_us": {"conv2_fwd_2B": 1e6 * t_f, "conv3_dgrad_B": 1e6 * t_d},
            "alg_gflop_per_launch": {"conv2_fwd_2B": fl_f / 1e9, "conv3_dgrad_B": fl_d / 1e9},
            "timing": "hipEvent pairs recorded by the library around the two launches inside 20 update() calls"}


def cpu_baseline(task, B):
    """The CPU oracle timed on the host cores: same workload, bounded sample (2 warm-up + 4 timed updates)."""
    from drqv2_amd import synth
    from oracle import drq_oracle as O
    A, F, lr, sched = TASKS[task]
    # the GPU box gives one GPU's job a 16-core share of a much larger host: more threads than that thrash
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    enc, actor, critic = synth.make_weights(9, A, F, 1024, 0)
    ag = O.OracleAgent(enc, actor, critic, lr, stddev_schedule=sched)
    batch = synth.make_batch(B, A, 9, seed=0, smooth=True)
    draws = synth.make_draws(B, A, seed=0)
    ts = []
    for u in range(6):
        t0 = time.perf_counter()
        ag.update(batch, 2 * u, *draws)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    med = 0.5 * (ts[1] + ts[2])
    return {"value": 1.0 / med, "unit": "updates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/drq_oracle.py OracleAgent.update, {task} B={B} fp32, 2 warm-up + 4 timed updates, median"}


if __name__ == "__main__":
    main()
