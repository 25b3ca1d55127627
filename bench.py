#!/usr/bin/env python3
"""Headline benchmark: agent updates/sec of DrQV2Agent.update() (HIP path) on synthetic, device-resident
replay batches.  Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.

  step      = one DrQV2Agent.update(): aug x2, encoder fwd x2, critic loss+backward, Adam(critic, encoder),
              actor loss+backward, Adam(actor), Polyak; metrics fetched (use_tb=True) every update.
  N = 1     workload = BASELINE.json configs[1]: cheetah_run, batch_size=256, 9x84x84 uint8 observations, A=6,
              feature_dim=50, hidden_dim=1024, fp32 (`value`).  The line also carries `scaling_base`: the N=1
              figure of the multi-GPU workload (configs[3], humanoid_run B=256), so the 1/2/4/8 curve has its
              own first point.
  N > 1     one process per GPU (torch.distributed, backend nccl = RCCL).  Started either by the driver
              (torch.distributed.run: RANK/LOCAL_RANK/WORLD_SIZE in the environment) or by this script itself:
              without WORLD_SIZE, `--gpus N` spawns N child ranks BEFORE anything touches the GPU and relays
              rank 0's line.  Workload = BASELINE.json configs[3]: humanoid_run (A=21, feature_dim=100), ONE global
              batch of 256 split N ways (`value`, "scaling": "strong"), two dependent gradient exchanges per update
              hidden under compute.  The same line carries `weak` (256 samples PER GPU, value
              = global samples/s / 256) and `n1_same_workload` (rank 0 alone on the plain single-GPU path, same
              job, same box).  --workload / --task / --batch / --weak override.
  roofline  = fp32: dominant kernel conv3x3_wino_kernel<41> (conv2 forward on both views + conv3 dgrad, Winograd
              F(2x2,3x3) on the f32 MFMA), timed live with events on the launch stream.  `achieved` counts the matrix FLOPs
              the kernel EXECUTES (256 v_mfma_f32_16x16x4_f32 per 16 output tiles): `frac` == `frac_executed` is a true
              fraction of the MFMA peak; `frac_direct_form` prices the same launches at SURVEY 8(d)'s direct-form FLOPs
              (2*32*288 per output pixel), which Winograd's 2.25x saving can push past 1; `frac_whole_step` is
              F_alg * updates/s / peak.  `other_kernels` carries the two weakest parts of the update, timed the same way.
              bf16 (configs[4]): the update is HBM-bound on its fp32 activations: `bound` "hbm", algorithmic bytes of
              SURVEY 8(d) per update over the measured update time against 8 TB/s.
              `traffic` is replayed from a committed --pmc measurement of the same command (`traffic_replayed`: true).
  cpu_baseline = the CPU oracle (oracle/drq_oracle.py, kind "port") on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from drqv2_amd import tasks as _tasks   # the reference's cfgs/ surface as data (one table for benches and tests)

# name: (A, feature_dim, lr, stddev_schedule) as cfgs/config.yaml + cfgs/task/<name>.yaml of the reference resolve
TASKS = {n: (_tasks.ACTION_DIM[n], _tasks.resolve(n)["feature_dim"], _tasks.resolve(n)["lr"],
             _tasks.resolve(n)["stddev_schedule"])
         for n in ("cheetah_run", "quadruped_walk", "humanoid_run", "cartpole_swingup")}
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 matrix == vector peak
H = 1024


def alg_flops_per_update(B, A, F, H=1024):
    """F_alg = B*[2*E_f + E_b + 8T + 4P + 12Q]."""
    E_f, E_b = 84_561_984, 160_409_664
    T = 2 * 39200 * F
    P = 2 * (F * H + H * H + H * A)
    Q = 2 * ((F + A) * H + H * H + H)
    return B * (2 * E_f + E_b + 8 * T + 4 * P + 12 * Q)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", choices=["auto", "config2", "config4"],
                    help="auto: config2 (cheetah_run B=256) on one GPU, config4 (humanoid_run, global B=256 split "
                         "over the GPUs) on several")
    ap.add_argument("--task", default=None, choices=list(TASKS), help="overrides the workload's task")
    ap.add_argument("--batch", type=int, default=256, help="global batch (strong) or per-GPU batch (--weak)")
    ap.add_argument("--weak", action="store_true", help="N>1: `value` is the weak-scaling run (--batch per GPU)")
    ap.add_argument("--strong", action="store_true", help="N>1: split ONE --batch over the GPUs (the default)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "allreduce", "direct", "zero1"],
                    help="gradient exchange of the data-parallel path (drqv2_amd.engine.GradExchange)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="rehearsal only: gloo carries the CUDA tensors through the host, so that several ranks can "
                         "share the single GPU of a development box (with --devices 0,0)")
    ap.add_argument("--devices", default=None,
                    help="rehearsal only: comma-separated device index per local rank (default: LOCAL_RANK)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: BASELINE configs[4] -- conv2..4 and the fc GEMMs of the update on the bf16 MFMA (fp32 "
                         "accumulation and storage).  New functionality, never the headline line: use with "
                         "--task humanoid_run --batch 2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip scaling_base / weak / n1_same_workload")
    ap.add_argument("--host-batch", action="store_true",
                    help="the replay iterator yields pinned HOST tensors (the reference boundary): every update "
                         "pays the H2D copy of its batch.  Reported for DESIGN.md section 6; never the headline value")
    ap.add_argument("--device-replay", nargs="?", const="indexed", default=None, choices=["indexed", "copy"],
                    help="batches come from the device-resident replay (index draws on the host) instead of one fixed "
                         "resident batch: the SURVEY 8f rank-2 path, reported in DESIGN.md.  indexed (default): the frames "
                         "stay in the store and the fused aug+conv1 launch gathers them; copy: drq_nstep_gather "
                         "materialises the batch first (the round-2 form)")
    ap.add_argument("--dp-schedule", action="store_true",
                    help="development: run the data-parallel schedule on a one-rank RCCL group (N=1 only)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as child processes.  The parent
    has not touched (and never touches) the GPU; it does not re-exec itself.  Rank 0's stdout (the JSON line) is
    relayed; the exit code is the first non-zero child code (the other ranks are then stopped), 124 after
    DRQ_BENCH_RANK_TIMEOUT_S seconds."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # supervise: the first rank that fails takes the others with it (a survivor would sit in init_process_group or a
    # collective until the RCCL / gloo timeout); SIGTERM / SIGINT to the parent reach the children; the whole run is bounded
    import signal
    import threading

    def kill_all(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()

    old_handlers = {sig: signal.signal(sig, lambda *_: (kill_all(), sys.exit(128 + sig))) for sig in (signal.SIGTERM, signal.SIGINT)}
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("DRQ_BENCH_RANK_TIMEOUT_S", "1500"))
    t_end = time.monotonic() + limit
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                print(f"bench.py: a rank exited with code {rc}; stopping the other ranks", file=sys.stderr)
                kill_all()
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_end:
                print(f"bench.py: ranks still running after {limit:.0f} s; stopping them", file=sys.stderr)
                kill_all()
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    reader.join(timeout=5.0)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return rc


class Runner:
    """One agent + one resident synthetic batch; times K updates the way the contract says."""

    def __init__(self, task, B_local, B_global, dev, rank, world, dp, exchange="auto", host_batch=False,
                 device_replay=False, dtype="f32"):
        import torch
        import drqv2
        from drqv2_amd import synth
        self.torch = torch
        self.task, self.B_local, self.B_global, self.world, self.dp = task, B_local, B_global, world, dp
        A, F, lr, sched = TASKS[task]
        self.A, self.F = A, F
        torch.manual_seed(1)
        self.agent = drqv2.DrQV2Agent((9, 84, 84), (A,), dev, lr, F, H, 0.01, 2000, 2, sched, 0.3, True)
        self.dtype = dtype
        if dtype == "bf16":
            self.agent.set_compute_dtype("bf16")
        if os.environ.get("DRQ_BENCH_METRICS_SPIN"):     # A/B of the metrics wait only (tools/): polls before sleeping
            self.agent._engine.metrics_spin = int(os.environ["DRQ_BENCH_METRICS_SPIN"])
        if dp:
            self.agent.enable_data_parallel(batch_is_global=False, exchange=exchange)
        batch = synth.make_batch(B_local, A, 9, seed=rank, smooth=True)
        batch = tuple(t.pin_memory() for t in batch) if host_batch else tuple(t.to(dev) for t in batch)
        self.batch = batch

        def replay():
            while True:
                yield batch

        self.it = replay()
        if device_replay:
            import numpy as np
            from drqv2_amd.replay import DeviceReplay
            store = DeviceReplay(4096, (9, 84, 84), A, 3, 0.99, dev, seed=rank, indexed=device_replay != "copy")
            obs_pool = batch[0].cpu().numpy()
            r = np.random.RandomState(rank)
            for e in range(16):                          # 16 episodes of 200 steps drawn from the synthetic frames
                T1 = 201
                store.add_episode({"observation": obs_pool[r.randint(0, obs_pool.shape[0], T1)],
                                   "action": r.uniform(-1, 1, (T1, A)).astype(np.float32),
                                   "reward": r.randn(T1, 1).astype(np.float32),
                                   "discount": np.ones((T1, 1), np.float32)})
            store.batch_size = B_local
            self.it = iter(store)
        self.step = 0

    def sync(self):
        if self.world > 1 and self.dp:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup):
        """W untimed updates, then exactly K timed ones bracketed by barrier + synchronize; MAX over ranks."""
        torch = self.torch
        ag = self.agent
        for _ in range(warmup):
            ag.update(self.it, self.step)
            self.step += 2
        ag.flush()
        self.sync()
        # the interpreter's cyclic collector stays out of the timed region (a generation-2 pass is a pause of milliseconds,
        # several updates long; update() itself leaves no cycles behind): collected before, switched back on after
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        stamps = []
        t0 = time.perf_counter()
        for _ in range(steps):
            metrics = ag.update(self.it, self.step)
            self.step += 2
            stamps.append(time.perf_counter())       # update() returns when its metrics have left the GPU
        ag.flush()         # data parallel: the last update's deferred Adam steps belong to the timed region
        self.sync()
        dt = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        per = sorted(1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:]))
        pct = (lambda q: per[min(len(per) - 1, int(q * len(per)))]) if per else (lambda q: None)
        if self.world > 1 and self.dp:
            import torch.distributed as dist
            t = torch.tensor([dt], device=ag._engine.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return {"value": steps * (self.B_global / 256.0) / dt, "ms_per_step": 1e3 * dt / steps,
                "ms_per_step_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)], "ms_per_step_max": per[-1] if per else None,
                "dt": dt, "last_metrics": metrics}

    def describe(self):
        prec = "fp32" if self.dtype == "f32" else "bf16 MFMA (conv2-4 + fc GEMMs; fp32 accumulation and storage)"
        return (f"{self.task} batch_size={self.B_local}/GPU ({self.B_global} global) 9x84x84 u8 obs, A={self.A}, "
                f"feature_dim={self.F}, hidden_dim={H}, {prec}, use_tb=True")

    def close(self):
        self.agent.flush()
        self.torch.cuda.synchronize()
        self.agent = None
        self.batch = None
        self.it = None
        self.torch.cuda.empty_cache()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # nothing above this line has initialised the GPU

    # stdout carries exactly one JSON line: anything else that writes to fd 1 (RCCL prints a version banner
    # there at init, libdrm complains about amdgpu.ids) is sent to stderr, the line goes out on a private dup.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = int(args.devices.split(",")[local_rank]) if args.devices else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dp = world > 1 or args.dp_schedule
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            args.exchange = "allreduce"        # gloo has no all-to-all for device tensors

    workload = args.workload if args.workload != "auto" else ("config2" if world == 1 else "config4")
    task = args.task or ("cheetah_run" if workload == "config2" else "humanoid_run")
    strong = world > 1 and not args.weak
    if strong and args.batch % world:
        raise SystemExit(f"--batch {args.batch} does not split over {world} GPUs")
    B_local = args.batch // world if strong else args.batch
    B_global = B_local * world
    A, F, _, _ = TASKS[task]

    main_run = Runner(task, B_local, B_global, dev, rank, world, use_dp, args.exchange, args.host_batch,
                      args.device_replay, args.dtype)
    res = main_run.run(args.steps, args.warmup)
    dt = res.pop("dt")
    out = {
        "metric": "agent updates/sec (batch=256, 9x84x84 obs)", "value": res["value"], "unit": "updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "ms_per_step_p10_p50_p90": res["ms_per_step_p10_p50_p90"], "ms_per_step_max": res["ms_per_step_max"],
        "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" + (" (batch copied host->device every update)" if args.host_batch else "")
                                + (" (batches assembled by the device replay)" if args.device_replay else ""),
        "config": {"workload": main_run.describe(), "parallelism": f"dp{world}", "global_batch": B_global,
                   "backend": ("rccl" if args.backend == "nccl" else "gloo (rehearsal)") if use_dp else None,
                   "baseline_config": {"config2": 1, "config4": 3}[workload] if args.task is None else None},
        "alg_gflop_per_update": alg_flops_per_update(B_global, A, F, H) / 1e9,
        "last_metrics": res["last_metrics"],
    }
    frac_whole = alg_flops_per_update(B_global, A, F, H) * args.steps / dt / (PEAK_FP32_TFLOPS * 1e12 * world)
    if args.dtype == "f32":
        out["frac_fp32_peak_whole_step"] = frac_whole
    if use_dp and main_run.agent._engine.exchange is not None:
        ex = main_run.agent._engine.exchange
        out["exchange"] = {"mode": ex.mode, "choice_by_bucket_floats": {str(k): v for k, v in ex.choice.items()},
                           "measured_us_by_bucket_floats": {str(k): v for k, v in ex.timings_us.items()}}

    if not args.no_roofline and args.dtype == "f32":
        # every rank runs it (the updates inside carry the data-parallel collectives); rank 0 reports
        roof = roofline_conv(main_run.agent, B_local, main_run.it, main_run.step, A, F)
        roof["frac_whole_step"] = frac_whole
        main_run.agent.flush()
        if rank == 0:
            out["roofline"] = roof
    elif not args.no_roofline:
        if rank == 0:
            out["roofline"] = roofline_hbm(main_run.agent, task, B_local, A, res["ms_per_step"], args.dtype)
    if use_dp and world > 1 and not args.no_extras:
        exposed = exposed_exchange_us(main_run)
        if rank == 0 and exposed is not None:
            out.setdefault("exchange", {})["exposed_us_per_update"] = exposed
    main_run.close()

    extras = not (args.no_extras or args.host_batch or args.device_replay or args.dp_schedule or args.dtype != "f32")
    k2, w2 = max(20, args.steps // 2), max(5, args.warmup // 2)
    if extras and world == 1 and workload == "config2" and args.task is None:
        # first point of the configs[3] scaling curve, on the same box in the same run
        r = Runner("humanoid_run", args.batch, args.batch, dev, 0, 1, False)
        rr = r.run(k2, w2)
        out["scaling_base"] = {"workload": r.describe(), "value": rr["value"], "ms_per_step": rr["ms_per_step"],
                               "steps": k2, "warmup": w2, "n_gpus": 1,
                               "note": "N=1 of BASELINE configs[3] (humanoid_run, global batch 256): the workload "
                                       "`value` is quoted on when --gpus > 1"}
        r.close()
    if extras and world > 1:
        import torch.distributed as dist
        if strong:
            # weak scaling beside it: --batch samples on EVERY GPU
            r = Runner(task, args.batch, args.batch * world, dev, rank, world, True, args.exchange)
            rr = r.run(k2, w2)
            out["weak"] = {"scaling": "weak", "workload": r.describe(), "value": rr["value"],
                           "ms_per_step": rr["ms_per_step"], "steps": k2, "warmup": w2, "global_batch": args.batch * world,
                           "note": "value = global samples/s / 256 (batch-256 equivalents)"}
            r.close()
        # the same global batch on ONE GPU through the plain (non-data-parallel) path: rank 0 alone, others wait
        if rank == 0:
            r = Runner(task, B_global if strong else B_local, B_global if strong else B_local, dev, 0, 1, False)
            rr = r.run(k2, w2)
            out["n1_same_workload"] = {"workload": r.describe(), "value": rr["value"], "ms_per_step": rr["ms_per_step"],
                                       "steps": k2, "warmup": w2}
            r.close()
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(task, B_local)
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if use_dp:
        import torch.distributed as dist
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    return 0


def kernel_traffic(name="kernel_traffic.json"):
    """Memory-side KB per launch of the kernels of the bench, as measured by tools/pmc_bench.sh (separate rocprofv3
    --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `python bench.py ...`) and written, with the commit it was measured at,
    to profiles/<name>.  None when that file is missing."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def roofline_conv(agent, B, it, step, A, F):
    """conv3x3_wino_kernel<41,...>: its two launches per update (conv2 forward on 2B frames, conv3 dgrad on B), timed
    IN the update with HIP events the library records around those two launches on the stream they run on
    (DrqStep.timing_events).  In isolation, back to back, the same launches run ~10 % slower (the chip holds a
    lower clock under an MFMA-only load than inside the update's mix of kernels), and rocprofv3's per-kernel
    average of the bench agrees with the in-update figure, not with the isolated one.  The same events bracket the
    Winograd weight-gradient launch and the two head sections (`other_kernels`)."""
    import torch
    eng = agent._engine
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
    for e in ev:
        e.record()                     # torch creates the hipEvent_t at the first record
    torch.cuda.synchronize()
    eng.set_timing_events(ev)
    acc = [0.0] * 5
    n = 0
    try:
        for u in range(24):
            agent.update(it, step)
            step += 2
            torch.cuda.synchronize()   # the events of THIS update have been reached before they are re-recorded
            if u >= 4:
                for k in range(5):
                    acc[k] += ev[2 * k].elapsed_time(ev[2 * k + 1]) * 1e-3
                n += 1
    finally:
        eng.set_timing_events(None)
    t_f, t_d, t_w, t_hc, t_ha = (a / n for a in acc)
    traffic, note, tcommit = None, None, None
    tr = kernel_traffic()
    names = ("conv3x3_wino_kernel<41, false, true, 0>", "conv3x3_wino_kernel<41, true, false, 0>")   # fwd 2B, dgrad B
    if tr is not None and B == tr.get("B") and all(n in tr.get("kernels", {}) for n in names):
        per = [(2 * tr["kernels"][n]["FETCH_SIZE_KB"] + tr["kernels"][n]["WRITE_SIZE_KB"]) * 1024 for n in names]
        traffic = sum(per) / len(per)
        tcommit = tr.get("commit")
        note = (f"bytes per launch, mean of the two launches, (2*FETCH_SIZE+WRITE_SIZE)*1024 from separate --pmc passes of "
                f"the bench at B={tr['B']} (profiles/kernel_traffic.json, measured at commit {tcommit}); NOT measured in "
                "this run; algorithmic bytes are 210 MB (fwd: 110 in + 100 out) / 155 MB (dgrad: 55 in + 50 mask + 50 out)")
    # executed matrix work: a unit = 16 output tiles (2x2 pixels each) x 32 channels x 16 positions x 8 k-steps
    # = 256 v_mfma_f32_16x16x4_f32 of 2*16*16*4 FLOP; 20x20 tiles cover the 39x39 outputs of a frame
    units_f, units_d = (2 * B * 400 + 15) // 16, (B * 400 + 15) // 16
    ex_f, ex_d = units_f * 256 * 2048, units_d * 256 * 2048
    fl_f = 2 * 32 * 288 * (2 * B) * 39 * 39          # the direct form's FLOPs for the same outputs
    fl_d = 2 * 32 * 288 * B * 39 * 39
    ach = (ex_f + ex_d) / (t_f + t_d) / 1e12
    direct = (fl_f + fl_d) / (t_f + t_d) / 1e12
    # the Winograd weight gradient of conv2..4 (one launch): per tile 16 positions x 32x32 MACs; tiles 20^2, 19^2, 18^2
    ex_w = B * (400 + 361 + 324) * 16 * 32 * 32 * 2
    fl_w = 2 * 32 * 288 * B * (39 * 39 + 37 * 37 + 35 * 35)
    T = 2 * 39200 * F
    P = 2 * (F * H + H * H + H * A)
    Q = 2 * ((F + A) * H + H * H + H)
    fl_hc, fl_ha = B * (6 * T + 2 * P + 8 * Q), B * (2 * T + 2 * P + 4 * Q)
    other = {
        "conv3x3_wgrad_wino3_kernel": {
            "what": "weight gradients of conv2..4 in one launch (Winograd, 256 accumulators per wave)", "us": 1e6 * t_w,
            "executed_tflops": ex_w / t_w / 1e12, "frac_executed": ex_w / t_w / 1e12 / PEAK_FP32_TFLOPS,
            "frac_direct_form": fl_w / t_w / 1e12 / PEAK_FP32_TFLOPS},
        "heads_critic_update": {
            "what": "phase 4: four trunks, policy, twin-Q forward, TD loss, backward to the encoder output "
                    "(6T + 2P + 8Q of SURVEY 8d)", "us": 1e6 * t_hc, "alg_gflop": fl_hc / 1e9,
            "tflops": fl_hc / t_hc / 1e12, "frac": fl_hc / t_hc / 1e12 / PEAK_FP32_TFLOPS},
        "heads_actor_update": {
            "what": "phases 6-7 without the critic's optimiser step: critic trunk + twin-Q forward, actor loss, backward "
                    "to the policy and the actor trunk (2T + 2P + 4Q)", "us": 1e6 * t_ha, "alg_gflop": fl_ha / 1e9,
            "tflops": fl_ha / t_ha / 1e12, "frac": fl_ha / t_ha / 1e12 / PEAK_FP32_TFLOPS},
    }
    return {"kernel": "conv3x3_wino_kernel<41> (Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32)", "bound": "mfma",
            "achieved": ach, "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "frac_executed": ach / PEAK_FP32_TFLOPS,
            "frac_direct_form": direct / PEAK_FP32_TFLOPS, "direct_form_tflops": direct,
            "traffic": traffic, "traffic_replayed": traffic is not None, "traffic_measured_at_commit": tcommit,
            "traffic_note": note,
            "achieved_counts": "matrix FLOPs the kernel executes (Winograd needs 16/36 of the direct form's, on 20x20 "
                               "tiles of 2x2 for 39x39 outputs); the VALU transforms (56 adds per tile and channel) "
                               "share the f32 datapath with the MFMA and are not counted",
            "avg_launch_us": 0.5e6 * (t_f + t_d), "launch_us": {"conv2_fwd_2B": 1e6 * t_f, "conv3_dgrad_B": 1e6 * t_d},
            "executed_gflop_per_launch": {"conv2_fwd_2B": ex_f / 1e9, "conv3_dgrad_B": ex_d / 1e9},
            "alg_gflop_per_launch": {"conv2_fwd_2B": fl_f / 1e9, "conv3_dgrad_B": fl_d / 1e9}, "frames_per_launch": B,
            "other_kernels": other,
            "timing": "hipEvent pairs recorded by the library around the launches inside 20 update() calls"}


PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable with a float4 copy)


def alg_hbm_bytes_per_update(agent, B, A):
    """inputs + saved activations of the obs branch written once and read once (fp32) + Adam 28 B/param +
    Polyak 12 B/critic param."""
    n_all = sum(p.numel() for m in (agent.encoder, agent.actor, agent.critic) for p in m.parameters())
    n_critic = sum(p.numel() for p in agent.critic.parameters())
    inputs = B * (2 * 9 * 84 * 84 + 4 * A + 8)
    acts = 2 * B * 4 * 32 * (41 * 41 + 39 * 39 + 37 * 37 + 35 * 35)
    return inputs + acts + 28 * n_all + 12 * n_critic


def roofline_hbm(agent, task, B, A, ms_per_step, dtype):
    """bf16 (BASELINE configs[4]): with the matrix work on the bf16 MFMA the update is bound by the HBM traffic of its
    fp32 activations (balance ~312 FLOP/B against ~240 FLOP/B of the path).  Whole-update figure: algorithmic
    bytes per update over the measured update time; `traffic` = memory-side bytes per update summed over every kernel of
    a committed --pmc measurement of this command, when there is one for this workload."""
    alg = alg_hbm_bytes_per_update(agent, B, A)
    ach = alg / (ms_per_step * 1e-3) / 1e9
    traffic, note, tcommit = None, None, None
    tr = kernel_traffic(f"kernel_traffic_{dtype}_{task}_b{B}.json")
    if tr is not None and tr.get("B") == B:
        traffic = sum((2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024 * k.get("launches_per_update", 1.0)
                      for k in tr["kernels"].values() if k["FETCH_SIZE_KB"] == k["FETCH_SIZE_KB"])
        tcommit = tr.get("commit")
        note = (f"bytes per update: sum over all kernels of launches x (2*FETCH_SIZE+WRITE_SIZE)*1024, separate --pmc passes "
                f"of this command (profiles/kernel_traffic_{dtype}_{task}_b{B}.json, commit {tcommit}); NOT measured in this run")
    return {"kernel": "whole update (HBM-bound on the fp32 activations with the matrix work on the bf16 MFMA)",
            "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
            "alg_mbytes_per_update": alg / 1e6, "ms_per_update": ms_per_step,
            "traffic": traffic, "traffic_replayed": traffic is not None, "traffic_measured_at_commit": tcommit,
            "traffic_note": note,
            "achieved_counts": "SURVEY 8(d) algorithmic HBM bytes per update (inputs, obs-branch activations written once and "
                               "read once in fp32, Adam 28 B/param, Polyak 12 B/param) / measured time per update"}


def exposed_exchange_us(runner, updates=12):
    """Data parallel: how long the compute stream WAITS for each gradient exchange, per update (events around the stream
    waits; a separate short pass, the events themselves cost the stream a bubble each).  None without a process group."""
    eng = runner.agent._engine
    if eng.pg is None:
        return None
    eng.profile_exchange = True
    eng.exchange_wait_us = {}
    try:
        for _ in range(updates):
            runner.agent.update(runner.it, runner.step)
            runner.step += 2
        runner.agent.flush()
        runner.torch.cuda.synchronize()
        res = eng.collect_exchange_waits()
    finally:
        eng.profile_exchange = False
    return res


def cpu_baseline(task, B):
    """The CPU oracle timed on the host cores: same workload, bounded sample (2 warm-up + 4 timed updates)."""
    import torch
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
    # bounded sample (~10-30 s of CPU work): fewer timed updates at the large-batch configurations
    warm, timed = (2, 4) if B <= 512 else (1, 2)
    ts = []
    for u in range(warm + timed):
        t0 = time.perf_counter()
        ag.update(batch, 2 * u, *draws)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[warm:])
    med = 0.5 * (ts[(timed - 1) // 2] + ts[timed // 2])
    return {"value": 1.0 / med, "unit": "updates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/drq_oracle.py OracleAgent.update, {task} B={B} fp32, {warm} warm-up + {timed} timed updates, "
                      "median"}


if __name__ == "__main__":
    sys.exit(main())
