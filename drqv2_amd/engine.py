"""Host side of the HIP update step: parameter/gradient/Adam arenas, the static workspace, the
DrqStep descriptor and the data-parallel exchange points.  Memory is allocated and owned by
PyTorch-ROCm; the library only ever sees raw pointers."""
import ctypes

import torch

from . import _lib
from ._lib import DrqStep, check, ptr

NETS = ("enc", "critic", "actor", "target")


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (defaults; drqv2.py:148-150) over one contiguous segment of the
    parameter arena.  `step()` is one fused launch; inside DrQV2Agent.update() the same kernel is
    issued by the step library, which advances `t` through `begin_step()`."""

    def __init__(self, params, lr, seg_p, seg_g, seg_m, seg_v):
        super().__init__(list(params), dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False))
        self._p, self._g, self._m, self._v = seg_p, seg_g, seg_m, seg_v
        self.t = 0

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def begin_step(self):
        self.t += 1
        return self.t

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten (never accumulated) by the backward kernels
        return None

    @torch.no_grad()
    def step(self, closure=None, gscale=1.0, tgt=None, tau=0.0):
        from . import ops
        ops.adam_flat(self._p, self._g, self._m, self._v, self.lr, self.begin_step(), gscale, tgt, tau)

    # snapshot helpers (train.py:192-204 pickles the agent)
    def export_state(self):
        return {"t": self.t, "lr": self.lr, "m": self._m.detach().cpu().clone(), "v": self._v.detach().cpu().clone()}

    def import_state(self, st):
        self.t = int(st["t"])
        self.param_groups[0]["lr"] = st["lr"]
        self._m.copy_(st["m"])
        self._v.copy_(st["v"])


def shard_bounds(n_rows, world, rank, batch_is_global):
    """Rows [lo,hi) of the GLOBAL batch that `rank` trains on, and the global batch size.
    batch_is_global: the iterator yields the same global batch on every rank (n_rows rows);
    otherwise it yields this rank's shard (n_rows rows each)."""
    if world > 1 and batch_is_global:
        if n_rows % world != 0:
            raise ValueError("global batch must divide evenly over the ranks")
        per, n_global = n_rows // world, n_rows
    else:
        per, n_global = n_rows, n_rows * world
    return rank * per, (rank + 1) * per, n_global


def grad_buckets(layout):
    """The two dependent exchanges of one update as [beg,end) ranges of the gradient arena:
    (encoder+critic) after the critic backward, (actor) after the actor backward."""
    seg = layout["seg"]
    return (seg["enc"][0], seg["critic"][1]), (seg["actor"][0], seg["actor"][1])


def grad_buckets_overlap(layout):
    """The same exchanges as the overlapped schedule issues them: critic (complete before the encoder
    backward starts, reduced while it runs), encoder (tiny, after it), actor (reduced while the NEXT update's
    encoder forward runs)."""
    seg = layout["seg"]
    return tuple((seg[k][0], seg[k][1]) for k in ("critic", "enc", "actor"))


class _StreamWork:
    """Handle of an exchange that runs on a side stream: wait() makes the CURRENT stream wait for it (like the
    Work objects of torch.distributed on RCCL), never the host."""

    def __init__(self, stream, keep=None):
        self.stream, self.keep = stream, keep

    def wait(self):
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
        self.keep = None


class GradExchange:
    """SUM of a contiguous range of the gradient arena over the ranks, in place, asynchronously.

    allreduce  one all-reduce per bucket (RCCL chooses ring / tree / its own direct algorithms).
    direct     two-phase exchange sized for a fully connected xGMI node: the bucket is cut
               into `world` equal slices; an all-to-all hands rank r the `world` copies of slice r (every rank talks
               to its world-1 peers at once: all links carry 1/world of the bucket instead of a ring pushing
               (world-1)/world of it through each hop); rank r adds them in rank order (the same order on every
               rank: bit-identical results everywhere); an all-gather returns the reduced slices.  Needs the
               bucket length to divide by world*64 floats (segments are padded to 512 floats: world 2, 4, 8).
    zero1      sharded optimiser step (ZeRO-1) on the same two-phase pattern: all-to-all of the GRADIENT slices, then
               rank r adds its `world` copies in rank order and steps Adam on the slice of the segment it owns in ONE
               kernel (drq_adam_reduce_flat: the summed gradient never returns to memory), then an all-gather of the
               stepped PARAMETER slices.  Same bytes on the links as `direct`, 1/world of the optimiser's HBM traffic
               per rank, and the optimiser step leaves the compute stream.  Adam moments are maintained only for the
               owned slice (StepEngine.gather_optimizer_state() reassembles them for snapshots).  Bit-identical to
               `direct` + replicated Adam.
    auto       times the candidates on scratch copies of the real buckets at enable time (max over ranks): exchange +
               replicated Adam for `allreduce` and `direct`, the whole sharded sequence for `zero1`, and keeps the
               fastest per bucket -- which one wins depends on the bucket size and the RCCL build, and this
               repository's CI box has one GPU, so the choice is made where the job runs, by measurement."""

    MODES = ("allreduce", "direct", "auto", "zero1")

    def __init__(self, pg, world, device, mode="allreduce"):
        if mode not in self.MODES:
            raise ValueError(f"exchange mode {mode!r}: one of {self.MODES}")
        self.pg, self.world, self.device = pg, world, torch.device(device)
        import torch.distributed as dist
        self.rank = dist.get_rank(pg) if world > 1 or dist.is_initialized() else 0
        self.mode = mode
        self.choice = {}          # bucket length -> "allreduce" | "direct"  (auto)
        self.timings_us = {}      # bucket length -> {"allreduce": us, "direct": us}
        self._side = None
        self._scratch = {}

    def direct_ok(self, n):
        return self.world > 1 and n % (self.world * 64) == 0

    def _bufs(self, n):
        b = self._scratch.get(n)
        if b is None:
            b = (torch.empty(n, device=self.device, dtype=torch.float32),
                 torch.empty(n // self.world, device=self.device, dtype=torch.float32))
            self._scratch[n] = b
        return b

    def _direct(self, t):
        """Runs on the current stream (the callers put it on a side stream)."""
        import torch.distributed as dist
        n = t.numel()
        recv, red = self._bufs(n)
        dist.all_to_all_single(recv, t, group=self.pg)                    # recv[j] = rank j's copy of MY slice
        self._sum_slices(recv, red, n // self.world)                      # fixed (rank) order
        dist.all_gather_into_tensor(t, red, group=self.pg)

    def _sum_slices(self, recv, red, ns):
        if recv.is_cuda:
            with torch.cuda.device(self.device):
                check(_lib.load().drq_sum_slices(ptr(recv), ns, self.world, ptr(red), ns,
                                                 torch.cuda.current_stream(self.device).cuda_stream), "drq_sum_slices")
        else:       # gloo rehearsals on CPU tensors (tests): the same rank-order sum
            acc = recv[:ns].clone()
            for r in range(1, self.world):
                acc += recv[r * ns:(r + 1) * ns]
            red.copy_(acc)

    def _zero1(self, g, p, m, v, lr, step, gscale, step_fn):
        """Runs on the current stream: gradient slices in, Adam on the owned slice, parameter slices out."""
        import torch.distributed as dist
        n = g.numel()
        ns = n // self.world
        recv, red = self._bufs(n)
        dist.all_to_all_single(recv, g, group=self.pg)
        lo = self.rank * ns
        ps, ms, vs = p[lo:lo + ns], m[lo:lo + ns], v[lo:lo + ns]
        if step_fn is not None:                     # CPU tests hand in the oracle's Adam
            step_fn(ps, recv.view(self.world, ns), ms, vs)
        else:
            with torch.cuda.device(self.device):
                check(_lib.load().drq_adam_reduce_flat(ptr(ps), ptr(recv), ns, self.world, ptr(ms), ptr(vs), ns,
                                                       float(lr), int(step), float(gscale),
                                                       torch.cuda.current_stream(self.device).cuda_stream),
                      "drq_adam_reduce_flat")
        red.copy_(ps)
        dist.all_gather_into_tensor(p, red, group=self.pg)

    def start_zero1(self, g, p, m, v, lr, step, gscale=1.0, step_fn=None):
        """Sharded optimiser step of one segment (g, p, m, v: the segment's views of the four arenas) once the work
        queued so far on the current stream has produced g; returns a handle whose wait() orders the current stream
        after the stepped parameters have arrived from every rank."""
        if not self.direct_ok(g.numel()):
            raise _lib.DrqError(f"zero1: a segment of {g.numel()} floats does not split over {self.world} ranks")
        if self.device.type != "cuda":
            self._zero1(g, p, m, v, lr, step, gscale, step_fn)
            return _Done()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._zero1(g, p, m, v, lr, step, gscale, step_fn)
        return _StreamWork(side, (g, p))

    def sharded(self, n):
        """Does the bucket of n floats take the sharded optimiser step?"""
        return self._pick(n) == "zero1"

    def _pick(self, n):
        if self.mode == "allreduce" or not self.direct_ok(n):
            return "allreduce"
        if self.mode in ("direct", "zero1"):
            return self.mode
        return self.choice.get(n, "allreduce")

    def start(self, t):
        """Begin SUM(t) over the ranks once the work queued so far on the current stream has produced t; returns
        a handle whose wait() orders the current stream after the exchange."""
        import torch.distributed as dist
        if self._pick(t.numel()) == "allreduce":
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if self.device.type != "cuda":
            self._direct(t)
            return _Done()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._direct(t)
        return _StreamWork(side, t)

    def calibrate(self, lengths, iters=20, warmup=5):
        """auto: time the candidates for each bucket length on scratch tensors; all ranks agree on the result (MAX over
        ranks of the per-update time).  allreduce / direct: the exchange plus the replicated Adam step it needs
        afterwards; zero1: the whole sharded sequence."""
        import time
        import torch.distributed as dist
        cuda = self.device.type == "cuda"
        for n in sorted(set(lengths)):
            if not self.direct_ok(n):
                self.choice[n] = "allreduce"
                continue
            t, pp, mm, vv = (torch.zeros(n, device=self.device, dtype=torch.float32) for _ in range(4))
            res = {}

            def adam_full():
                if cuda:
                    from . import ops
                    ops.adam_flat(pp, t, mm, vv, 1e-4, 1)

            cands = {"allreduce": lambda: (dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg), adam_full()),
                     "direct": lambda: (self._direct(t), adam_full())}
            if cuda:
                cands["zero1"] = lambda: self._zero1(t, pp, mm, vv, 1e-4, 1, 1.0, None)
            for name, fn in cands.items():
                for _ in range(warmup):
                    fn()
                if cuda:
                    torch.cuda.synchronize(self.device)
                dist.barrier(group=self.pg)
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                if cuda:
                    torch.cuda.synchronize(self.device)
                dt = torch.tensor([(time.perf_counter() - t0) / iters * 1e6], device=self.device, dtype=torch.float64)
                dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=self.pg)
                res[name] = float(dt.item())
            self.timings_us[n] = res
            self.choice[n] = min(res, key=res.get)
        return self.choice


class _Done:
    def wait(self):
        return None


class StepEngine:
    # rows per GPU and update: the Winograd / fused aug+conv1 kernels use 32-bit byte offsets into the first layer's
    # output [2B][32][41][41] fp32, < 2^31 bytes (conv_wino.hip: DRQ_EARG beyond); parity is pinned up to 2048 rows
    MAX_BATCH = 4096

    def __init__(self, encoder, actor, critic, critic_target, obs_shape, action_dim, feature_dim, hidden_dim, lr,
                 device):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.C, self.A, self.F, self.H = int(obs_shape[0]), int(action_dim), int(feature_dim), int(hidden_dim)
        # what the kernels are instantiated for (check_step in csrc/step.hip): said here, not at the first launch
        if self.C != 9 or tuple(obs_shape[1:]) != (84, 84):
            raise _lib.DrqError(f"obs_shape {tuple(obs_shape)}: the HIP encoder is built for (9, 84, 84) "
                                "(frame_stack=3, cfgs/config.yaml:7)")
        if not (0 < self.F <= 256) or self.A <= 0 or self.H <= 0:
            raise _lib.DrqError(f"feature_dim={self.F} (1..256), action_dim={self.A}, hidden_dim={self.H}: unsupported")
        self.layout = _lib.param_layout(self.C, self.A, self.F, self.H)
        total = self.layout["total"]
        dev = self.device
        self.params = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(total, device=dev, dtype=torch.float32)
        self.adam_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.adam_v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.modules = {"enc": encoder, "critic": critic, "actor": actor, "target": critic_target}
        for name, mod in self.modules.items():
            self._adopt(name, mod)
        seg = self.layout["seg"]
        mk = lambda n, mod: FlatAdam(mod.parameters(), lr, *(a[seg[n][0]:seg[n][1]] for a in
                                                            (self.params, self.grads, self.adam_m, self.adam_v)))
        self.encoder_opt = mk("enc", encoder)
        self.actor_opt = mk("actor", actor)
        self.critic_opt = mk("critic", critic)
        self._ws = None
        self._ws_B = None
        self._base = None
        self.sums = torch.zeros(8, device=dev, dtype=torch.float32)
        # pinned host mirror of the metric sums (DrqStep.sums_host): slot 8 = sequence word of the update
        self.sums_host = torch.zeros(16, dtype=torch.float32).pin_memory() if dev.type == "cuda" else None
        self._sums_seq = self.sums_host[8:9].view(torch.int32) if self.sums_host is not None else None
        self._last_seq = None
        self._enqueued = []             # sequence words of the last two updates handed to the GPU (_throttle)
        self.pg = None            # torch.distributed process group for data parallelism
        self._pending = None      # (all-reduce handle, descriptor, tensors kept alive) of a deferred Adam(actor)
        self._pending_enc = None  # the same for Adam(encoder)
        self.global_metrics = False   # data parallel: all-reduce the metric sums (else: this rank's shard)
        self._timing_array = None # ctypes array of hipEvent_t (set_timing_events)
        self._timing_n = 0
        self.step_flags = 0       # DrqStep.flags: 1 = no row fusion, 2 = no gemm3 (A/B measurements, both-form tests)
        self.metrics_spin = 256         # read_sums(): polls of the host mirror before the waiting thread starts to sleep
        self.metrics_wait_est = None    # read_sums(): running estimate (s) of how long the sums take to arrive
        self.profile_exchange = False   # bench.py: event pairs around every wait for a gradient exchange
        self.exchange_wait_us = {}      # bucket name -> [us the compute stream waited], see collect_exchange_waits()
        self._wait_events = []
        self._side = None         # side stream of the metric-sums exchange
        self._side_busy = False
        self.bf16 = False             # DrqStep.bf16: the update's convs / GEMMs on the bf16 MFMA (set_compute_dtype)
        self.store_aug_next = False   # verification: keep the next_obs view's encoder input too (DrqStep.store_aug_next)
        self.world = 1
        self.rank = 0
        self.exchange = None      # GradExchange (enable_data_parallel)
        self.fused_rng = True     # the update's four draws in one launch (rng_draws); None/False: torch's own four calls
        self._rng_ok = None       # verdict of _rng_selftest
        self._rng_bufs = None

    # ---- arenas --------------------------------------------------------------------------
    def _adopt(self, name, mod):
        """Move the module's parameters into the arena (parameters() order) and re-point them."""
        offs = self.layout[name]
        plist = list(mod.parameters())
        assert len(plist) == len(offs), (name, len(plist), len(offs))
        with torch.no_grad():
            for p, off in zip(plist, offs):
                n = p.numel()
                view = self.params[off:off + n].view(p.shape)
                view.copy_(p.data.to(torch.float32))
                p.data = view
                if name != "target":
                    p.grad = self.grads[off:off + n].view(p.shape)
        b, e = self.layout["seg"][name]
        mod._drq_segment = self.params[b:e]

    # ---- workspace -----------------------------------------------------------------------
    def _workspace(self, B):
        if self._ws is None or self._ws_B != B:
            lib = _lib.load()
            nbytes = lib.drq_step_ws_bytes(B, self.C, self.A, self.F, self.H)
            if nbytes == 0:
                raise _lib.DrqError("drq_step_ws_bytes: unsupported dimensions")
            self._ws = None
            self._ws = torch.zeros(nbytes // 4, device=self.device, dtype=torch.float32)  # zero: padded borders
            self._ws_B = B
        return self._ws

    def ws_view(self, name, B, shape):
        """Named workspace buffer as a tensor view (tests / tools)."""
        lib = _lib.load()
        off = lib.drq_step_ws_offset(B, self.C, self.A, self.F, self.H, _lib.WS_IDS.index(name))
        n = 1
        for s in shape:
            n *= s
        return self._workspace(B)[off:off + n].view(shape)

    def base_grid(self):
        if self._base is None:
            from .ops import aug_base_grid
            self._base = aug_base_grid(84, 4, self.device)
        return self._base

    # ---- data parallel -------------------------------------------------------------------
    def enable_data_parallel(self, process_group=None, global_metrics=False, exchange="allreduce"):
        """exchange: how the gradient buckets are summed over the ranks (GradExchange): "allreduce", "direct" or
        "auto" (measures both on the real bucket sizes now and keeps the faster)."""
        import torch.distributed as dist
        self.global_metrics = bool(global_metrics)
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        self.exchange = GradExchange(self.pg, self.world, self.device, exchange)
        if exchange == "auto" and self.world > 1:
            try:
                self.exchange.calibrate([e - b for b, e in grad_buckets_overlap(self.layout)])
            except RuntimeError as err:          # a backend without all-to-all: every bucket takes the all-reduce
                self.exchange.choice = {}
                self.exchange.timings_us = {"error": str(err)[:200]}

    def gather_optimizer_state(self):
        """Sharded optimiser (zero1): every rank keeps the Adam moments of its own slice of a segment only; before a
        snapshot the slices are exchanged so that adam_m / adam_v are complete everywhere.  No-op otherwise."""
        if self.exchange is None or self.world <= 1:
            return
        import torch.distributed as dist
        self.flush()
        for b, e in grad_buckets_overlap(self.layout):
            n = e - b
            if not self.exchange.sharded(n):
                continue
            ns = n // self.world
            for arena in (self.adam_m, self.adam_v):
                mine = arena[b + self.rank * ns: b + (self.rank + 1) * ns].clone()
                dist.all_gather_into_tensor(arena[b:e], mine, group=self.pg)

    def _allreduce_async(self, t):
        """SUM over the ranks that starts once the work queued so far has produced `t` and runs beside what is
        queued next (RCCL: its own stream; .wait() makes the current stream wait, not the host)."""
        return self.exchange.start(t)

    def set_timing_events(self, events):
        """bench.py instrumentation: 4..10 torch.cuda.Event(enable_timing=True) (pairs), each recorded once already (so
        that torch has created the hipEvent_t), or None.  See DrqStep.timing_events."""
        if events is None:
            self._timing_array = None
            return
        arr = (ctypes.c_void_p * len(events))(*[int(e.cuda_event) for e in events])
        self._timing_array = arr
        self._timing_n = len(events)

    # ---- the update's random draws ----------------------------------------------------------
    def _rng_launch(self, gen, n, A, pad_range, with_noise):
        """One launch for the draws, from the generator's current (seed, offset); advances the offset like the ATen
        launches it replaces.  Returns (shift_obs [n,1,1,2], shift_next, noise_critic [n,A], noise_actor)."""
        key = (n, A)
        if self._rng_bufs is None or self._rng_bufs[0] != key:
            mk = lambda *sh: torch.empty(sh, device=self.device, dtype=torch.float32)
            self._rng_bufs = (key, (mk(n, 1, 1, 2), mk(n, 1, 1, 2), mk(n, A), mk(n, A)))
        bufs = self._rng_bufs[1]
        seed, off = gen.initial_seed(), gen.get_offset()
        with torch.cuda.device(self.device):
            check(_lib.load().drq_rng_draws(seed, off, 2 * n, n * A if with_noise else 0, pad_range,
                                            *(ptr(b) for b in bufs), self._stream()), "drq_rng_draws")
        gen.set_offset(off + (16 if with_noise else 8))
        if not with_noise:                              # torch's own normal_ calls, in the reference's order
            bufs[2].normal_()
            bufs[3].normal_()
        return bufs

    def _rng_selftest(self, gen, n, A, pad_range):
        """Once per engine: what the fused launch reproduces of torch's own four calls, bit for bit, from the same
        generator state (the stream contract of SURVEY App. C).  "all": the four draws in one launch; "shifts": the
        two integer draws (exact by construction) in one launch, torch's normal_ for the noises -- the normal draws go
        through logf / sincosf, and a torch built against another device-library release rounds their last bit
        differently; None: torch's own calls for everything."""
        from torch.distributions.utils import _standard_normal
        state = gen.get_state()
        ref = [torch.randint(0, pad_range, size=(n, 1, 1, 2), device=self.device, dtype=torch.float32) for _ in range(2)]
        ref += [_standard_normal((n, A), dtype=torch.float32, device=self.device) for _ in range(2)]
        end_off = gen.get_offset()
        verdict = None
        for mode in ("all", "shifts"):
            gen.set_state(state)
            try:
                got = [t.clone() for t in self._rng_launch(gen, n, A, pad_range, mode == "all")]
            except _lib.DrqError:
                break
            if gen.get_offset() == end_off and all(torch.equal(a.view(-1), b.view(-1)) for a, b in zip(ref, got)):
                verdict = mode
                break
        gen.set_state(state)
        return verdict

    def rng_draws(self, n, A, pad_range):
        """The four draws of an update in the reference's order from torch's default generator of this device with
        fewer launches -- or None when that path is off, refused the size, or failed its self test (the caller then
        issues torch's own four calls).  The returned tensors are reused by the next update."""
        if not self.fused_rng or self.device.type != "cuda" or 2 * n > 65536 or n * A > 65536:
            return None
        gen = torch.cuda.default_generators[self.device.index]
        if self._rng_ok is None:
            self._rng_ok = self._rng_selftest(gen, n, A, pad_range) or False
        if not self._rng_ok:
            return None
        return self._rng_launch(gen, n, A, pad_range, self._rng_ok == "all")

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _phase(self, desc, k):
        # the library launches on the CURRENT HIP device: make that the agent's (an agent on cuda:1 in a process
        # whose current device is 0 would otherwise launch device-1 pointers on device 0)
        with torch.cuda.device(self.device):
            check(_lib.load().drq_update_phase(ctypes.byref(desc), k), f"drq_update_phase({k})")

    def _wait(self, work, name):
        """work.wait() (the current stream waits for the exchange); with profile_exchange the wait is bracketed by two
        events on the compute stream: their distance is the time the exchange was NOT hidden under compute."""
        if not self.profile_exchange or self.device.type != "cuda":
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        work.wait()
        e1.record()
        self._wait_events.append((name, e0, e1))

    def collect_exchange_waits(self):
        """After a synchronize: mean exposed microseconds per update and bucket since profile_exchange was switched on."""
        acc = {}
        for name, e0, e1 in self._wait_events:
            acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        self._wait_events = []
        return {k: {"mean_us": sum(v) / len(v), "max_us": max(v), "samples": len(v)} for k, v in acc.items()}

    def flush_encoder(self):
        """Data parallel: the deferred Adam(encoder) of the last update (phase 8).  Before the encoder is read."""
        if self._pending_enc is None:
            return
        work, desc, keep = self._pending_enc
        self._pending_enc = None
        self._wait(work, "encoder_grads")
        if desc is not None:                                    # replicated step (the sharded one came with the exchange)
            if self.device.type == "cuda":
                desc.stream = self._stream()
            self._phase(desc, 8)
        del keep

    def flush(self):
        """Data parallel: complete the deferred Adam(encoder) and Adam(actor) of the last update (their gradient
        all-reduces were left running).  Called before anything reads those weights: the next update, act(),
        snapshots."""
        self.flush_encoder()
        if self._pending is None:
            return
        work, desc, keep = self._pending
        self._pending = None
        self._wait(work, "actor_grads")
        if desc is not None:
            if self.device.type == "cuda":
                desc.stream = self._stream()
            self._phase(desc, 9)
        del keep

    # ---- the step ------------------------------------------------------------------------
    def make_desc(self, B_local, B_global, std, clip, tau, steps):
        if self.device.type != "cuda":
            raise _lib.DrqError("DrQV2Agent.update/act need the GPU: the HIP path has no CPU fallback")
        ws = self._workspace(B_local)
        d = DrqStep()
        d.B, d.global_B, d.C, d.A, d.F, d.H = B_local, B_global, self.C, self.A, self.F, self.H
        d.base_grid = ptr(self.base_grid())
        d.params, d.grads, d.adam_m, d.adam_v = ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v)
        d.ws, d.ws_bytes = ptr(ws), ws.numel() * 4
        d.sums = ptr(self.sums)
        d.lr, d.tau = float(self.critic_opt.lr), float(tau)
        d.std, d.clip = float(std), float(clip)
        d.step_critic, d.step_enc, d.step_actor = steps
        # local gradients are already scaled by 1/global_B (drq_td_mse / drq_actor_loss), so the SUM
        # all-reduce over ranks yields the global-batch mean gradient: no further scaling
        d.gscale = 1.0
        d.stream = self._stream()
        d.sums_host = ptr(self.sums_host) if (self.pg is None and self.sums_host is not None) else None
        d.store_aug_next = int(self.store_aug_next)
        d.bf16 = int(self.bf16)
        d.timing_events = self._timing_array      # None, or hipEvent_t pairs for bench.py's roofline
        d.timing_n = self._timing_n if self._timing_array is not None else 0
        d.flags = int(self.step_flags)
        # the zero borders of the padded gradient buffers lie elsewhere in the bf16 layouts (DrqStep.flags): a workspace
        # that served the other form is zeroed again
        mode = (bool(self.bf16), int(self.step_flags) & 12)
        if getattr(self, "_ws_mode", mode) != mode:
            ws.zero_()
        self._ws_mode = mode
        return d

    def update(self, obs, action, reward, discount, next_obs, shift_obs, shift_next, noise_critic, noise_actor, std,
               clip, tau, B_global=None, obs_index=None, next_obs_index=None):
        """All tensors are this rank's shard, on the GPU.  Returns the 8-float sums tensor (device).
        obs_index / next_obs_index (int64 [B], device): the batch is not materialised -- obs / next_obs are frame stores
        ([slots, C*84*84] or [slots, C, 84, 84] uint8) and row b is frame index[b] (DrqStep.obs_index)."""
        indexed = obs_index is not None
        if indexed:
            B = obs_index.numel()
            for nm, t, ix in (("obs", obs, obs_index), ("next_obs", next_obs, next_obs_index)):
                if (t.dtype != torch.uint8 or not t.is_contiguous() or t.numel() % (self.C * 84 * 84) or
                        ix is None or ix.dtype != torch.int64 or ix.numel() != B or not ix.is_contiguous()):
                    raise _lib.DrqError(f"update(): indexed {nm}: a contiguous uint8 frame store and {B} int64 indices")
                if not (ix.is_cuda or ix.is_pinned()):       # a pageable host pointer would fault on the GPU
                    raise _lib.DrqError(f"update(): indexed {nm}: the index list must be a device tensor or PINNED host "
                                        "memory")
        else:
            B = obs.shape[0]
        for nm, t in ((("obs", obs), ("next_obs", next_obs)) if not indexed else ()):
            if t.dtype != torch.uint8 or tuple(t.shape) != (B, self.C, 84, 84) or not t.is_contiguous():
                raise _lib.DrqError(f"update(): {nm} must be contiguous uint8 [{B},{self.C},84,84] "
                                    f"(replay_buffer.py:185-189), got {t.dtype} {tuple(t.shape)}")
        if B > self.MAX_BATCH:
            raise _lib.DrqError(f"update(): batch of {B} rows per GPU; the fp32 encoder kernels address an activation "
                                f"buffer with 32-bit byte offsets, which holds up to {self.MAX_BATCH} rows "
                                "(2B x 32 x 41 x 41 floats < 2 GiB): split the batch over more GPUs")
        B_global = B * self.world if B_global is None else B_global
        steps = (self.critic_opt.begin_step(), self.encoder_opt.begin_step(), self.actor_opt.begin_step())
        d = self.make_desc(B, B_global, std, clip, tau, steps)
        keep = (obs, action, reward, discount, next_obs, shift_obs, shift_next, noise_critic, noise_actor, obs_index,
                next_obs_index)
        d.obs, d.next_obs = ptr(obs), ptr(next_obs)
        d.obs_index, d.next_obs_index = ptr(obs_index), ptr(next_obs_index)
        d.action, d.reward, d.discount = ptr(action), ptr(reward), ptr(discount)
        d.shift_obs, d.shift_next = ptr(shift_obs), ptr(shift_next)
        d.noise_critic, d.noise_actor = ptr(noise_critic), ptr(noise_actor)
        if self.pg is None:
            self._throttle(steps[2] & 0xFFFFFFFF)
            self._phase(d, -1)
            self._last_seq = steps[2] & 0xFFFFFFFF
        else:
            # overlapped schedule.  Every exchange is a SUM of gradients already scaled by
            # 1/global_B; the order of the phases and their data dependencies are those of the 1-GPU step.
            # Every event recorded on the compute stream costs it a ~20 us bubble (system-scope release), so
            # the schedule hands over to RCCL only twice: after phase 4 and after phase 7.
            (c0, c1), (e0, e1), (a0, a1) = grad_buckets_overlap(self.layout)
            mirror = self.sums_host is not None
            seq = steps[2] & 0xFFFFFFFF
            if mirror:
                self._throttle(seq)                             # the host stays at most two updates ahead
            if self._side_busy:                                 # the previous update's sums exchange owns self.sums
                torch.cuda.current_stream(self.device).wait_stream(self._side)
                self._side_busy = False
            self.flush_encoder()                                # previous update's Adam(encoder)
            self._phase(d, 3)                                   # aug + encoder forward: no actor weights yet
            self.flush()                                        # previous update's Adam(actor), reduce done by now
            self._phase(d, 4)                                   # critic loss, backward to the encoder output
            ex = self.exchange
            z_critic = ex is not None and ex.sharded(c1 - c0)
            if z_critic:     # gradient slices -> Adam on the owned slice -> stepped parameters back, beside phase 5
                w_critic = ex.start_zero1(self.grads[c0:c1], self.params[c0:c1], self.adam_m[c0:c1], self.adam_v[c0:c1],
                                          d.lr, steps[0])
            else:
                w_critic = self._allreduce_async(self.grads[c0:c1])  # 16.7 MB, under the encoder backward
            self._phase(d, 5)
            self._wait(w_critic, "critic_grads")
            # phase 6 = critic_opt.step() + Polyak + the actor loss; with the sharded step already done: Polyak + loss
            ph6 = (lambda: (self._phase(d, 12), self._phase(d, 11))) if z_critic else (lambda: self._phase(d, 6))
            if not self.global_metrics:
                # metrics of THIS rank's shard (means over its rows): published by phase 6 itself, no exchange
                d.sums_host = ptr(self.sums_host) if mirror else None
                ph6()                                           # Adam(critic), actor loss (sums complete)
                self._last_seq = seq if mirror else None
            else:
                ph6()
                # global metric sums: reduced and published beside phase 7 (a side stream, so that the 32-byte
                # exchange's latency is not inserted between phases 6 and 7)
                if self.device.type == "cuda":
                    main = torch.cuda.current_stream(self.device)
                    side = self._side_stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        self._allreduce_async(self.sums).wait()
                        if mirror:
                            check(_lib.load().drq_publish_sums(ptr(self.sums), ptr(self.sums_host), seq,
                                                               side.cuda_stream), "drq_publish_sums")
                    self._side_busy = True
                    self._last_seq = seq if mirror else None
                else:
                    self._allreduce_async(self.sums).wait()
                    self._last_seq = None
            self._phase(d, 7)                                   # actor backward
            # encoder (0.1 MB) and actor (12.3 MB) gradients: reduced under the next update's encoder forward
            def hand_over(b, e, step_no):
                if ex is not None and ex.sharded(e - b):        # the optimiser step travels with the exchange
                    return (ex.start_zero1(self.grads[b:e], self.params[b:e], self.adam_m[b:e], self.adam_v[b:e], d.lr,
                                           step_no), None, keep)
                return (self._allreduce_async(self.grads[b:e]), d, keep)
            self._pending_enc = hand_over(e0, e1, steps[1])
            self._pending = hand_over(a0, a1, steps[2])
        del keep
        return self.sums

    # ---- the update in the reference's pieces (DrQV2Agent.encode / update_critic / update_actor) -------------------
    def begin_manual(self, obs, next_obs, shift_obs, shift_next, std, clip, tau):
        """aug + encoder forward of both views (phase 3) for an update issued piece by piece; returns the feature views
        (obs, next_obs) of the workspace.  Single GPU only."""
        if self.pg is not None:
            raise _lib.DrqError("update_critic / update_actor as separate calls are not available with data parallelism: "
                                "use update()")
        B = obs.shape[0]
        for nm, t in (("obs", obs), ("next_obs", next_obs)):
            if t.dtype != torch.uint8 or tuple(t.shape) != (B, self.C, 84, 84) or not t.is_contiguous():
                raise _lib.DrqError(f"encode(): {nm} must be contiguous uint8 [{B},{self.C},84,84], got {t.dtype} "
                                    f"{tuple(t.shape)}")
        if B > self.MAX_BATCH:
            raise _lib.DrqError(f"encode(): batch of {B} rows; at most {self.MAX_BATCH} per GPU")
        steps = (self.critic_opt.begin_step(), self.encoder_opt.begin_step(), self.actor_opt.begin_step())
        d = self.make_desc(B, B, std, clip, tau, steps)
        d.sums_host = None
        d.obs, d.next_obs = ptr(obs), ptr(next_obs)
        d.shift_obs, d.shift_next = ptr(shift_obs), ptr(shift_next)
        # not known yet and not read by phase 3 (drq_update_phase wants every pointer set)
        for f in ("action", "reward", "discount", "noise_critic", "noise_actor"):
            setattr(d, f, ptr(shift_obs))
        self._manual = {"d": d, "keep": [obs, next_obs, shift_obs, shift_next], "B": B, "stage": "encoded"}
        self._phase(d, 3)
        feat = self.ws_view("FEAT", B, (2 * B, 39200))
        return feat[:B], feat[B:]

    def manual_critic(self, feat_obs, action, reward, discount, feat_next, noise_critic):
        m = getattr(self, "_manual", None)
        if m is None or m["stage"] != "encoded":
            raise _lib.DrqError("update_critic(): call encode() for this batch first (the encoder backward needs the "
                                "activations that call saved)")
        d, B = m["d"], m["B"]
        feat = self.ws_view("FEAT", B, (2 * B, 39200))
        for dst, src, nm in ((feat[:B], feat_obs, "obs"), (feat[B:], feat_next, "next_obs")):
            if tuple(src.shape) != (B, 39200) or src.dtype != torch.float32 or src.device != self.device:
                raise _lib.DrqError(f"update_critic(): {nm} must be the [B, 39200] features encode() returned")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)                      # features the caller produced some other way: the heads use them
        m["keep"] += [action, reward, discount, noise_critic]
        d.action, d.reward, d.discount = ptr(action), ptr(reward), ptr(discount)
        d.noise_critic, d.noise_actor = ptr(noise_critic), ptr(noise_critic)   # the actor's draw comes with update_actor
        d.stream = self._stream()
        for k in (4, 5, 10, 8):
            self._phase(d, k)
        m["stage"] = "critic"
        return self.sums

    def manual_actor(self, noise_actor):
        m = getattr(self, "_manual", None)
        if m is None or m["stage"] != "critic":
            raise _lib.DrqError("update_actor(): call update_critic() for this batch first")
        d = m["d"]
        m["keep"].append(noise_actor)
        d.noise_actor = ptr(noise_actor)
        d.stream = self._stream()
        for k in (13, 11, 7, 9):
            self._phase(d, k)
        self._last_seq = None
        self._manual = None
        return self.sums

    def _throttle(self, seq):
        """Single GPU: at most two updates are queued behind the one the GPU works on.  A caller that never reads the
        metrics (use_tb=False, metrics_on_device) would otherwise run the host as far ahead as the HIP queue lets it --
        nothing gained (the GPU is the bottleneck), and every host-side staging buffer that feeds the updates (the
        device replay's pinned index sets, drqv2_amd/replay.py) would have to be as deep as that queue.  With the metrics
        read every update this never waits: the previous update has published by then.  `seq`: this update's
        sequence word (its actor step count), the mirror shows the last one published."""
        if self._sums_seq is None:
            return
        q = self._enqueued
        if q and ((q[-1] + 1) & 0xFFFFFFFF) != seq:          # step counts were set from outside (snapshot): start over
            q.clear()
        if len(q) >= 2:
            import time
            t0 = None
            while (int(self._sums_seq) & 0xFFFFFFFF) not in q:   # neither of the last two has published its sums yet
                if t0 is None:
                    t0 = time.monotonic()
                time.sleep(100e-6)
                if time.monotonic() - t0 > 30.0:
                    torch.cuda.synchronize()                     # surfaces a device fault
                    if (int(self._sums_seq) & 0xFFFFFFFF) not in q:
                        raise _lib.DrqError("update(): the updates queued before this one never published their sums")
        q.append(seq)
        del q[:-2]

    def read_sums(self):
        """The 8 metric sums of the last update as Python floats.  Single GPU: waits only until the update has
        published them (after the actor loss), not for the rest of the update still queued behind it."""
        if self._last_seq is None:
            if self._side_busy:
                self._side.synchronize()
            return self.sums.tolist()                      # no host mirror: drain the stream
        want = self._last_seq if self._last_seq < 2 ** 31 else self._last_seq - 2 ** 32
        seq = self._sums_seq
        import time
        # The wait is ~0.8 ms at batch 256 and the same from one update to the next, so the thread does not poll through
        # it: a short spin (the sums are there already when the host lags the GPU), then ONE sleep that ends a safe
        # margin before the sums are due (running estimate of the last waits), then polling for the last stretch.
        # Sleeping in short naps until the sums show up (the first version of this) woke up late often enough to cost
        # 3 % of the bench (a tenth of the updates 0.1 ms late: the next update reached the GPU after the previous one's
        # tail had drained); polling all the way keeps a host core busy that train.py's simulator wants.
        t_in = time.perf_counter()
        spins = 0
        while int(seq) != want and spins < self.metrics_spin:
            spins += 1
        if int(seq) != want:
            est = self.metrics_wait_est
            if est is not None and est > 300e-6:
                time.sleep(est - max(200e-6, 0.25 * est))
                if int(seq) == want:             # overslept (the estimate was too long): shorten it, no sample
                    self.metrics_wait_est = 0.8 * est
                    return self.sums_host[:8].tolist()
            limit = (est or 0.0) + 2e-3
            while int(seq) != want:
                waited = time.perf_counter() - t_in
                if waited > limit:               # unusually long (first update, another process on the GPU): nap
                    time.sleep(50e-6)
                    if waited > 30.0:
                        torch.cuda.synchronize()           # surfaces a device fault, if that is why nothing arrived
                        if int(seq) != want:
                            raise _lib.DrqError("metric sums of the update never arrived in the host mirror")
            waited = time.perf_counter() - t_in
            self.metrics_wait_est = waited if est is None else 0.75 * est + 0.25 * waited
        return self.sums_host[:8].tolist()

    def act_forward(self, obs_u8):
        """obs u8 [n,C,84,84] on the GPU -> mu [n,A]."""
        if obs_u8.dtype != torch.uint8 or not obs_u8.is_cuda or obs_u8.device != self.device:
            raise _lib.DrqError(f"act(): uint8 frames on {self.device} required (dmc.py:79-84 yields uint8), got "
                                f"{obs_u8.dtype} on {obs_u8.device}")
        if tuple(obs_u8.shape[1:]) != (self.C, 84, 84):
            raise _lib.DrqError(f"act(): frames of shape {(self.C, 84, 84)} required, got {tuple(obs_u8.shape[1:])}")
        self.flush()
        lib = _lib.load()
        n = obs_u8.shape[0]
        B = self._ws_B if (self._ws_B is not None and 2 * self._ws_B >= n) else max(1, (n + 1) // 2)
        d = self.make_desc(B, B, 1.0, 0.0, 0.0, (1, 1, 1))
        d.bf16 = 0                                  # acting stays in fp32 (DrqStep.bf16)
        mu = torch.empty((n, self.A), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(lib.drq_act_forward(ctypes.byref(d), ptr(obs_u8), n, ptr(mu)), "drq_act_forward")
        return mu
