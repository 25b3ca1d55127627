"""Drop-in `drqv2` module: the Hydra target `drqv2.DrQV2Agent` (cfgs/config.yaml:35 of the reference)
with the reference's constructor surface, backed by the MI355X HIP library (libdrqv2_hip.so).

Reference: /root/reference/drqv2.py.  Parity contract (tests/):
  * constructors keep names, argument order, sub-module names and state_dict keys
    (convnet.{0,2,4,6}, trunk.{0,1}, policy.{0,2,4}, Q{1,2}.{0,2,4}) and consume the global torch
    RNG exactly like the reference (same nn layers built in the same order, same init);
  * update() issues the four random draws of the reference in its order (drqv2.py:34-38 twice,
    utils.py:119 twice) and runs aug -> encoder -> critic step -> actor step -> target update in
    hand-written HIP kernels; there is no PyTorch/CPU fallback for that path;
  * metrics: the same 8 keys as python floats when use_tb, {} on gated-off steps.
"""
import math

import numpy as np
import torch
import torch.nn as nn

import utils
from drqv2_amd import _lib, ops
from drqv2_amd.engine import StepEngine, shard_bounds
from torch.distributions.utils import _standard_normal


class RandomShiftsAug(nn.Module):
    """Random shift by up to `pad` pixels: replicate pad + bilinear resample (drqv2.py:14-45)."""

    def __init__(self, pad):
        super().__init__()
        self.pad = pad

    def draw(self, n, device, dtype=torch.float32):
        # identical call to drqv2.py:34-38 -> identical shifts under the same seed/device
        return torch.randint(0, 2 * self.pad + 1, size=(n, 1, 1, 2), device=device, dtype=dtype)

    def forward(self, x):
        n, c, h, w = x.size()
        assert h == w
        shift = self.draw(n, x.device, torch.float32)
        return ops.random_shifts_aug(x.contiguous(), shift, self.pad)


class Encoder(nn.Module):
    def __init__(self, obs_shape):
        super().__init__()
        assert len(obs_shape) == 3
        self.repr_dim = 32 * 35 * 35
        layers = [nn.Conv2d(obs_shape[0], 32, 3, stride=2), nn.ReLU()]
        for _ in range(3):
            layers += [nn.Conv2d(32, 32, 3, stride=1), nn.ReLU()]
        self.convnet = nn.Sequential(*layers)
        self.apply(utils.weight_init)

    def forward(self, obs):
        """obs: uint8 or float [B,C,84,84] on the GPU -> [B, 39200] (inference path, drqv2.py:63-67)."""
        if obs.dtype == torch.uint8:
            x = ops.u8_normalize(obs.contiguous())
        else:
            x = (obs.float() / 255.0 - 0.5).contiguous()
        for li, i in enumerate((0, 2, 4, 6)):
            conv = self.convnet[i]
            x = ops.conv3x3_fwd(x, conv.weight.data, conv.bias.data, 2 if li == 0 else 1, relu=True)
        return x.view(x.shape[0], -1)


def _trunk(seq, obs):
    z = ops.linear_fwd(obs.contiguous(), seq[0].weight.data, seq[0].bias.data)
    return ops.ln_tanh_fwd(z, seq[1].weight.data, seq[1].bias.data, save=False)[0]


def _mlp3(seq, x):
    x = ops.linear_fwd(x, seq[0].weight.data, seq[0].bias.data, relu=True)
    x = ops.linear_fwd(x, seq[2].weight.data, seq[2].bias.data, relu=True)
    return ops.linear_fwd(x, seq[4].weight.data, seq[4].bias.data)


class Actor(nn.Module):
    def __init__(self, repr_dim, action_shape, feature_dim, hidden_dim):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(repr_dim, feature_dim), nn.LayerNorm(feature_dim), nn.Tanh())
        self.policy = nn.Sequential(nn.Linear(feature_dim, hidden_dim), nn.ReLU(inplace=True),
                                    nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True),
                                    nn.Linear(hidden_dim, action_shape[0]))
        self.apply(utils.weight_init)

    def forward(self, obs, std):
        mu = ops.tanh(_mlp3(self.policy, _trunk(self.trunk, obs)))
        return utils.TruncatedNormal(mu, torch.ones_like(mu) * std)


class Critic(nn.Module):
    def __init__(self, repr_dim, action_shape, feature_dim, hidden_dim):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(repr_dim, feature_dim), nn.LayerNorm(feature_dim), nn.Tanh())

        def q_head():
            return nn.Sequential(nn.Linear(feature_dim + action_shape[0], hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, 1))

        self.Q1 = q_head()
        self.Q2 = q_head()
        self.apply(utils.weight_init)

    def forward(self, obs, action):
        h_action = torch.cat([_trunk(self.trunk, obs), action], dim=-1).contiguous()
        return _mlp3(self.Q1, h_action), _mlp3(self.Q2, h_action)


class DrQV2Agent:
    def __init__(self, obs_shape, action_shape, device, lr, feature_dim, hidden_dim, critic_target_tau,
                 num_expl_steps, update_every_steps, stddev_schedule, stddev_clip, use_tb):
        self._init_kwargs = dict(obs_shape=tuple(obs_shape), action_shape=tuple(action_shape), device=device, lr=lr,
                                 feature_dim=feature_dim, hidden_dim=hidden_dim,
                                 critic_target_tau=critic_target_tau, num_expl_steps=num_expl_steps,
                                 update_every_steps=update_every_steps, stddev_schedule=stddev_schedule,
                                 stddev_clip=stddev_clip, use_tb=use_tb)
        self.device = device
        self.critic_target_tau = critic_target_tau
        self.update_every_steps = update_every_steps
        self.use_tb = use_tb
        self.num_expl_steps = num_expl_steps
        self.stddev_schedule = stddev_schedule
        self.stddev_clip = stddev_clip

        # models: same construction order as drqv2.py:137-145 (RNG stream parity)
        self.encoder = Encoder(obs_shape).to(device)
        self.actor = Actor(self.encoder.repr_dim, action_shape, feature_dim, hidden_dim).to(device)
        self.critic = Critic(self.encoder.repr_dim, action_shape, feature_dim, hidden_dim).to(device)
        self.critic_target = Critic(self.encoder.repr_dim, action_shape, feature_dim, hidden_dim).to(device)
        self.critic_target.load_state_dict(self.critic.state_dict())

        # arenas + fused optimisers (replace the three torch.optim.Adam of drqv2.py:148-150)
        self._engine = StepEngine(self.encoder, self.actor, self.critic, self.critic_target, obs_shape,
                                  action_shape[0], feature_dim, hidden_dim, lr, device)
        self.encoder_opt = self._engine.encoder_opt
        self.actor_opt = self._engine.actor_opt
        self.critic_opt = self._engine.critic_opt

        self.aug = RandomShiftsAug(pad=4)
        self._draw_hook = None     # tests inject the four random draws here
        # False: update() returns python floats like the reference (its .item() calls, drqv2.py:191-198,223-226).
        # True: it returns 0-d DEVICE tensors and never waits for the GPU; Logger.log() calls .item() on tensors
        # itself (logger.py:143-144), so train.py works unchanged and the wait moves to the logging cadence.
        self.metrics_on_device = False

        self.train()
        self.critic_target.train()

    def set_compute_dtype(self, dtype):
        """"fp32" (default: the reference's arithmetic) or "bf16" (BASELINE configs[4]; new functionality): the update's
        conv2..4 and nn.Linear GEMMs run on the bf16 MFMA with fp32 accumulation and fp32 storage (DrqStep.bf16)."""
        name = {"fp32": "fp32", "f32": "fp32", "float32": "fp32", torch.float32: "fp32",
                "bf16": "bf16", "bfloat16": "bf16", torch.bfloat16: "bf16"}.get(dtype)
        if name is None:
            raise ValueError(f"compute dtype {dtype!r}: 'fp32' or 'bf16'")
        self._engine.bf16 = name == "bf16"
        return self

    def train(self, training=True):
        self.training = training
        self.encoder.train(training)
        self.actor.train(training)
        self.critic.train(training)

    # ---- drqv2.py:164-175 ----------------------------------------------------------------------
    def act(self, obs, step, eval_mode):
        obs = torch.as_tensor(obs, device=self.device)
        mu = self._engine.act_forward(obs.unsqueeze(0).contiguous())
        stddev = utils.schedule(self.stddev_schedule, step)
        dist = utils.TruncatedNormal(mu, torch.ones_like(mu) * stddev)
        if eval_mode:
            action = dist.mean
        else:
            action = dist.sample(clip=None)
            if step < self.num_expl_steps:
                action.uniform_(-1.0, 1.0)
        return action.cpu().numpy()[0]

    # ---- data parallel (new: one process per GPU, RCCL all-reduce of the flat gradient arenas) ---
    def enable_data_parallel(self, process_group=None, batch_is_global=True, global_metrics=False,
                             exchange="allreduce"):
        """batch_is_global: every rank's replay_iter yields the same global batch and this rank trains
        on its contiguous slice; otherwise the iterator already yields this rank's shard.
        global_metrics: the returned metrics are means over the GLOBAL batch (one more 32-byte all-reduce per
        update); by default they are the means over this rank's shard (the gradients are always global).
        exchange: "allreduce" | "direct" | "zero1" | "auto" -- drqv2_amd.engine.GradExchange ("zero1": Adam sharded over the
        ranks, each steps 1/world of every segment and the stepped parameters are all-gathered)."""
        self._engine.enable_data_parallel(process_group, global_metrics, exchange)
        self._batch_is_global = batch_is_global

    def _draws(self, n_global, A):
        if self._draw_hook is not None:
            return self._draw_hook(n_global, A)
        dev = self.device
        fused = self._engine.rng_draws(n_global, A, 2 * self.aug.pad + 1)           # the same four draws, one launch
        if fused is not None:
            return fused
        sh_o = self.aug.draw(n_global, dev)                                        # RNG draw 1 (drqv2.py:241)
        sh_n = self.aug.draw(n_global, dev)                                        # RNG draw 2 (:242)
        n_c = _standard_normal((n_global, A), dtype=torch.float32, device=dev)     # draw 3 (:183)
        n_a = _standard_normal((n_global, A), dtype=torch.float32, device=dev)     # draw 4 (:211)
        return sh_o, sh_n, n_c, n_a

    # ---- drqv2.py:230-262 ----------------------------------------------------------------------
    def update(self, replay_iter, step):
        metrics = dict()
        if step % self.update_every_steps != 0:
            return metrics

        batch = next(replay_iter)
        frames = getattr(batch, "frames", None)      # drqv2_amd.replay.IndexedBatch: obs / next_obs are indices into it
        obs, action, reward, discount, next_obs = utils.to_torch(batch, self.device)
        eng = self._engine
        A = eng.A
        world, rank = eng.world, eng.rank
        big = getattr(self, "_batch_is_global", True)
        lo, hi, n_global = shard_bounds(obs.shape[0], world, rank, big)
        if world > 1 and big:
            obs_l, action_l, reward_l, discount_l, next_l = (t[lo:hi] for t in (obs, action, reward, discount,
                                                                                 next_obs))
        else:
            obs_l, action_l, reward_l, discount_l, next_l = obs, action, reward, discount, next_obs
        # every rank draws for the GLOBAL batch and keeps its slice: same numbers as the 1-GPU run
        sh_o, sh_n, n_c, n_a = (t.reshape(n_global, -1)[lo:hi].contiguous() for t in self._draws(n_global, A))

        stddev = utils.schedule(self.stddev_schedule, step)
        f32 = lambda t: t.to(torch.float32).contiguous()
        if frames is not None:
            sums = eng.update(frames, f32(action_l), f32(reward_l).view(-1), f32(discount_l).view(-1), frames, f32(sh_o),
                              f32(sh_n), f32(n_c), f32(n_a), stddev, self.stddev_clip, self.critic_target_tau,
                              B_global=n_global, obs_index=obs_l.contiguous(), next_obs_index=next_l.contiguous())
        else:
            sums = eng.update(obs_l.contiguous(), f32(action_l), f32(reward_l).view(-1), f32(discount_l).view(-1),
                              next_l.contiguous(), f32(sh_o), f32(sh_n), f32(n_c), f32(n_a), stddev, self.stddev_clip,
                              self.critic_target_tau, B_global=n_global)

        # the update is queued: the next batch's host work goes here, beside the GPU (drqv2_amd.replay.BatchIterator)
        ahead = getattr(replay_iter, "prefetch", None)
        if ahead is not None:
            ahead()
        if self.use_tb and self.metrics_on_device:
            inv = 1.0 / (n_global if (eng.pg is None or eng.global_metrics) else (hi - lo))
            if eng._side_busy:           # data parallel with global metrics: the sums exchange runs on a side stream
                torch.cuda.current_stream(eng.device).wait_stream(eng._side)
            vals = sums[:7] * inv        # a fresh tensor: the sums buffer is overwritten by the next update
            for i, k in enumerate(("batch_reward", "critic_target_q", "critic_q1", "critic_q2", "critic_loss",
                                   "actor_loss", "actor_logprob")):
                metrics[k] = vals[i]
            metrics["actor_ent"] = torch.tensor(A * (0.5 + 0.5 * math.log(2 * math.pi) + math.log(stddev)),
                                                dtype=torch.float32)
        elif self.use_tb:
            s = eng.read_sums()          # the single device->host wait of the update
            # data parallel without global_metrics: the sums cover this rank's rows only
            inv = 1.0 / (n_global if (eng.pg is None or eng.global_metrics) else (hi - lo))
            metrics["batch_reward"] = s[0] * inv
            metrics["critic_target_q"] = s[1] * inv
            metrics["critic_q1"] = s[2] * inv
            metrics["critic_q2"] = s[3] * inv
            metrics["critic_loss"] = s[4] * inv
            metrics["actor_loss"] = s[5] * inv
            metrics["actor_logprob"] = s[6] * inv
            metrics["actor_ent"] = A * (0.5 + 0.5 * math.log(2 * math.pi) + math.log(stddev))
        return metrics

    # ---- the update in the reference's pieces (drqv2.py:177-228, :241-246) -----------------------------------------
    def encode(self, obs, next_obs, step):
        """aug + encoder of both views as update() does (drqv2.py:241-246), for callers that issue update_critic /
        update_actor themselves: obs / next_obs uint8 [B,C,84,84]; returns the features (obs, next_obs) [B, 39200], views
        of the step workspace.  Consumes the two shift draws of the reference.  The encoder's backward inside
        update_critic uses the activations THIS call saved (the autograd graph of the reference's `self.encoder(obs)`)."""
        obs, next_obs = (torch.as_tensor(t, device=self.device).contiguous() for t in (obs, next_obs))
        n = obs.shape[0]
        sh_o, sh_n = self.aug.draw(n, self.device), self.aug.draw(n, self.device)      # RNG draws 1, 2
        stddev = utils.schedule(self.stddev_schedule, step)
        return self._engine.begin_manual(obs, next_obs, sh_o.reshape(n, 2).contiguous(), sh_n.reshape(n, 2).contiguous(),
                                         stddev, self.stddev_clip, self.critic_target_tau)

    def update_critic(self, obs, action, reward, discount, next_obs, step):
        """drqv2.py:177-204.  obs / next_obs: the features encode() returned for this batch.  Issues the critic loss, its
        backward through the encoder, critic_opt.step() and encoder_opt.step(); consumes the reference's third draw
        (utils.py:119).  `step` must be the step encode() was called with (the stddev schedule is evaluated there)."""
        metrics = dict()
        f32 = lambda t: torch.as_tensor(t, device=self.device).to(torch.float32).contiguous()
        n = obs.shape[0]
        n_c = _standard_normal((n, self._engine.A), dtype=torch.float32, device=self.device)   # draw 3 (:183)
        sums = self._engine.manual_critic(obs, f32(action), f32(reward).view(-1), f32(discount).view(-1), next_obs, n_c)
        if self.use_tb:
            s = sums.tolist()
            inv = 1.0 / n
            metrics["critic_target_q"] = s[1] * inv
            metrics["critic_q1"] = s[2] * inv
            metrics["critic_q2"] = s[3] * inv
            metrics["critic_loss"] = s[4] * inv
        return metrics

    def update_actor(self, obs, step):
        """drqv2.py:206-228, after update_critic of the same batch.  Consumes the reference's fourth draw."""
        metrics = dict()
        m = getattr(self._engine, "_manual", None)
        if m is None:
            raise _lib.DrqError("update_actor(): call update_critic() for this batch first")
        n, A = m["B"], self._engine.A
        n_a = _standard_normal((n, A), dtype=torch.float32, device=self.device)               # draw 4 (:211)
        stddev = utils.schedule(self.stddev_schedule, step)
        sums = self._engine.manual_actor(n_a)
        if self.use_tb:
            s = sums.tolist()
            inv = 1.0 / n
            metrics["actor_loss"] = s[5] * inv
            metrics["actor_logprob"] = s[6] * inv
            metrics["actor_ent"] = A * (0.5 + 0.5 * math.log(2 * math.pi) + math.log(stddev))
        return metrics

    # ---- snapshots: train.py:192-204 pickles the whole agent --------------------------------
    def flush(self):
        """Data parallel only: finish the Adam(actor) step the last update() left overlapped with its gradient
        all-reduce.  update(), act() and snapshots do this themselves; call it before reading actor weights
        directly."""
        self._engine.flush()

    # ---- weights-only interchange with the reference's classes -----------------------------
    def export_reference_state(self):
        """The agent as plain tensors in the formats the REFERENCE's objects load with their own methods:
        `state_dict()`s of the four nn.Modules (same keys: drqv2.py:48-121) and `torch.optim.Adam.state_dict()`s of the
        three optimisers (drqv2.py:148-150: per-parameter `step`, `exp_avg`, `exp_avg_sq`, one param group).
        No code is pickled: `torch.save(agent.export_reference_state(), f)` loads with `weights_only=True`, and a
        reference agent takes it with `agent.encoder.load_state_dict(s["encoder"])` ...
        `agent.encoder_opt.load_state_dict(s["encoder_opt"])`."""
        self._engine.flush()
        self._engine.gather_optimizer_state()
        cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
        out = {"format": "drqv2-reference-state-v1",
               "encoder": cpu(self.encoder.state_dict()), "actor": cpu(self.actor.state_dict()),
               "critic": cpu(self.critic.state_dict()), "critic_target": cpu(self.critic_target.state_dict())}
        eng = self._engine
        for net, opt_name, mod in (("enc", "encoder_opt", self.encoder), ("actor", "actor_opt", self.actor),
                                   ("critic", "critic_opt", self.critic)):
            opt = getattr(self, opt_name)
            state = {}
            plist = list(mod.parameters())
            if opt.t > 0:
                for i, (p, off) in enumerate(zip(plist, eng.layout[net])):
                    n = p.numel()
                    state[i] = {"step": torch.tensor(float(opt.t)),
                                "exp_avg": eng.adam_m[off:off + n].view(p.shape).detach().cpu().clone(),
                                "exp_avg_sq": eng.adam_v[off:off + n].view(p.shape).detach().cpu().clone()}
            group = dict(lr=opt.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False,
                         foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                         params=list(range(len(plist))))
            out[opt_name] = {"state": state, "param_groups": [group]}
        return out

    def import_reference_state(self, st):
        """The reverse: weights, target and Adam state from `export_reference_state()`-shaped data, e.g. what a
        reference agent produces with {name: module.state_dict()} and {name: optimiser.state_dict()}."""
        # data parallel: the deferred Adam(encoder/actor) of the last update must land BEFORE the imported weights do
        self._engine.flush()
        for n in ("encoder", "actor", "critic", "critic_target"):
            getattr(self, n).load_state_dict(st[n])
        eng = self._engine
        for net, opt_name, mod in (("enc", "encoder_opt", self.encoder), ("actor", "actor_opt", self.actor),
                                   ("critic", "critic_opt", self.critic)):
            mine = getattr(self, opt_name)
            ref = st[opt_name]["state"]
            t = 0
            for i, (p, off) in enumerate(zip(mod.parameters(), eng.layout[net])):
                ent = ref.get(i, ref.get(str(i)))
                if ent is None:
                    # no state for this parameter in the reference optimiser (it has not stepped): fresh moments, not
                    # whatever this agent had accumulated (bias correction restarts at t = 0)
                    eng.adam_m[off:off + p.numel()].zero_()
                    eng.adam_v[off:off + p.numel()].zero_()
                    continue
                m, v = ent["exp_avg"], ent["exp_avg_sq"]
                eng.adam_m[off:off + m.numel()].copy_(m.reshape(-1))
                eng.adam_v[off:off + v.numel()].copy_(v.reshape(-1))
                t = max(t, int(ent["step"]))
            mine.t = t
            mine.param_groups[0]["lr"] = st[opt_name]["param_groups"][0]["lr"]

    def __getstate__(self):
        self._engine.flush()
        self._engine.gather_optimizer_state()       # sharded optimiser: every rank's snapshot holds complete moments
        cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
        return {"init": self._init_kwargs, "training": self.training,
                "encoder": cpu(self.encoder.state_dict()), "actor": cpu(self.actor.state_dict()),
                "critic": cpu(self.critic.state_dict()), "critic_target": cpu(self.critic_target.state_dict()),
                "opt": {n: getattr(self, n).export_state() for n in ("encoder_opt", "actor_opt", "critic_opt")}}

    def _load_reference_state(self, st):
        """`st` is the __dict__ of an agent pickled by the REFERENCE's class (train.py:192-198 pickles the object and
        the reference defines no __getstate__): nn.Modules, torch.optim.Adam objects, scalar attributes.  With this
        module on the path pickle resolves `drqv2.DrQV2Agent` to this class and hands that dict over here; the agent
        is rebuilt on the arenas, weights and Adam moments are copied in."""
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        sds = {n: cpu(st[n]) for n in ("encoder", "actor", "critic", "critic_target")}
        a = sds["actor"]
        dev = st.get("device", "cuda")
        if str(dev).startswith("cuda") and not torch.cuda.is_available():
            dev = "cpu"
        kw = dict(obs_shape=(sds["encoder"]["convnet.0.weight"].shape[1], 84, 84),
                  action_shape=(a["policy.4.weight"].shape[0],), device=dev,
                  lr=st["actor_opt"].param_groups[0]["lr"], feature_dim=a["trunk.0.weight"].shape[0],
                  hidden_dim=a["policy.0.weight"].shape[0], critic_target_tau=st["critic_target_tau"],
                  num_expl_steps=st["num_expl_steps"], update_every_steps=st["update_every_steps"],
                  stddev_schedule=st["stddev_schedule"], stddev_clip=st["stddev_clip"], use_tb=st["use_tb"])
        with torch.random.fork_rng(devices=[]):
            self.__init__(**kw)
        for n, sd in sds.items():
            getattr(self, n).load_state_dict(sd)
        eng = self._engine
        for net, opt_name in (("enc", "encoder_opt"), ("actor", "actor_opt"), ("critic", "critic_opt")):
            ref = st[opt_name].state_dict()["state"]          # param index -> {step, exp_avg, exp_avg_sq}
            mine = getattr(self, opt_name)
            t = 0
            for i, off in enumerate(eng.layout[net]):
                if i not in ref:
                    continue
                m, v = ref[i]["exp_avg"], ref[i]["exp_avg_sq"]
                eng.adam_m[off:off + m.numel()].copy_(m.reshape(-1))
                eng.adam_v[off:off + v.numel()].copy_(v.reshape(-1))
                t = max(t, int(ref[i]["step"]))
            mine.t = t
        self.train(bool(st.get("training", True)))

    def __setstate__(self, st):
        if "init" not in st:
            return self._load_reference_state(st)
        kw = dict(st["init"])
        if str(kw["device"]).startswith("cuda") and not torch.cuda.is_available():
            kw["device"] = "cpu"
        with torch.random.fork_rng(devices=[]):      # building the nets must not move the global RNG
            self.__init__(**kw)
        self.encoder.load_state_dict(st["encoder"])
        self.actor.load_state_dict(st["actor"])
        self.critic.load_state_dict(st["critic"])
        self.critic_target.load_state_dict(st["critic_target"])
        for n, s in st["opt"].items():
            getattr(self, n).import_state(s)
        self.train(st["training"])
