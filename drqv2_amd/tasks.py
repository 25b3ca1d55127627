"""Hyper-parameter surface of the reference (cfgs/config.yaml + cfgs/task/*.yaml) as data, for benches
and tests that run without Hydra.  With Hydra installed the reference's own cfgs/ tree is used unchanged:
its `_target_: drqv2.DrQV2Agent` (config.yaml:35) resolves to this repo's drqv2.py."""

BASE = dict(frame_stack=3, action_repeat=2, discount=0.99, num_seed_frames=4000, eval_every_frames=10000,
            num_eval_episodes=10, save_snapshot=False, replay_buffer_size=1000000, replay_buffer_num_workers=4,
            nstep=3, batch_size=256, seed=1, device="cuda", save_video=True, save_train_video=False, use_tb=True,
            experiment="exp", lr=1e-4, feature_dim=50)
AGENT = dict(_target_="drqv2.DrQV2Agent", critic_target_tau=0.01, update_every_steps=2, num_expl_steps=2000,
             hidden_dim=1024, stddev_clip=0.3)
# difficulty tiers (cfgs/task/{easy,medium,hard}.yaml): training budget and exploration-noise decay
TIERS = {"easy": (1100000, "linear(1.0,0.1,100000)"), "medium": (3100000, "linear(1.0,0.1,500000)"),
         "hard": (30100000, "linear(1.0,0.1,2000000)")}
_HUM = dict(lr=8e-5, feature_dim=100)
_WALK = dict(nstep=1, batch_size=512)
TASKS = {
    "acrobot_swingup": ("medium", {}), "cartpole_balance": ("easy", {}), "cartpole_balance_sparse": ("easy", {}),
    "cartpole_swingup": ("easy", {}), "cartpole_swingup_sparse": ("medium", {}), "cheetah_run": ("medium", {}),
    "cup_catch": ("easy", {}), "finger_spin": ("easy", {}), "finger_turn_easy": ("medium", {}),
    "finger_turn_hard": ("medium", {}), "hopper_hop": ("medium", {}), "hopper_stand": ("easy", {}),
    "humanoid_run": ("hard", _HUM), "humanoid_stand": ("hard", _HUM), "humanoid_walk": ("hard", _HUM),
    "pendulum_swingup": ("easy", {}), "quadruped_run": ("medium", dict(replay_buffer_size=100000)),
    "quadruped_walk": ("medium", {}), "reach_duplo": ("medium", {}), "reacher_easy": ("medium", {}),
    "reacher_hard": ("medium", {}), "walker_run": ("medium", _WALK), "walker_stand": ("easy", _WALK),
    "walker_walk": ("easy", _WALK),
}
# action dimensions of the dm_control tasks (not in the reference's YAML)
ACTION_DIM = {"cartpole_swingup": 1, "cheetah_run": 6, "quadruped_walk": 12, "quadruped_run": 12,
              "humanoid_run": 21, "humanoid_walk": 21, "humanoid_stand": 21, "walker_run": 6, "walker_walk": 6}


def resolve(task):
    tier, over = TASKS[task]
    cfg = dict(BASE)
    cfg["num_train_frames"], cfg["stddev_schedule"] = TIERS[tier]
    cfg.update(over)
    cfg["task_name"] = task
    cfg["agent"] = dict(AGENT, device=cfg["device"], lr=cfg["lr"], use_tb=cfg["use_tb"],
                        feature_dim=cfg["feature_dim"], stddev_schedule=cfg["stddev_schedule"])
    return cfg


def agent_kwargs(task, obs_shape, action_shape, device=None):
    """kwargs for drqv2.DrQV2Agent, as hydra.utils.instantiate(cfg.agent) would pass them (train.py:28-31)."""
    a = dict(resolve(task)["agent"])
    a.pop("_target_")
    a.update(obs_shape=tuple(obs_shape), action_shape=tuple(action_shape))
    if device is not None:
        a["device"] = device
    return a
