"""isaacgym_amd — MI355X-native vectorised HumanoidPingpong environment (the VecTask hot path of mjmj531/isaacgym).

`make()` mirrors the reference's `isaacgymenvs.make` (reference __init__.py:14-55) for the pingpong tasks.
Importing this package does not need a GPU; creating an environment does (there is no CPU fallback).
"""
from . import scene  # noqa: F401

__all__ = ["make", "scene"]


def make(seed=0, task="HumanoidPingpongTiltG1", num_envs=None, sim_device="cuda:0", rl_device="cuda:0", graphics_device_id=-1,
         headless=True, multi_gpu=False, virtual_screen_capture=False, force_render=False, cfg=None):
    """Create a task by its reference name.  With multi_gpu=True the rank comes from LOCAL_RANK/RANK (reference
    train.py:117-120): each rank owns its own env shard on its own GPU.  The seed is used as given — train.py:119 has already
    added the rank to the seed it passes in — and the shard is told apart by its global env-id offset, which keys the reset
    draws: a sharded run with one common seed reproduces the single-handle run env for env (isaacgym_amd.distributed)."""
    import os

    from .tasks import isaacgym_task_map
    variant = scene.TASK_VARIANTS[task]
    cfg = scene.default_task_cfg(variant) if cfg is None else cfg
    if num_envs is not None:
        cfg["env"]["numEnvs"] = int(num_envs)
    if multi_gpu:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        rank = int(os.environ.get("RANK", "0"))
        sim_device = rl_device = f"cuda:{local_rank}"
        cfg["seed"] = int(seed)
        cfg["env_id_offset"] = rank * int(cfg["env"]["numEnvs"])
    else:
        cfg.setdefault("seed", int(seed))
    return isaacgym_task_map[task](cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture, force_render)
