"""The two pieces of `isaacgymenvs.utils.rlgames_utils` the reference's train.py uses (train.py:91,122-150,167),
over the native tasks: `get_rlgames_env_creator` and `RLGPUEnv`.

    # reference train.py:91 becomes
    from isaacgym_amd.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    # train.py:147-150 and 167 stay as they are:
    vecenv.register('RLGPU', lambda config_name, num_actors, **kwargs: RLGPUEnv(config_name, num_actors, **kwargs))
    env_configurations.register('rlgpu', {'vecenv_type': 'RLGPU', 'env_creator': lambda **kwargs: create_env_thunk(**kwargs)})

rl_games itself is not a dependency of this repository (it is not installed in the build image): when it is
importable RLGPUEnv derives from its `vecenv.IVecEnv`, otherwise from `object`, with the same methods.
"""
import os

from . import scene
from .tasks import isaacgym_task_map

try:   # pragma: no cover - rl_games is absent offline
    from rl_games.common import env_configurations, vecenv
    _IVecEnv = vecenv.IVecEnv
except Exception:   # noqa: BLE001
    env_configurations = None
    _IVecEnv = object


def get_rlgames_env_creator(seed, task_config, task_name, sim_device, rl_device, graphics_device_id=-1, headless=True, multi_gpu=False,
                            post_create_hook=None, virtual_screen_capture=False, force_render=False):
    """Same signature and behaviour as upstream: returns a thunk that builds the task.  With multi_gpu the rank comes from
    LOCAL_RANK / RANK (reference train.py:117-120): own env shard on own GPU, global env ids offset by the rank."""
    def create_rlgpu_env():
        cfg = task_config
        _sim, _rl = sim_device, rl_device
        if multi_gpu:
            local_rank, rank = int(os.getenv("LOCAL_RANK", "0")), int(os.getenv("RANK", "0"))
            _sim = _rl = f"cuda:{local_rank}"
            cfg = dict(cfg)
            cfg["env_id_offset"] = rank * int(cfg["env"]["numEnvs"])
            cfg["rank"] = rank
        cfg = dict(cfg)
        cfg.setdefault("seed", int(seed) if seed is not None else 0)
        if "scene" not in cfg or "sim" not in cfg:   # a cfg straight from the reference yaml has neither our scene block nor resolved sim
            defaults = scene.default_task_cfg(scene.TASK_VARIANTS[task_name])
            cfg.setdefault("scene", defaults["scene"])
            cfg.setdefault("sim", defaults["sim"])
        env = isaacgym_task_map[task_name](cfg, _rl, _sim, graphics_device_id, headless, virtual_screen_capture, force_render)
        if post_create_hook is not None:
            post_create_hook()
        return env
    return create_rlgpu_env


class RLGPUEnv(_IVecEnv):
    """rl_games vec-env wrapper, method for method as upstream's."""

    def __init__(self, config_name, num_actors, **kwargs):
        if env_configurations is not None and config_name in getattr(env_configurations, "configurations", {}):
            self.env = env_configurations.configurations[config_name]["env_creator"](**kwargs)
        else:   # no rl_games registry: `env_creator` may be passed directly
            self.env = kwargs.pop("env_creator")(**kwargs)

    def step(self, actions):
        return self.env.step(actions)

    def reset(self):
        return self.env.reset()

    def reset_done(self):
        return self.env.reset_done()

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.env.num_states > 0:
            info["state_space"] = self.env.state_space
        return info
