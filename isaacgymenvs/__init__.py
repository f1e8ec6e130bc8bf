"""`isaacgymenvs`-named shim: the names the reference's training entry imports (train.py:80-99, __init__.py:14-55), over the
MI355X-native HumanoidPingpong tasks of `isaacgym_amd`.  With this directory (and the `isaacgym` marker next to it) on the path,
train.py:122-167 runs unchanged up to the rl_games Runner (rl_games / hydra themselves are not part of this repository).

    import isaacgymenvs
    envs = isaacgymenvs.make(seed, "HumanoidPingpongTiltG1", num_envs, "cuda:0", "cuda:0", cfg=cfg)   # reference __init__.py:14

`cfg` is whatever train.py holds: an OmegaConf DictConfig when omegaconf is installed, or the plain nested dict
`isaacgymenvs.compose(task)` returns (the reference's cfg/task/*.yaml read as plain YAML with their ${resolve_default:...}
interpolations resolved — isaacgym_amd.cfgyaml); None composes the task's yaml from `cfg_dir`.
"""
import os

from isaacgym_amd import cfgyaml
from isaacgymenvs.utils.reformat import omegaconf_to_dict

# where cfg/task/<name>.yaml lives: the reference checkout when there is one, else the defaults compiled into isaacgym_amd.scene
CFG_DIR = os.environ.get("ISAACGYMENVS_CFG_DIR", "/root/reference/cfg")


def compose(task, overrides=None, cfg_dir=None):
    """What `hydra.compose(config_name="config", overrides=[f"task={task}"])` yields for the reference's files (a nested dict)."""
    return cfgyaml.compose(task, cfg_dir or CFG_DIR, overrides)


def make(seed, task, num_envs, sim_device, rl_device, graphics_device_id=-1, headless=False, multi_gpu=False, virtual_screen_capture=False,
         force_render=True, cfg=None, **kwargs):
    """reference __init__.py:14-55, argument for argument."""
    from isaacgymenvs.utils.rlgames_utils import get_rlgames_env_creator
    if cfg is None:
        if os.path.isdir(os.path.join(CFG_DIR, "task")):
            cfg_dict = compose(task)["task"]                                   # __init__.py:36-39
        else:   # no yaml tree on this machine: the defaults the yamls hold (isaacgym_amd.scene cites each line)
            from isaacgym_amd import scene
            cfg_dict = scene.default_task_cfg(scene.TASK_VARIANTS[task])
            cfg_dict["name"] = task
        cfg_dict["env"]["numEnvs"] = num_envs
    else:
        cfg_dict = omegaconf_to_dict(cfg["task"] if isinstance(cfg, dict) else cfg.task)   # __init__.py:42
    create_rlgpu_env = get_rlgames_env_creator(
        seed=seed, task_config=cfg_dict, task_name=cfg_dict["name"], sim_device=sim_device, rl_device=rl_device,
        graphics_device_id=graphics_device_id, headless=headless, multi_gpu=multi_gpu, virtual_screen_capture=virtual_screen_capture,
        force_render=force_render)
    return create_rlgpu_env()
