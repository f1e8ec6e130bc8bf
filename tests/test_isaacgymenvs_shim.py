"""N1: the `isaacgymenvs`-named shim.  The names the reference's train.py imports (train.py:80-99) resolve; the reference's task yamls,
read as plain YAML with their ${resolve_default:...} interpolations (isaacgym_amd.cfgyaml), build native configs; and — with -m gpu —
the body of train.py:122-167 runs against stub rl_games registries and steps every task name."""
import json
import os
import sys
import types

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from isaacgym_amd import cfgyaml, scene

REF_CFG = "/root/reference/cfg"
TASKS = ("HumanoidPingpongG1", "HumanoidPingpongTiltG1", "HumanoidPingpongTiltNoEarlyStopG1", "HumanoidPingpongTiltNESSparse27DOFG1")
# What a user of the reference has to pass on the command line as well, because the yamls lack keys their own task classes read
# (SURVEY.md §0.2): TT reads hitTableReward / nothitTablePenalty (TT:106-107; only the NoEarlyStop yaml has them, yaml:21,23); T3 reads
# alphaVelocityReward / powerCoefficient / penalty (T3:103-105) and is consistent only with the 10-body bodyStatesId of the Tilt yaml.
CLI_OVERRIDES = {
    "HumanoidPingpongTiltG1": {"hitTableReward": 2000.0, "nothitTablePenalty": -1000.0},
    "HumanoidPingpongG1": {"alphaVelocityReward": 50.0, "powerCoefficient": 0.0005, "penalty": -200.0, "bodyStatesId": [0, 31, 32, 33, 34, 35, 36, 37, 38, 39]},
}


def snapshot():
    return json.load(open(os.path.join(GOLDEN_DIR, "task_cfgs.json")))


def task_cfg(name, num_envs):
    cfg = snapshot()[name]["task"]
    cfg["env"]["numEnvs"] = num_envs
    cfg["env"].update(CLI_OVERRIDES.get(name, {}))
    return cfg


def test_every_name_train_py_imports_resolves():
    import isaacgym  # noqa: F401  (train.py:80)
    import isaacgymenvs
    from isaacgymenvs.learning import amp_continuous, amp_models, amp_network_builder, amp_players  # noqa: F401  (train.py:96-99)
    from isaacgymenvs.pbt.pbt import PbtAlgoObserver, initial_pbt_check  # noqa: F401  (train.py:81)
    from isaacgymenvs.tasks import isaacgym_task_map  # (train.py:85)
    from isaacgymenvs.utils.reformat import omegaconf_to_dict, print_dict  # noqa: F401  (train.py:87)
    from isaacgymenvs.utils.rlgames_utils import (ComplexObsRLGPUEnv, MultiObserver, RLGPUAlgoObserver, RLGPUEnv,  # noqa: F401
                                                  get_rlgames_env_creator, multi_gpu_get_rank)  # (train.py:82,91; __init__.py:27)
    from isaacgymenvs.utils.utils import set_np_formatting, set_seed  # noqa: F401  (train.py:88)
    from isaacgymenvs.utils.wandb_utils import WandbAlgoObserver  # noqa: F401  (train.py:92)
    assert callable(isaacgymenvs.make)
    assert {"HumanoidPingpongG1", "HumanoidPingpongTiltG1", "HumanoidPingpongTiltNoEarlyStopG1", "Humanoid12PingpongTiltG1"} <= set(isaacgym_task_map)
    assert multi_gpu_get_rank(False) == 0
    with pytest.raises(AttributeError, match="isaacgym_amd"):
        isaacgym.gymapi


def test_interpolation_resolver():
    root = {"num_envs": "", "pipeline": "gpu", "sim_device": "cuda:0", "checkpoint": "", "experiment": "",
            "task": {"name": "X", "env": {"numEnvs": "${resolve_default:4096,${...num_envs}}", "twice": "${.numEnvs}"},
                     "sim": {"use_gpu_pipeline": "${eq:${...pipeline},\"gpu\"}", "physx": {"use_gpu": "${contains:\"cuda\",${....sim_device}}"}}},
            "train": {"params": {"load_checkpoint": "${if:${...checkpoint},True,False}",
                                 "config": {"name": "${resolve_default:Humanoid,${....experiment}}", "full": "${.name}_run", "n": "${....task.env.numEnvs}"}}}}
    r = cfgyaml.resolve(root)
    assert r["task"]["env"] == {"numEnvs": 4096, "twice": 4096}
    assert r["task"]["sim"]["use_gpu_pipeline"] is True and r["task"]["sim"]["physx"]["use_gpu"] is True
    assert r["train"]["params"]["load_checkpoint"] is False
    assert r["train"]["params"]["config"] == {"name": "Humanoid", "full": "Humanoid_run", "n": 4096}
    root["num_envs"] = 64
    assert cfgyaml.resolve(root)["task"]["env"]["numEnvs"] == 64
    with pytest.raises(cfgyaml.InterpolationError):
        cfgyaml.resolve({"a": "${b}", "b": "${a}"})
    with pytest.raises(cfgyaml.InterpolationError):
        cfgyaml.resolve({"a": {"b": "${...nope}"}})


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="the reference checkout is not on this machine (the snapshot is used instead)")
def test_reference_yamls_compose_to_the_committed_snapshot():
    snap = snapshot()
    for name in TASKS:
        c = cfgyaml.compose(name, REF_CFG)
        assert c["task"] == snap[name]["task"], name
        assert c.get("train") == snap[name]["train"], name
    tt = cfgyaml.compose("HumanoidPingpongTiltG1", REF_CFG, {"num_envs": 16384, "alpha_velocity_reward": 75})
    assert tt["task"]["env"]["numEnvs"] == 16384 and tt["task"]["env"]["alphaVelocityReward"] == 75
    assert tt["train"]["params"]["config"]["num_actors"] == 16384 and tt["train"]["params"]["network"]["mlp"]["units"] == [2048, 1536, 1024, 1024, 512, 512]


@pytest.mark.parametrize("name", [t for t in TASKS if "27DOF" not in t])
def test_reference_yaml_builds_the_native_config(name):
    """cfg straight from the reference's yaml -> ppenv_config: yaml values arrive (episodeLength, reward constants, sim block)."""
    cfg = task_cfg(name, 32)
    if name in CLI_OVERRIDES:   # without them the reference's own constructor raises KeyError, and so does the native class
        bare = snapshot()[name]["task"]
        missing = set(CLI_OVERRIDES[name]) - set(bare["env"]) - {"bodyStatesId"}
        assert missing, "the yaml gained the keys: drop the override"
    defaults = scene.default_task_cfg(scene.TASK_VARIANTS[name])
    cfg.setdefault("scene", defaults["scene"])
    c = scene.build_config(scene.TASK_VARIANTS[name], cfg=cfg, num_envs=32)
    assert c.max_episode_length == cfg["env"]["episodeLength"]
    assert abs(c.dt - cfg["sim"]["dt"]) < 1e-9 and c.substeps == cfg["sim"]["substeps"]
    assert c.alpha_velocity_reward == cfg["env"]["alphaVelocityReward"] and c.clip_actions == cfg["env"]["clipActions"]


def test_set_seed_adds_the_rank():
    from isaacgymenvs.utils.utils import set_seed
    assert set_seed(7, rank=3) == 10 and set_seed(-1, torch_deterministic=True, rank=2) == 44


def test_make_with_multi_gpu_keeps_the_seed_and_offsets_the_env_ids(monkeypatch):
    """train.py:119 has already added the rank to cfg.seed; make() / the env creator must not add it again, and shards are told apart
    by their global env-id offset (isaacgym_amd.distributed)."""
    import isaacgym_amd.rlgames_utils as ru
    seen = {}

    class Fake:
        def __init__(self, cfg, rl, sim, *a):
            seen.update(cfg=cfg, rl=rl, sim=sim)
    monkeypatch.setitem(ru.isaacgym_task_map, "HumanoidPingpongTiltG1", Fake)
    monkeypatch.setenv("LOCAL_RANK", "2")
    monkeypatch.setenv("RANK", "5")
    ru.get_rlgames_env_creator(seed=11, task_config=task_cfg("HumanoidPingpongTiltG1", 100), task_name="HumanoidPingpongTiltG1", sim_device="cuda:0",
                               rl_device="cuda:0", multi_gpu=True)()
    assert seen["cfg"]["seed"] == 11 and seen["cfg"]["env_id_offset"] == 500 and seen["sim"] == seen["rl"] == "cuda:2"


# ------------------------------------------------------------------------------------------------------------------ GPU
class _AttrDict(dict):
    """cfg.task.env.numEnvs style access over the plain nested dict (what an OmegaConf DictConfig offers train.py)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _AttrDict(v) if isinstance(v, dict) else v


@pytest.mark.gpu
@pytest.mark.parametrize("name", TASKS)
def test_train_py_body_runs_against_stub_rl_games(name, monkeypatch):
    """train.py:122-167 — create_isaacgym_env, env_configurations.register('rlgpu', ...), vecenv.register('RLGPU', ... RLGPUEnv ...) — with
    stub rl_games registries, then what rl_games does with them: build the vec env through the registry and step it."""
    import torch
    configurations, vecenvs = {}, {}
    rl_games = types.ModuleType("rl_games")
    common = types.ModuleType("rl_games.common")
    env_configurations = types.ModuleType("rl_games.common.env_configurations")
    env_configurations.configurations = configurations
    env_configurations.register = lambda n, c: configurations.__setitem__(n, c)
    vecenv = types.ModuleType("rl_games.common.vecenv")
    vecenv.IVecEnv = object
    vecenv.register = lambda n, f: vecenvs.__setitem__(n, f)
    rl_games.common, common.env_configurations, common.vecenv = common, env_configurations, vecenv
    for k, m in {"rl_games": rl_games, "rl_games.common": common, "rl_games.common.env_configurations": env_configurations,
                 "rl_games.common.vecenv": vecenv}.items():
        monkeypatch.setitem(sys.modules, k, m)
    import isaacgym_amd.rlgames_utils as ru
    monkeypatch.setattr(ru, "env_configurations", env_configurations)

    import isaacgymenvs
    from isaacgymenvs.tasks import isaacgym_task_map
    from isaacgymenvs.utils.reformat import omegaconf_to_dict
    from isaacgymenvs.utils.rlgames_utils import ComplexObsRLGPUEnv, RLGPUEnv
    from isaacgymenvs.utils.utils import set_seed
    n = 192
    cfg = _AttrDict(dict(cfgyaml.ROOT_DEFAULTS, task_name=name, task=task_cfg(name, n), train=snapshot()[name]["train"], seed=5, graphics_device_id=-1))
    cfg_dict = omegaconf_to_dict(cfg)
    assert cfg_dict["task"]["env"]["numEnvs"] == n
    cfg["seed"] = set_seed(cfg.seed, torch_deterministic=cfg.torch_deterministic, rank=0)        # train.py:116

    def create_isaacgym_env(**kwargs):                                                           # train.py:122-145
        return isaacgymenvs.make(cfg.seed, cfg.task_name, cfg.task.env.numEnvs, cfg.sim_device, cfg.rl_device, cfg.graphics_device_id, cfg.headless,
                                 cfg.multi_gpu, cfg.capture_video, cfg.force_render, cfg, **kwargs)
    env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": lambda **kwargs: create_isaacgym_env(**kwargs)})   # train.py:147-150
    ige_env_cls = isaacgym_task_map[cfg.task_name]                                               # train.py:152
    dict_cls = ige_env_cls.dict_obs_cls if hasattr(ige_env_cls, "dict_obs_cls") and ige_env_cls.dict_obs_cls else False
    assert not dict_cls
    vecenv.register("RLGPU", lambda config_name, num_actors, **kwargs: RLGPUEnv(config_name, num_actors, **kwargs))   # train.py:167

    # rl_games' side: vecenv.create_vec_env(config['env_name'], num_actors) -> the registered factory
    ve = vecenvs[configurations["rlgpu"]["vecenv_type"]]("rlgpu", n)
    info = ve.get_env_info()
    task = ve.env
    assert task.num_envs == n and info["observation_space"].shape == (task.num_obs,) and info["action_space"].shape == (task.num_actions,)
    assert task.max_episode_length == cfg.task.env.episodeLength
    obs = ve.reset()["obs"]
    assert obs.shape == (n * task.num_agents, task.num_obs)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(5):
        a = torch.rand(n * task.num_agents, task.num_actions, device="cuda", generator=gen) * 2 - 1
        od, rew, done, extras = ve.step(a)
    assert torch.isfinite(od["obs"]).all() and torch.isfinite(rew).all() and done.dtype == torch.int64 and "time_outs" in extras
    assert int(task.progress_buf.max()) == 5
