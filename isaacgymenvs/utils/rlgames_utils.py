"""isaacgymenvs.utils.rlgames_utils (train.py:82,91): the env creator and the rl_games vec-env wrappers, from isaacgym_amd; the
observers are thin stand-ins (rl_games is not part of this repository: with it installed they derive from its AlgoObserver)."""
import os

from isaacgym_amd.rlgames_utils import RLGPUEnv, get_rlgames_env_creator  # noqa: F401

try:   # pragma: no cover - rl_games is absent offline
    from rl_games.common.algo_observer import AlgoObserver as _AlgoObserver
except Exception:   # noqa: BLE001
    class _AlgoObserver:
        def before_init(self, base_name, config, experiment_name): pass
        def after_init(self, algo): pass
        def process_infos(self, infos, done_indices): pass
        def after_steps(self): pass
        def after_clear_stats(self): pass
        def after_print_stats(self, frame, epoch_num, total_time): pass


def multi_gpu_get_rank(multi_gpu):
    """train.py:82: the rank of this process (torchrun sets LOCAL_RANK)."""
    return int(os.getenv("LOCAL_RANK", "0")) if multi_gpu else 0


class RLGPUAlgoObserver(_AlgoObserver):
    """Collects what the task exports in `extras` (isaacgym_amd.vec_task: time_outs, reward_mean, progress_mean — TT:767-768) for the
    learner's logs.  Device tensors are kept as they are; nothing here synchronises the env stream."""

    def __init__(self):
        super().__init__()
        self.algo = None
        self.last_extras = {}

    def after_init(self, algo):
        self.algo = algo

    def process_infos(self, infos, done_indices):
        if isinstance(infos, dict):
            self.last_extras = {k: v for k, v in infos.items() if k != "time_outs"}


class MultiObserver(_AlgoObserver):
    """Fans every callback out to a list of observers (train.py:201)."""

    def __init__(self, observers):
        super().__init__()
        self.observers = list(observers)

    def _call(self, name, *a, **k):
        for o in self.observers:
            getattr(o, name)(*a, **k)

    def before_init(self, *a, **k): self._call("before_init", *a, **k)
    def after_init(self, *a, **k): self._call("after_init", *a, **k)
    def process_infos(self, *a, **k): self._call("process_infos", *a, **k)
    def after_steps(self, *a, **k): self._call("after_steps", *a, **k)
    def after_clear_stats(self, *a, **k): self._call("after_clear_stats", *a, **k)
    def after_print_stats(self, *a, **k): self._call("after_print_stats", *a, **k)


class ComplexObsRLGPUEnv(RLGPUEnv):
    """Dict-observation wrapper (train.py:145).  None of the HumanoidPingpong task classes defines `dict_obs_cls`, so train.py never
    takes this branch for them; constructing it says so."""

    def __init__(self, config_name, num_actors, obs_spec, **kwargs):
        raise NotImplementedError("dict observations are not used by the HumanoidPingpong tasks (no dict_obs_cls; train.py:134-135)")
