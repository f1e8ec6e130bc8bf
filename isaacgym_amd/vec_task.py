"""The VecTask buffer surface the reference's training stack drives, over the native environment.

Mirrors the interface of `isaacgymenvs.tasks.base.vec_task.VecTask` as the reference task classes
use it (tasks/humanoid_pingpong_3_actor_tilt.py:60,118-123,1023-1037 — the base class itself is not
part of the reference): constructor signature, `num_envs / num_obs / num_actions / device`,
`obs_buf, rew_buf, reset_buf, progress_buf, randomize_buf, reset_buf_force, extras`,
`step(actions) -> (obs_dict, rew, reset, extras)` and `reset() -> obs_dict`, so that
`RLGPUEnv` / rl_games (reference train.py:147-167) can drive it unchanged.

Where the reference runs pre_physics_step -> gym.simulate -> post_physics_step as ~900 separate
torch dispatches plus PhysX, this class issues ONE kernel launch per step through the C ABI
(include/ppenv.h).  No host synchronisation happens inside step().
"""
import copy

import numpy as np
import torch

from . import scene
from .env import PPEnv


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency here); rl_games reads .shape/.low/.high."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(self.shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def _parse_device(sim_device):
    s = str(sim_device)
    if s.startswith("cpu"):
        raise ValueError("sim_device='cpu' is not supported: the native environment has no CPU pipeline "
                         "(the CPU restatement under oracle/ is test infrastructure, not a backend)")
    if ":" in s:
        return torch.device("cuda", int(s.split(":")[1]))
    return torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)


class VecTask:
    VARIANT = None   # 'T3' | 'TT' | 'TN', set by subclasses

    def __init__(self, config, rl_device, sim_device, graphics_device_id=-1, headless=True, virtual_screen_capture=False,
                 force_render=False):
        self.cfg = config
        env_cfg = config["env"]
        self.device = _parse_device(sim_device)
        self.device_id = self.device.index
        self.rl_device = torch.device(rl_device) if rl_device is not None else self.device
        self.graphics_device_id = graphics_device_id
        self.headless = True          # no viewer: rendering is out of scope
        self.viewer = None
        self.enable_viewer_sync = False
        self.force_render = force_render
        self.physics_engine = "ppenv"

        self.num_envs = int(env_cfg["numEnvs"])
        self.num_agents = int(getattr(self, "NUM_AGENTS", 1))   # rl_games multi-agent convention: batch rows = num_envs * num_agents
        self.num_observations = self.num_obs = int(env_cfg["numObservations"])
        self.num_states = int(env_cfg.get("numStates", 0))
        self.num_actions = self.num_acts = int(env_cfg["numActions"])
        self.control_freq_inv = int(env_cfg.get("controlFrequencyInv", 1))
        self.clip_obs = float(env_cfg.get("clipObservations", np.inf))
        self.clip_actions = float(env_cfg.get("clipActions", np.inf))
        self.obs_space = self.observation_space = Box(-np.inf, np.inf, (self.num_obs,))
        self.state_space = Box(-np.inf, np.inf, (self.num_states,))
        self.act_space = self.action_space = Box(-1.0, 1.0, (self.num_actions,))
        self.extras = {}
        self.obs_dict = {}
        self.control_steps = 0
        task_cfg = config.get("task", {}) or {}
        self.randomize = bool(task_cfg.get("randomize", False))                     # cfg/task/HumanoidPingpongTiltG1.yaml:101
        self.randomization_params = task_cfg.get("randomization_params", {}) or {}  # yaml:102-169
        self.first_randomization, self.last_step, self.last_rand_step = True, -1, -1
        self.stats_every = int(config.get("stats_every", 40))                       # TT:763: the reference prints the means every 40 steps

        self.sim = self.create_sim()       # the reference's VecTask.__init__ calls back create_sim (TT:325)
        self.allocate_buffers()

    # -- hooks the task class fills in
    def create_sim(self):
        raise NotImplementedError

    def allocate_buffers(self):
        """obs/rew/reset/progress alias the native arena; the rest are plain torch buffers (VecTask.allocate_buffers)."""
        e = self.env
        self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf = e.obs_buf, e.rew_buf, e.reset_buf, e.progress_buf
        rows = self.num_envs * self.num_agents
        self.states_buf = torch.zeros((rows, self.num_states), device=self.device, dtype=torch.float)
        self.timeout_buf = torch.zeros(rows, device=self.device, dtype=torch.long)
        self.randomize_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        # Fork-specific buffer of the reference's base class.  In the reference it is write-only: post_physics_step zeroes it
        # every step (TT:1037) and the one statement that would read it is commented out (TT:761), so it never influences a
        # reset.  Kept for attribute compatibility, all zeros, and — like the reference — never consulted.
        self.reset_buf_force = torch.zeros(rows, device=self.device, dtype=torch.long)
        self._obs_clipped = None if not np.isfinite(self.clip_obs) else torch.empty_like(self.obs_buf)

    # -- the surface rl_games drives
    def step(self, actions):
        """One control step: K1..K8 in a single kernel launch (TT:1002-1052).  controlFrequencyInv = k (upstream: pre_physics_step
        once, k simulate calls, post_physics_step once) is folded into the native config at create time — k * substeps physics
        substeps of the same length ahead of one reward / reset / observation pass — so it is still one launch."""
        if self.randomize:                   # upstream applies them from _reset_idx (TT:849-850); here once per step, gated by `frequency`
            self.apply_randomizations(self.randomization_params)
        self.env.step(actions)              # action clamp (clipActions) happens inside the kernel
        self.control_steps += 1
        self.last_step = self.control_steps
        if self.stats_every > 0 and self.control_steps % self.stats_every == 0:
            self._export_means()
        # upstream: timeout_buf = (progress_buf >= max_episode_length - 1) & (reset_buf != 0), evaluated after
        # post_physics_step.  These tasks zero progress_buf inside post_physics_step on every reset (TT:902), so
        # upstream's time_outs is identically False for them; keep the (constant) tensor instead of two launches.
        self.extras["time_outs"] = self.timeout_buf if self.rl_device == self.device else self.timeout_buf.to(self.rl_device)
        return self._obs_dict(), self._to_rl(self.rew_buf), self._to_rl(self.reset_buf), self.extras

    def _export_means(self):
        """extras['reward_mean'], extras['progress_mean'] (TT:763-768: the reference computes them every 40 steps and — its two
        `self.extras[...] =` lines are commented out — prints them).  One reduction launch (ppenv_reduce_stats), no host
        synchronisation: the values are 0-dim tensors on the sim device."""
        s = self.env.reduce_stats()
        self.extras["reward_mean"] = (s[0] / s[3]).float()
        self.extras["progress_mean"] = (s[1] / s[3]).float()

    def apply_randomizations(self, dr_params):
        """Domain randomisation (TT:849-850 -> upstream VecTask.apply_randomizations; parameters cfg/task/HumanoidPingpongTiltG1.yaml:
        102-169).  Every `frequency` control steps (and before the first) new values are drawn — per env, with the yaml's
        distribution / operation / schedule — for: observation and action noise, gravity, the humanoid's link masses, shape friction
        and restitution, dof stiffness and damping.  The draws land in per-env device tables the step kernel reads
        (ppenv_set_randomization); gravity is one value for the simulation.  `color`, `lower` / `upper` and `setup_only`'s
        distinction have no counterpart here and are ignored (the limits enter the compiled model)."""
        if not dr_params:
            return
        freq = int(dr_params.get("frequency", 1))
        if not self.first_randomization and (self.last_step - self.last_rand_step) < freq:
            return
        self.first_randomization, self.last_rand_step = False, self.last_step
        n, dev = self.num_envs, self.device
        gen = getattr(self, "_dr_gen", None)
        if gen is None:
            gen = self._dr_gen = torch.Generator(device=dev).manual_seed(int(self.cfg.get("seed", 0)) + 7919)

        def sched(p):
            if p.get("schedule") == "linear":
                return min(max(self.last_step, 0), int(p["schedule_steps"])) / float(p["schedule_steps"])
            if p.get("schedule") == "constant":
                return 1.0 if self.last_step > int(p["schedule_steps"]) else 0.0
            return 1.0

        def sample(p, shape):
            """Upstream semantics: `range` = (lo, hi) for uniform / (mu, sigma) for gaussian; a scaling is blended towards 1 and an
            additive term towards 0 by the schedule."""
            a, b = float(p["range"][0]), float(p["range"][1])
            if p.get("distribution", "uniform") == "gaussian":
                v = torch.randn(shape, device=dev, generator=gen) * b + a
            else:
                v = torch.rand(shape, device=dev, generator=gen) * (b - a) + a
            s = sched(p)
            return v * s + (1.0 - s) if p.get("operation") == "scaling" else v * s

        kw = {}
        act, obs = dr_params.get("actions"), dr_params.get("observations")
        if act:
            kw["action_noise_sigma"] = float(act["range"][1]) * sched(act)
        if obs:
            kw["observation_noise_sigma"] = float(obs["range"][1]) * sched(obs)
        g = (dr_params.get("sim_params") or {}).get("gravity")
        if g:      # one value per simulation; the 27-dof task re-uploads its scene constants (ppenv_ta_sim_set_gravity)
            base = self.native_config.gravity_z if hasattr(self, "native_config") else scene.TA_GRAVITY_Z
            dz = float(sample(g, (1,)).item()) if g.get("operation") == "additive" else 0.0
            self.env.set_gravity(min(base + dz, 0.0) if g.get("operation") == "additive" else base * float(sample(g, (1,)).item()))
        hum = ((dr_params.get("actor_params") or {}).get("humanoid") or {})
        mass = (hum.get("rigid_body_properties") or {}).get("mass")
        if mass:
            kw["link_mass_scale"] = sample(mass, (getattr(self, "DR_MASS_ROWS", scene.NUM_DOF), n))
        shape = hum.get("rigid_shape_properties") or {}
        if shape.get("friction"):
            kw["friction_scale"] = sample(shape["friction"], (n,))
        if shape.get("restitution"):
            kw["restitution_scale"] = sample(shape["restitution"], (n,))
        dof = hum.get("dof_properties") or {}
        if dof.get("stiffness"):
            kw["dof_stiffness_scale"] = sample(dof["stiffness"], (getattr(self, "DR_DOF_ROWS", scene.NUM_DOF), n))
        if dof.get("damping"):
            kw["dof_damping_scale"] = sample(dof["damping"], (getattr(self, "DR_DOF_ROWS", scene.NUM_DOF), n))
        self.env.set_randomization(**kw)
        self.randomize_buf.zero_()

    def reset(self):
        """Observation dictionary of the current state (upstream VecTask.reset does not step the simulator)."""
        return self._obs_dict()

    def reset_idx(self, env_ids=None):
        """reset_idx(env_ids) -> _reset_idx (TT:809-812, 847-906): the listed envs go back to their initial state with a fresh
        serve, progress 0 and the initial flags; every other env keeps its trajectory.  `env_ids` as the reference passes them:
        an int64 tensor of env indices (`reset_buf.nonzero()`, TT:1034).  None resets every env (VecTask.reset_idx() of a fresh
        task).  For the two-agent task an id is an ENV index (both agent rows of that env reset together, T4:853-912)."""
        if env_ids is None:
            self.env.reset_all()
        else:
            self.env.reset_idx(env_ids)

    def reset_done(self):
        return self._obs_dict(), torch.nonzero(self.reset_buf, as_tuple=False).flatten()

    def zero_actions(self):
        return torch.zeros((self.num_envs * self.num_agents, self.num_actions), dtype=torch.float32, device=self.rl_device)

    def get_number_of_agents(self):
        return self.num_agents

    def get_env_info(self):
        return {"action_space": self.action_space, "observation_space": self.observation_space, "state_space": self.state_space,
                "agents": self.num_agents}

    def _to_rl(self, t):
        return t if self.rl_device == self.device else t.to(self.rl_device)

    def _obs_dict(self):
        obs = self.obs_buf if self._obs_clipped is None else torch.clamp(self.obs_buf, -self.clip_obs, self.clip_obs, out=self._obs_clipped)
        self.obs_dict["obs"] = self._to_rl(obs)
        if self.num_states > 0:
            self.obs_dict["states"] = self._to_rl(self.states_buf)
        return self.obs_dict


class _HumanoidPingpongBase(VecTask):
    """Shared body of the three 7-DoF task classes; cfg keys are the reference yamls' (cfg/task/*.yaml)."""

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=-1, headless=True, virtual_screen_capture=False,
                 force_render=False):
        cfg = copy.deepcopy(cfg) if cfg is not None else scene.default_task_cfg(self.VARIANT)
        defaults = scene.default_task_cfg(self.VARIANT)
        cfg.setdefault("sim", defaults["sim"])
        cfg.setdefault("scene", defaults["scene"])
        for k, v in defaults["env"].items():
            if k not in cfg["env"] and k in ("bodyStatesId", "plane", "clipActions"):
                cfg["env"][k] = v
        env = cfg["env"]
        self.max_episode_length = env["episodeLength"]            # TT:84
        env["numObservations"] = scene.NUM_OBS                    # TT:98  30+30+7+7+3+3
        env["numActions"] = scene.NUM_DOF                         # TT:101
        # the reference reads these unconditionally (TT:103-107); a missing key is the same KeyError
        self.alpha = env["alphaVelocityReward"]
        self.power_coefficient = env["powerCoefficient"]
        self.penalty = env["penalty"]
        if self.VARIANT != "T3":   # TT:106-107, T4:106-107
            self.hit_table_reward = env["hitTableReward"]
            self.not_hit_table_penalty = env["nothitTablePenalty"]
        else:
            env.setdefault("hitTableReward", 0.0)
            env.setdefault("nothitTablePenalty", 0.0)
        self.initial_speed_range = tuple(cfg["scene"]["serve_speed"])
        self.tilt_angle_range = tuple(cfg["scene"]["serve_tilt"])
        self.tilt_z_angle_range = tuple(cfg["scene"]["serve_tilt_z"])
        A = int(getattr(self, "NUM_AGENTS", 1))
        self.actors_per_env, self.dofs_per_env = A + 2, A * scene.NUM_DOF                    # T4:125-126
        self.num_humanoid_bodies = 40
        self.rigid_bodies_per_env = A * self.num_humanoid_bodies + 2                          # T4:127
        self._seed = int(cfg.get("seed", 0))
        self._env_id_offset = int(cfg.get("env_id_offset", 0))
        self.dt = cfg["sim"]["dt"]
        super().__init__(config=cfg, rl_device=rl_device, sim_device=sim_device, graphics_device_id=graphics_device_id,
                         headless=headless, virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        off, scale = scene.pd_action_offset_scale(self.native_config)
        self._pd_action_offset = torch.tensor(off, device=self.device)   # TT:664-668
        self._pd_action_scale = torch.tensor(scale, device=self.device)
        self.body_states_id = torch.tensor(env["bodyStatesId"], dtype=torch.long, device=self.device)
        self.num_steps = 0

    def create_sim(self):
        """TT:325-344: build the scene.  Here: scene constants -> ppenv_config -> native handle."""
        self.up_axis_idx = 2
        table, ball = scene.asset_geometry(self.cfg["scene"])         # pingpong_table.urdf / small_ball.urdf when the cfg names them (TT:496,502)
        self.native_config = scene.build_config(self.VARIANT, cfg=self.cfg, num_envs=self.num_envs, seed=self._seed,
                                                device_id=self.device_id, env_id_offset=self._env_id_offset, table=table, ball=ball)
        k = self.control_freq_inv
        if k < 1:
            raise ValueError("controlFrequencyInv must be >= 1")
        if k > 1:   # k simulate calls of dt each = k * substeps substeps of dt / substeps, then ONE post_physics_step
            self.native_config.dt = self.native_config.dt * k
            self.native_config.substeps = self.native_config.substeps * k
        self.env = PPEnv(self.native_config, device=self.device)
        return self.env

    # gym.refresh_* equivalents (TT:801-807) in the reference's tensor layouts, materialised on demand
    def refresh_sim_tensors(self):
        A = self.num_agents
        self.root_states = self.env.refresh_root_states()
        self.vec_root_states = self.root_states
        self.humanoid1_root_states = self.root_states[:, 0, :]
        self.table_root_states = self.root_states[:, A, :]         # actor order: humanoid(s), table, ball (TT:179-183 / T4:181-185)
        self.ball2_root_states = self.root_states[:, A + 1, :]
        self.vec_dof_states = self.env.refresh_dof_states()
        self.dof_pos, self.dof_vel = self.vec_dof_states[..., 0], self.vec_dof_states[..., 1]
        self.body_states = self.vec_rb_states = self.env.refresh_rigid_body_states()
        self.humanoid1_paddle_rb_states = self.body_states[:, 39, :]
        if A == 2:                                                  # T4:169-172,181-185
            self.humanoid2_root_states = self.root_states[:, 1, :]
            self.humanoid1_rb_states, self.humanoid2_rb_states = self.body_states[:, 0:40, :], self.body_states[:, 40:80, :]
            self.humanoid2_paddle_rb_states = self.body_states[:, 79, :]
        self.dof_force_tensor = self.env.refresh_dof_force()

    def step(self, actions):
        out = super().step(actions)
        self.num_steps += 1
        return out

    # sticky flags as the reference's bool tensors (TT:241-243, TN:244-248), decoded on demand
    def _flag(self, bit):
        return (self.env.flags & bit) != 0

    @property
    def reward_calculated(self):
        return self._flag(scene.FLAG_REWARD_CALC)

    @property
    def condition_calculated(self):
        return self._flag(scene.FLAG_COND_CALC)

    paddle_condition_calculated = condition_calculated

    @property
    def no_bounce_before_half_mask(self):
        return self._flag(scene.FLAG_NO_BOUNCE)

    @property
    def missed_ball_calculated(self):
        return self._flag(scene.FLAG_MISSED_CALC)


class HumanoidPingpong(_HumanoidPingpongBase):
    """HumanoidPingpongG1 — tasks/humanoid_interos_edit_pingpong_only_3_actor.py (T3)."""
    VARIANT = "T3"


class HumanoidPingpongTilt(_HumanoidPingpongBase):
    """HumanoidPingpongTiltG1 — tasks/humanoid_pingpong_3_actor_tilt.py:58 (TT)."""
    VARIANT = "TT"


class HumanoidPingpongTiltNoEarlyStop(_HumanoidPingpongBase):
    """HumanoidPingpongTiltNoEarlyStopG1 — tasks/humanoid_pingpong_3_actor_tilt_no_earlystop.py (TN)."""
    VARIANT = "TN"


class Humanoid12PingpongTilt(_HumanoidPingpongBase):
    """Humanoid12PingpongTiltG1 — tasks/humanoid_pingpong_4_actor_tilt.py:58 (T4): two humanoids, one ball.

    The reference class is unfinished: it declares 14 actions and one 80-wide observation row per env (T4:98-101) but
    builds 7-wide PD tables (T4:669-674), calls a reward function that does not exist (T4:743) and lets the second
    humanoid's observations overwrite the first's (T4:786-803, "TODO").  This class is the build's completion of it in the
    rl_games multi-agent convention: `num_agents = 2`, agent a of env e owns row 2e + a of `obs_buf [2N, 80]`,
    `rew_buf / reset_buf / progress_buf [2N]` and of the `[2N, 7]` action tensor; both agents share the ball, the
    progress counter and the reset decision; sticky flags are per side (`flags [2, N]`)."""
    VARIANT = "T4"
    NUM_AGENTS = 2

    def _flag(self, bit):   # [2, N]: row a = side a
        return (self.env.flags & bit) != 0


class HumanoidPingpongTiltNESSparse27DOF(VecTask):
    """tasks/humanoid_pingpong_3_actor_all_dof.py:65 (TA; the reference does not register it, its yaml is
    cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml): the free-floating 27-dof humanoid.  313 observations (TA:106),
    27 actions (TA:109); `step` = ppenv_ta_simulate + ppenv_ta_post_physics_step on the Isaac-Gym-layout tensors the
    reference class wraps (TA:161-251), which are attributes here under the reference's names."""
    VARIANT = "TA"

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=-1, headless=True, virtual_screen_capture=False, force_render=False):
        cfg = copy.deepcopy(cfg) if cfg is not None else scene.default_task_cfg("TA")
        env = cfg["env"]
        env["numObservations"] = scene.TA_NUM_OBS      # TA:106
        env["numActions"] = scene.TA_NUM_DOF           # TA:109
        self.max_episode_length = env["episodeLength"]
        self._seed = int(cfg.get("seed", 0))
        self._env_id_offset = int(cfg.get("env_id_offset", 0))
        super().__init__(config=cfg, rl_device=rl_device, sim_device=sim_device, graphics_device_id=graphics_device_id, headless=headless,
                         virtual_screen_capture=virtual_screen_capture, force_render=force_render)
        self.actors_per_env, self.dofs_per_env, self.rigid_bodies_per_env = 3, scene.TA_NUM_DOF, scene.NUM_BODIES     # TA:156-158

    def create_sim(self):
        from .tensor_api import TAEnv
        keys = ("episodeLength", "alphaVelocityReward", "powerCoefficient", "hitTableReward", "nothitTablePenalty", "crossNetRewardFloat",
                "diePenaltyFloat", "hitPaddleReward", "missPaddlePenaltyCoefficient")
        env = {k: self.cfg["env"][k] for k in keys if k in self.cfg["env"]}
        if self.control_freq_inv != 1:
            raise NotImplementedError("controlFrequencyInv != 1 is not wired for the 27-dof task (its yaml has none; "
                                      "cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml)")
        table, ball = scene.asset_geometry(self.cfg.get("scene", {}))     # TA:551,557
        scene_cfg = scene.build_ta_scene(self.num_envs, device_id=self.device_id, table=table, ball=ball) if (table or ball) else None
        with torch.cuda.device(self.device):
            self.env = TAEnv(self.num_envs, device=self.device, seed=self._seed, env_id_offset=self._env_id_offset, env=env, scene_cfg=scene_cfg)
        e = self.env
        self.root_states = self.vec_root_states = e.root_states
        self.vec_dof_states = e.dof_states
        self.dof_pos, self.dof_vel = e.dof_states[..., 0], e.dof_states[..., 1]
        self.initial_body_states = e.initial_rb_states
        self.dof_force_tensor = e.dof_force_tensor
        return e

    @property
    def body_states(self):
        """rigid_body_states [N,42,13] (TA:195-200); refreshed on demand when the step does not materialise it (TAEnv.rb_states)."""
        return self.env.rb_states

    vec_rb_states = body_states

    # the 27-dof tree: 27 dofs, 28 links (link 0 = the pelvis) — the table shapes of ppenv_ta_randomization
    DR_DOF_ROWS, DR_MASS_ROWS = scene.TA_NUM_DOF, scene.TA_NUM_LINKS

    def step(self, actions):
        if self.randomize:                  # as the 7-dof tasks: once per step, gated by `frequency` (upstream: from _reset_idx, TA's reset path)
            self.apply_randomizations(self.randomization_params)
        self.env.step(actions)              # the clipActions clamp happens inside the kernel
        self.control_steps += 1
        self.last_step = self.control_steps
        if self.stats_every > 0 and self.control_steps % self.stats_every == 0:   # TA:860-866 prints the same two means every 40 steps
            self.extras["reward_mean"] = self.rew_buf.mean()
            self.extras["progress_mean"] = self.progress_buf.float().mean()
        self.extras["time_outs"] = self.timeout_buf if self.rl_device == self.device else self.timeout_buf.to(self.rl_device)
        return self._obs_dict(), self._to_rl(self.rew_buf), self._to_rl(self.reset_buf), self.extras

    def reset_idx(self, env_ids=None):
        """_reset_idx (TA:965-1028) for the listed envs (None: all): root states, dof states, ball y / z and serve of the env's
        next episode, progress 0, the four sticky flags cleared.  Like the reference's, it leaves obs_buf to the next step."""
        self.env.reset_idx(env_ids)
