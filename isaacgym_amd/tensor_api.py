"""Tensor-API mode for the 27-DoF variant (tasks/humanoid_pingpong_3_actor_all_dof.py, "TA").

`TAState.post_physics_step` is the drop-in for the reference's post_physics_step (TA:1145-1192) on the
simulator tensors the task already wraps: reward (TA:1440-1690), masked reset (TA:965-1028) and the
313-wide observation (TA:867-904) in one launch (+ one tiny launch for the global count-flag clear).
`TASim` is the rigid-body step of this variant (ppenv_ta_simulate: the free-floating 27-DoF humanoid, its ground contacts
and the ball) on the same tensors; `TAEnv` chains the two into the task's VecTask step.
"""
import ctypes as C

import torch

from . import _lib, scene


class TAState:
    """Per-env buffers the reference keeps across steps (progress, sticky/count flags) plus the outputs."""

    def __init__(self, params, device="cuda:0", library=None):
        self.L = library if library is not None else _lib.lib()
        self.params = params
        self.device = torch.device(device)
        n = self.num_envs = params.num_envs
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        self.obs_buf = z((n, scene.TA_NUM_OBS), torch.float32)
        self.rew_buf = z((n,), torch.float32)
        self.reset_buf = z((n,), torch.int64)
        self.progress_buf = z((n,), torch.int64)
        self.flags = z((n,), torch.int32)
        self.episode = z((n,), torch.int32)
        self._any_reset = z((1,), torch.int32)

    def _ck(self, rc):
        _lib.check(rc, self.L)          # the message is the thread-local one of THIS library instance

    def post_physics_step(self, rb_states, initial_rb_states, root_states, dof_states, dof_force, pre_ball_vx, reset_override=None):
        n = self.num_envs
        irb_n = 1 if self.params.initial_rb_shared else n
        for t, numel in ((rb_states, n * 42 * 13), (initial_rb_states, irb_n * 42 * 13), (root_states, n * 39), (dof_states, n * 54),
                         (dof_force, n * 27), (pre_ball_vx, n)):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and t.numel() == numel
        ov = None
        if reset_override is not None:
            ov = reset_override.to(self.device, torch.float32).reshape(n, 5).contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # (the library launches on the device that owns obs_buf, whatever the caller's current device is)
        self._ck(self.L.ppenv_ta_post_physics_step(
            C.byref(self.params), rb_states.data_ptr(), initial_rb_states.data_ptr(), root_states.data_ptr(), dof_states.data_ptr(),
            dof_force.data_ptr(), pre_ball_vx.data_ptr(), ov.data_ptr() if ov is not None else None, self.flags.data_ptr(),
            self.episode.data_ptr(), self.progress_buf.data_ptr(), self.obs_buf.data_ptr(), self.rew_buf.data_ptr(),
            self.reset_buf.data_ptr(), self._any_reset.data_ptr(), stream))
        if ov is not None:
            torch.cuda.current_stream(self.device).synchronize()   # keep `ov` alive until the kernel has read it


class TASim:
    """pre_physics_step + gym.simulate + refresh for the 27-DoF task (TA:1124-1143, 1150) on Isaac-Gym-layout tensors."""

    def __init__(self, num_envs, device="cuda:0", scene_cfg=None, model=None, library=None):
        """model: another 28-link tree (isaacgym_amd.urdf.ta_model of an asset).  library: a libppenv built with THAT tree compiled into
        the chain-wave kernel (_lib.load(_lib.build_for_ta_model(...))); the default library steps such a model on its table-driven kernels."""
        self.L = library if library is not None else _lib.lib()
        self.device = torch.device(device)
        self.num_envs = int(num_envs)
        self.scene = scene_cfg if scene_cfg is not None else scene.build_ta_scene(self.num_envs, device_id=self.device.index or 0)
        self.model = model if model is not None else scene.build_ta_model()
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._ck(self.L.ppenv_ta_sim_create(C.byref(self.scene), C.byref(self.model), self._stream(), C.byref(self.h)))
            torch.cuda.current_stream(self.device).synchronize()

    def _ck(self, rc):
        _lib.check(rc, self.L)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.L.ppenv_ta_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, t, numel):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and t.numel() == numel

    def simulate(self, actions, root_states, dof_states, rb_states, dof_force, pre_ball_vx):
        n = self.num_envs
        for t, k in ((actions, n * 27), (root_states, n * 39), (dof_states, n * 54), (rb_states, n * 42 * 13), (dof_force, n * 27), (pre_ball_vx, n)):
            self._check(t, k)
        self._ck(self.L.ppenv_ta_simulate(self.h, n, actions.data_ptr(), root_states.data_ptr(), dof_states.data_ptr(), rb_states.data_ptr(),
                                            dof_force.data_ptr(), pre_ball_vx.data_ptr(), self._stream()))

    @property
    def kernel(self):
        """Which kernel ppenv_ta_step launches: 'chain' (one lane per env, one wave per limb), 'quad' or 'lane'."""
        return {2: "chain", 1: "quad", 0: "lane"}[int(self.L.ppenv_ta_sim_kernel(self.h))]

    def set_gravity(self, gravity_z):
        """sim_params.gravity (ppenv_ta_sim_set_gravity): every later step of this simulation runs under gravity_z (<= 0)."""
        self._ck(self.L.ppenv_ta_sim_set_gravity(self.h, float(gravity_z), self._stream()))
        self.scene.gravity_z = float(gravity_z)

    @property
    def kernel_name(self):
        """... by the name rocprofv3's kernel trace shows (ppenv_ta_sim_kernel_name; 'ta_chain_kernel<true>' while a randomisation is set)."""
        return self.L.ppenv_ta_sim_kernel_name(self.h).decode()

    @property
    def status(self):
        return int(self.L.ppenv_ta_sim_status(self.h))

    def step(self, state, actions, initial_rb_states, root_states, dof_states, rb_states, dof_force, pre_ball_vx, reset_override=None,
             obs=None, rew=None, reset=None):
        """ppenv_ta_step: simulate + post_physics_step (on `state`: a TAState) in one launch.  rb_states may be None with the
        chain-wave kernel: rigid_body_states [N,42,13] is then not materialised.  obs / rew / reset: where this step's observation
        rows [N,313], rewards [N] and reset flags [N] (int64) go instead of the state's own buffers (a rollout collector's slices)."""
        n = self.num_envs
        obs = state.obs_buf if obs is None else obs
        rew = state.rew_buf if rew is None else rew
        reset = state.reset_buf if reset is None else reset
        self._check(obs, n * scene.TA_NUM_OBS)
        self._check(rew, n)
        assert reset.dtype == torch.int64 and reset.is_contiguous() and reset.device == self.device and reset.numel() == n
        irb_n = 1 if state.params.initial_rb_shared else n
        for t, k in ((actions, n * 27), (initial_rb_states, irb_n * 42 * 13), (root_states, n * 39), (dof_states, n * 54),
                     (dof_force, n * 27), (pre_ball_vx, n)) + (((rb_states, n * 42 * 13),) if rb_states is not None else ()):
            self._check(t, k)
        ov = None
        if reset_override is not None:
            ov = reset_override.to(self.device, torch.float32).reshape(n, 5).contiguous()
        self._ck(self.L.ppenv_ta_step(
            self.h, C.byref(state.params), actions.data_ptr(), initial_rb_states.data_ptr(), root_states.data_ptr(), dof_states.data_ptr(),
            rb_states.data_ptr() if rb_states is not None else None, dof_force.data_ptr(), pre_ball_vx.data_ptr(), ov.data_ptr() if ov is not None else None, state.flags.data_ptr(),
            state.episode.data_ptr(), state.progress_buf.data_ptr(), obs.data_ptr(), rew.data_ptr(), reset.data_ptr(),
            state._any_reset.data_ptr(), self._stream()))
        if ov is not None:
            torch.cuda.current_stream(self.device).synchronize()

    def set_policy_input(self, out=None, mean=None, inv_std=None, clip=5.0):
        """ppenv_ta_sim_set_policy_input: from the next `step` on the chain-wave kernel also writes out [N, ld] fp16 =
        clamp((obs - mean) * inv_std, +-clip), zero beyond column 312 (what policy.prepare_input makes of obs_buf).  out None: off.
        The tensors are read / written by every later step: the caller keeps them alive."""
        if out is None:
            self._ck(self.L.ppenv_ta_sim_set_policy_input(self.h, None, None, 0.0, None, 0))
            self._pin = None
            return
        assert out.dtype == torch.float16 and out.shape[0] == self.num_envs and out.stride(1) == 1 and mean.dtype == inv_std.dtype == torch.float32
        self._pin = (out, mean, inv_std)
        self._ck(self.L.ppenv_ta_sim_set_policy_input(self.h, mean.data_ptr(), inv_std.data_ptr(), float(clip), out.data_ptr(), out.stride(0)))

    def set_randomization(self, dof_stiffness_scale=None, dof_damping_scale=None, link_mass_scale=None, restitution_scale=None, friction_scale=None,
                          action_noise_sigma=0.0, observation_noise_sigma=0.0):
        """ppenv_ta_sim_set_randomization: per-env tables as float32 device tensors — drive stiffness / damping scales [27, N], link mass scales
        [28, N] (link 0 = pelvis), restitution / friction scales [N]; None = not randomised — and the two noise amplitudes.  Read by every later
        `step` (kept alive here); chain-wave kernel only."""
        n = self.num_envs

        def tab(t, rows):
            if t is None:
                return None
            t = torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous()
            assert tuple(t.shape) == ((rows, n) if rows else (n,)), tuple(t.shape)
            return t
        self._dr = [tab(dof_stiffness_scale, 27), tab(dof_damping_scale, 27), tab(link_mass_scale, 28), tab(restitution_scale, 0), tab(friction_scale, 0)]
        r = scene.Randomization()      # ppenv_ta_randomization has the fields of ppenv_randomization
        (r.dof_stiffness_scale, r.dof_damping_scale, r.link_mass_scale, r.restitution_scale, r.friction_scale) = [t.data_ptr() if t is not None else None for t in self._dr]
        r.action_noise_sigma, r.observation_noise_sigma = float(action_noise_sigma), float(observation_noise_sigma)
        self._ck(self.L.ppenv_ta_sim_set_randomization(self.h, C.byref(r)))

    def clear_randomization(self):
        self._ck(self.L.ppenv_ta_sim_set_randomization(self.h, None))
        self._dr = None

    def pd_targets(self, actions):
        """pre_physics_step's PD targets (TA:1131) for actions [N,27]."""
        a = actions.to(device=self.device, dtype=torch.float32).reshape(self.num_envs, 27).contiguous()
        out = torch.empty_like(a)
        self._ck(self.L.ppenv_ta_pd_targets(self.h, self.num_envs, a.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def serve_from_draws(self, draws):
        """TA's generate_random_speed_for_ball (TA:346-377) on [M,3] draws (speed, tilt deg, tilt_z deg)."""
        d = torch.as_tensor(draws, dtype=torch.float32).to(self.device).reshape(-1, 3).contiguous()
        out = torch.empty_like(d)
        self._ck(self.L.ppenv_ta_serve_from_draws(self.h, d.data_ptr(), d.shape[0], out.data_ptr(), self._stream()))
        return out

    def forward_kinematics(self, root_states, dof_states, rb_states):
        n = self.num_envs
        for t, k in ((root_states, n * 39), (dof_states, n * 54), (rb_states, n * 42 * 13)):
            self._check(t, k)
        self._ck(self.L.ppenv_ta_forward_kinematics(self.h, n, root_states.data_ptr(), dof_states.data_ptr(), rb_states.data_ptr(), self._stream()))


class TAEnv:
    """HumanoidPingpongTiltNESSparse27DOF (tasks/humanoid_pingpong_3_actor_all_dof.py:65) as a native task: the tensors the
    reference class wraps (TA:161-251) live here, `step` = pre_physics_step + simulate + post_physics_step in one launch
    (ppenv_ta_step) plus the tiny count-flag clear; `fused=False` keeps the two launches (ppenv_ta_simulate +
    ppenv_ta_post_physics_step).  Surface: obs_buf [N,313], rew_buf, reset_buf, progress_buf, 27 actions."""

    def __init__(self, num_envs, device="cuda:0", seed=0, env_id_offset=0, env=None, fused=True, materialize_rb=None, share_initial_rb=True,
                 scene_cfg=None, model=None, library=None):
        """materialize_rb: write rigid_body_states [N,42,13] in every step (the reference's refresh_rigid_body_state_tensor, pre-reset
        body states).  Default: only where the kernel needs the tensor itself (the two-launch path and the table-driven kernels);
        the chain-wave kernel keeps the body states in registers and `rb_states` is then produced on demand by forward kinematics
        of the CURRENT (post-reset) state."""
        self.fused = bool(fused)
        self.device = torch.device(device)
        n = self.num_envs = int(num_envs)
        self.num_obs, self.num_actions, self.num_agents = scene.TA_NUM_OBS, scene.TA_NUM_DOF, 1
        self.params = scene.build_ta_params(n, env=env, seed=seed, env_id_offset=env_id_offset)
        self.sim = TASim(n, device=self.device, scene_cfg=scene_cfg, model=model, library=library)
        self.state = TAState(self.params, device=self.device, library=library)
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
        self.root_states, self.dof_states = z(n, 3, 13), z(n, 27, 2)          # TA:187-193, 237-240
        self._rb_states, self.dof_force_tensor, self.pre_ball_vx = z(n, 42, 13), z(n, 27), z(n)
        self.materialize_rb = bool(materialize_rb) if materialize_rb is not None else not (self.fused and self.sim.kernel == "chain")
        init = torch.tensor([[self.params.init_root[a][k] for k in range(7)] for a in range(3)], dtype=torch.float32, device=self.device)
        self.root_states[:, :, 0:7] = init
        # creation = episode 0: the serve and ball position every env starts with come from the same keyed draws a reset uses
        ov = scene.ta_reset_draws(self.params, torch.arange(n), torch.zeros(n, dtype=torch.int64))
        self.root_states[:, 2, 1:3] = ov[:, 0:2].to(self.device)
        self.root_states[:, 2, 7:10] = ov[:, 2:5].to(self.device)
        self.sim.forward_kinematics(self.root_states, self.dof_states, self._rb_states)
        # TA:1152 initial_body_states.  Every env is created in the same pose (TA:578-579), so the rows the task reads — the 23 balance
        # bodies, all on the humanoid — are the same in every env: one shared [1,42,13] block serves all of them (ppenv_ta_params.
        # initial_rb_shared); share_initial_rb=False keeps the reference's per-env tensor.
        self.initial_rb_states = self._rb_states.clone()
        if share_initial_rb:
            assert torch.equal(self.initial_rb_states[:, :40], self.initial_rb_states[:1, :40].expand(n, 40, 13))
            self.initial_rb_states = self.initial_rb_states[:1].contiguous()
            self.params.initial_rb_shared = 1
        self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf = self.state.obs_buf, self.state.rew_buf, self.state.reset_buf, self.state.progress_buf
        self.reset_buf.fill_(1)   # upstream VecTask.allocate_buffers

    def step(self, actions, obs=None, rew=None, reset=None):
        """obs / rew / reset (fused mode only): tensors that receive this step's observations, rewards and reset flags instead of
        obs_buf / rew_buf / reset_buf — a rollout collector passes its horizon-major slices, so nothing is copied afterwards."""
        if actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if self.fused:
            self.sim.step(self.state, actions, self.initial_rb_states, self.root_states, self.dof_states,
                          self._rb_states if self.materialize_rb else None, self.dof_force_tensor, self.pre_ball_vx, obs=obs, rew=rew, reset=reset)
            if obs is not None or rew is not None or reset is not None:
                return ({"obs": self.obs_buf if obs is None else obs}, self.rew_buf if rew is None else rew,
                        self.reset_buf if reset is None else reset, {})
        else:
            assert obs is None and rew is None and reset is None, "output slices need the fused step"
            self.sim.simulate(actions, self.root_states, self.dof_states, self._rb_states, self.dof_force_tensor, self.pre_ball_vx)
            self.state.post_physics_step(self._rb_states, self.initial_rb_states, self.root_states, self.dof_states, self.dof_force_tensor, self.pre_ball_vx)
        return {"obs": self.obs_buf}, self.rew_buf, self.reset_buf, {}

    @property
    def rb_states(self):
        """rigid_body_states [N,42,13].  Materialised by the step when `materialize_rb`; otherwise gym.refresh_rigid_body_state_tensor
        on demand: forward kinematics of the current root / dof states (which, for an env that has just reset, are the reset ones)."""
        if not self.materialize_rb:
            self.sim.forward_kinematics(self.root_states, self.dof_states, self._rb_states)
        return self._rb_states

    def set_policy_input(self, out=None, mean=None, inv_std=None, clip=5.0):
        """The step kernel writes the policy's first-layer input itself (TASim.set_policy_input; NativeMLP.attach_env wires it)."""
        self.sim.set_policy_input(out, mean, inv_std, clip)

    def set_randomization(self, **kw):
        """Domain-randomisation tables + noise amplitudes for every later step (TASim.set_randomization; fused step on the chain-wave kernel)."""
        assert self.fused, "the randomisation tables are read by the fused step (ppenv_ta_step)"
        self.sim.set_randomization(**kw)

    def clear_randomization(self):
        self.sim.clear_randomization()

    def set_gravity(self, gravity_z):
        self.sim.set_gravity(gravity_z)

    def reset_idx(self, env_ids=None):
        """_reset_idx (TA:965-1028) outside a step, for the listed env ids (None: all).  A rare host-driven path: plain torch
        indexing on the task's own tensors with the draws of each env's next episode (the same keyed draws the kernel uses)."""
        n = self.num_envs
        ids = torch.arange(n) if env_ids is None else torch.as_tensor(env_ids, dtype=torch.int64).reshape(-1).cpu()
        if ids.numel() == 0:
            return
        if int(ids.min()) < 0 or int(ids.max()) >= n:
            raise IndexError(f"env id outside [0, {n})")
        ids = torch.unique(ids)
        dev_ids = ids.to(self.device)
        st = self.state
        ep = (st.episode[dev_ids].to(torch.int64) + 1).cpu()
        ov = scene.ta_reset_draws(self.params, ids, ep).to(self.device)               # y, z, vx, vy, vz (TA:976-979)
        init = torch.tensor([[self.params.init_root[a][k] for k in range(7)] for a in range(3)], dtype=torch.float32, device=self.device)
        rows = torch.zeros((ids.numel(), 3, 13), dtype=torch.float32, device=self.device)
        rows[:, :, 0:7] = init
        rows[:, 2, 1:3] = ov[:, 0:2]
        rows[:, 2, 7:10] = ov[:, 2:5]
        self.root_states[dev_ids] = rows                                              # TA:969-983
        dof0 = torch.tensor([[self.params.init_dof_pos[d], self.params.init_dof_vel[d]] for d in range(27)], dtype=torch.float32, device=self.device)
        self.dof_states[dev_ids] = dof0
        st.episode[dev_ids] = ep.to(self.device, torch.int32)
        st.progress_buf[dev_ids] = 0
        sticky = scene.TA_FLAG_PADDLE_COND | scene.TA_FLAG_HIT_TABLE_CALC | scene.TA_FLAG_DIE_PENALTY_CALC | scene.TA_FLAG_HUMANOID_DIE_CALC
        st.flags[dev_ids] = st.flags[dev_ids] & ~sticky                               # TA:1021-1024

    def close(self):
        self.sim.close()
