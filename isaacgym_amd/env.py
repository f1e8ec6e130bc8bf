"""Low-level handle on the native environment: one `ppenv` per GPU, buffers wrapped as torch tensors.

The analogue of the reference's `gymtorch.wrap_tensor(acquire_*_tensor(sim))` block
(tasks/humanoid_pingpong_3_actor_tilt.py:131-134,153-208): the simulator owns nothing the
Python side copies — every tensor below aliases the device arena the library steps in place.
PyTorch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, scene


class PPEnv:
    def __init__(self, config, device=None, library=None):
        """library: a libppenv build other than the default one (_lib.load of _lib.build_for_arm_model's output: another arm
        model compiled in); the handle then only ever talks to that build."""
        self.config = config
        self.L = library if library is not None else _lib.lib()   # raises when the HIP extension is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.PPEnvError("no ROCm GPU visible to PyTorch; the native environment runs on an MI355X only")
        self.device = torch.device(device if device is not None else f"cuda:{config.device_id}")
        if self.device.type != "cuda":
            raise _lib.PPEnvError(f"sim device must be a GPU, got {self.device}")
        config.device_id = self.device.index if self.device.index is not None else torch.cuda.current_device()
        n = self.num_envs = config.num_envs
        nbytes = self.L.ppenv_arena_bytes(C.byref(config))
        # torch owns the arena (256-byte aligned by the caching allocator); the library steps it in place
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.ppenv_create(C.byref(config), self.arena.data_ptr(), nbytes, self._stream(), C.byref(self.h)), self.L)
        b = scene.Buffers()
        _lib.check(self.L.ppenv_buffers_of(self.h, C.byref(b)), self.L)
        base = self.arena.data_ptr()

        def view(ptr, count, dtype, shape):
            off = ptr - base
            itemsize = torch.empty((), dtype=dtype).element_size()
            return self.arena[off: off + count * itemsize].view(dtype).view(*shape)

        # A agents per env (2 for the 4-actor variant): agent a of env e owns row A*e + a of the surface tensors
        A = self.num_agents = b.num_agents
        rows = self.num_rows = n * A
        nd = self.num_dofs = A * scene.NUM_DOF
        self.obs_buf = view(b.obs_buf, rows * scene.NUM_OBS, torch.float32, (rows, scene.NUM_OBS))
        self.rew_buf = view(b.rew_buf, rows, torch.float32, (rows,))
        self.reset_buf = view(b.reset_buf, rows, torch.int64, (rows,))
        self.progress_buf = view(b.progress_buf, rows, torch.int64, (rows,))
        self.dof_pos = view(b.dof_pos, nd * n, torch.float32, (nd, n))       # SoA [7A][N]
        self.dof_vel = view(b.dof_vel, nd * n, torch.float32, (nd, n))
        self.dof_force = view(b.dof_force, nd * n, torch.float32, (nd, n))
        self.ball = view(b.ball, 13 * n, torch.float32, (13, n))             # SoA [13][N]
        self.flags = view(b.flags, A * n, torch.int32, (n,) if A == 1 else (A, n))
        self.episode = view(b.episode, n, torch.int32, (n,))

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.L.ppenv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path
    def step(self, actions, obs=None, rew=None, reset=None):
        """actions: float32 [A*N, 7] on this device, contiguous.  One fused kernel launch, no sync.
        obs [A*N, 80] f32 / rew [A*N] f32 / reset [A*N] int64: tensors that receive this step's outputs instead of obs_buf / rew_buf /
        reset_buf (ppenv_step_into: a rollout collector's horizon-major slices)."""
        if actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous() \
                or tuple(actions.shape) != (self.num_rows, scene.NUM_DOF):
            actions = actions.to(device=self.device, dtype=torch.float32).reshape(self.num_rows, scene.NUM_DOF).contiguous()
        if obs is None and rew is None and reset is None:
            _lib.check(self.L.ppenv_step(self.h, actions.data_ptr(), self._stream()), self.L)
            return
        for t, dt, numel in ((obs, torch.float32, self.num_rows * scene.NUM_OBS), (rew, torch.float32, self.num_rows), (reset, torch.int64, self.num_rows)):
            assert t is None or (t.dtype == dt and t.is_contiguous() and t.device == self.device and t.numel() == numel)
        p = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.L.ppenv_step_into(self.h, actions.data_ptr(), p(obs), p(rew), p(reset), self._stream()), self.L)

    def step_sequence(self, actions_list):
        """ppenv_step_sequence: one fused step per tensor of `actions_list` (float32 [A*N, 7], contiguous, on this device), launched back to back by one
        native call — an open-loop burst (scripted / logged / repeated actions) without a host round trip or a graph launch per burst."""
        import ctypes as C
        for a in actions_list:
            assert a.dtype == torch.float32 and a.device == self.device and a.is_contiguous() and tuple(a.shape) == (self.num_rows, scene.NUM_DOF)
        arr = (C.c_void_p * len(actions_list))(*[a.data_ptr() for a in actions_list])
        _lib.check(self.L.ppenv_step_sequence(self.h, arr, len(actions_list), self._stream()), self.L)

    def reset_all(self):
        _lib.check(self.L.ppenv_reset_all(self.h, self._stream()), self.L)

    def reset_idx(self, env_ids, refresh_obs=True):
        """reset_idx(env_ids) -> _reset_idx (TT:809-812, 847-906) for the listed local env ids only (int64 tensor / sequence)."""
        ids = torch.as_tensor(env_ids, dtype=torch.int64).reshape(-1)
        if ids.numel() == 0:
            return
        if ids.device.type != "cuda":        # host-visible ids are validated here; device ids are range-checked by the kernel
            if int(ids.min()) < 0 or int(ids.max()) >= self.num_envs:
                raise IndexError(f"env id outside [0, {self.num_envs})")
        ids = ids.to(self.device).contiguous()
        _lib.check(self.L.ppenv_reset_idx(self.h, ids.data_ptr(), ids.numel(), int(bool(refresh_obs)), self._stream()), self.L)
        ids.record_stream(torch.cuda.current_stream(self.device))   # the kernel reads `ids` after this frame is gone

    def pd_targets(self, actions):
        """pre_physics_step's PD targets (TT:1008-1014) for actions [A*N, 7]: what set_dof_position_target_tensor receives."""
        a = actions.to(device=self.device, dtype=torch.float32).reshape(self.num_rows, scene.NUM_DOF).contiguous()
        out = torch.empty_like(a)
        _lib.check(self.L.ppenv_pd_targets(self.h, a.data_ptr(), out.data_ptr(), self._stream()), self.L)
        return out

    def serve_from_draws(self, draws):
        """generate_random_speed_for_ball of this variant on [M,3] draws (speed, tilt deg, tilt_z deg) -> [M,3] velocities."""
        d = torch.as_tensor(draws, dtype=torch.float32).to(self.device).reshape(-1, 3).contiguous()
        out = torch.empty_like(d)
        _lib.check(self.L.ppenv_serve_from_draws(self.h, d.data_ptr(), d.shape[0], out.data_ptr(), self._stream()), self.L)
        return out

    def set_randomization(self, dof_stiffness_scale=None, dof_damping_scale=None, link_mass_scale=None, restitution_scale=None,
                          friction_scale=None, action_noise_sigma=0.0, observation_noise_sigma=0.0):
        """Per-env domain-randomisation tables (ppenv_set_randomization): float32 device tensors [7, N] / [N] (None = not randomised).
        The tensors are kept alive here and read by every following step; rewriting them in place changes the randomisation."""
        def tab(t, rows):
            if t is None:
                return None
            t = torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous()
            assert tuple(t.shape) == ((rows, self.num_envs) if rows else (self.num_envs,)), tuple(t.shape)
            return t
        self._dr = [tab(dof_stiffness_scale, 7), tab(dof_damping_scale, 7), tab(link_mass_scale, 7), tab(restitution_scale, 0), tab(friction_scale, 0)]
        r = scene.Randomization()
        (r.dof_stiffness_scale, r.dof_damping_scale, r.link_mass_scale, r.restitution_scale, r.friction_scale) = [t.data_ptr() if t is not None else None for t in self._dr]
        r.action_noise_sigma, r.observation_noise_sigma = float(action_noise_sigma), float(observation_noise_sigma)
        _lib.check(self.L.ppenv_set_randomization(self.h, C.byref(r)), self.L)

    def clear_randomization(self):
        _lib.check(self.L.ppenv_set_randomization(self.h, None), self.L)
        self._dr = None

    def set_gravity(self, gravity_z):
        _lib.check(self.L.ppenv_set_gravity(self.h, float(gravity_z)), self.L)

    @property
    def status(self):
        """PPENV_STATUS_* bits reported by the kernels (0 = healthy); read without synchronising."""
        return int(self.L.ppenv_status(self.h))

    @property
    def step_kernel_name(self):
        """The kernel ppenv_step launches for this handle now, as rocprofv3's kernel trace names it (ppenv_step_kernel_name)."""
        return self.L.ppenv_step_kernel_name(self.h).decode()

    def reduce_stats(self, out=None):
        """float64[4] on the device: sum rew_buf, sum progress_buf, sum episode, num_envs (one reduction launch)."""
        if out is None:
            if not hasattr(self, "_stats"):
                self._stats = torch.zeros(4, dtype=torch.float64, device=self.device)
            out = self._stats
        assert out.dtype == torch.float64 and out.numel() == 4 and out.device == self.device and out.is_contiguous()
        _lib.check(self.L.ppenv_reduce_stats(self.h, out.data_ptr(), self._stream()), self.L)
        return out

    # ---- Isaac-Gym tensor-API mode
    def post_physics_step(self, rigid_body_states, root_states, dof_states, dof_force, pre_ball_vx):
        for t in (rigid_body_states, root_states, dof_states, dof_force, pre_ball_vx):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        n = self.num_envs
        assert rigid_body_states.numel() == n * scene.NUM_BODIES * 13 and root_states.numel() == n * scene.NUM_ACTORS * 13
        assert dof_states.numel() == n * scene.NUM_DOF * 2 and dof_force.numel() == n * scene.NUM_DOF and pre_ball_vx.numel() == n
        _lib.check(self.L.ppenv_post_physics_step(self.h, rigid_body_states.data_ptr(), root_states.data_ptr(), dof_states.data_ptr(),
                                                  dof_force.data_ptr(), pre_ball_vx.data_ptr(), self._stream()), self.L)

    def _refresh(self, fn, shape):
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(fn(self.h, out.data_ptr(), self._stream()), self.L)
        return out

    def refresh_root_states(self):
        return self._refresh(self.L.ppenv_refresh_root_states, (self.num_envs, self.num_agents + 2, 13))

    def refresh_dof_states(self):
        return self._refresh(self.L.ppenv_refresh_dof_states, (self.num_envs, self.num_dofs, 2))

    def refresh_dof_force(self):
        return self._refresh(self.L.ppenv_refresh_dof_force, (self.num_envs, self.num_dofs))

    def refresh_rigid_body_states(self):
        return self._refresh(self.L.ppenv_refresh_rigid_body_states, (self.num_envs, self.num_agents * scene.NUM_HUMANOID_BODIES + 2, 13))

    # ---- state I/O
    def set_serve_override(self, serve, on=True):
        """serve: [N,3] tensor/array of serve velocities used at the next resets instead of the RNG."""
        if serve is None or not on:
            _lib.check(self.L.ppenv_set_serve_override(self.h, None, int(bool(on)), self._stream()), self.L)
            return
        s = torch.as_tensor(serve, dtype=torch.float32).to(self.device).reshape(self.num_envs, 3).contiguous()
        _lib.check(self.L.ppenv_set_serve_override(self.h, s.data_ptr(), 1, self._stream()), self.L)
        torch.cuda.current_stream(self.device).synchronize()   # `s` must outlive the transpose kernel

    def get_state(self):
        n = self.L.ppenv_state_bytes(self.h)
        buf = np.empty(n, np.uint8)
        _lib.check(self.L.ppenv_get_state(self.h, buf.ctypes.data, n), self.L)
        return buf

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        _lib.check(self.L.ppenv_set_state(self.h, blob.ctypes.data, blob.size), self.L)
