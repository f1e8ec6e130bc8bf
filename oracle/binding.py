"""ctypes binding of oracle/libppenv_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package (isaacgym_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from isaacgym_amd import scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppenv_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ppenv_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "ppenv.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


class OracleBuffers(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_agents", C.c_int32),
        ("obs_buf", C.POINTER(C.c_float)), ("rew_buf", C.POINTER(C.c_float)),
        ("reset_buf", C.POINTER(C.c_int64)), ("progress_buf", C.POINTER(C.c_int64)),
        ("dof_pos", C.POINTER(C.c_float)), ("dof_vel", C.POINTER(C.c_float)),
        ("dof_force", C.POINTER(C.c_float)), ("ball", C.POINTER(C.c_float)),
        ("flags", C.POINTER(C.c_uint32)), ("episode", C.POINTER(C.c_uint32)),
        ("serve_override", C.POINTER(C.c_float)),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ppo_create.restype = C.c_void_p
        L.ppo_create.argtypes = [C.POINTER(scene.Config)]
        for name in ("ppo_destroy", "ppo_init", "ppo_reset_all"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.ppo_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.ppo_buffers_of.argtypes = [C.c_void_p, C.POINTER(OracleBuffers)]
        L.ppo_step.argtypes = [C.c_void_p, C.c_void_p]
        L.ppo_set_serve_override.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ppo_post_physics_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        for name in ("ppo_refresh_root_states", "ppo_refresh_dof_states", "ppo_refresh_dof_force",
                     "ppo_refresh_rigid_body_states"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.ppo_state_bytes.restype = C.c_size_t
        L.ppo_state_bytes.argtypes = [C.c_void_p]
        L.ppo_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ppo_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        dp = C.POINTER(C.c_double)
        L.ppo_arm_qdd.argtypes = [C.POINTER(scene.Config), dp, dp, dp, dp, dp]
        L.ppo_arm_inverse_dynamics.argtypes = [C.POINTER(scene.Config), dp, dp, dp, dp]
        L.ppo_arm_body_states.argtypes = [C.POINTER(scene.Config), dp, dp, C.POINTER(C.c_float)]
        L.ppo_arm_energy.restype = C.c_double
        L.ppo_arm_energy.argtypes = [C.POINTER(scene.Config), dp, dp]
        L.ppo_serve_velocity.argtypes = [C.POINTER(scene.Config), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.ppo_compute_obs.argtypes = [C.c_void_p] * 5
        _lib = L
    return _lib


def _np_view(ptr, shape, dtype):
    n = int(np.prod(shape))
    arr = np.ctypeslib.as_array(ptr, shape=(n,))
    return arr.view(dtype).reshape(shape)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleEnv:
    """numpy-facing wrapper; array attributes alias the oracle's memory."""

    def __init__(self, config, threads=1):
        self.L = lib()
        self.config = config
        self.h = self.L.ppo_create(C.byref(config))
        if not self.h:
            raise ValueError("ppo_create rejected the config")
        self.L.ppo_set_threads(self.h, threads)
        self.L.ppo_init(self.h)
        b = OracleBuffers()
        self.L.ppo_buffers_of(self.h, C.byref(b))
        n = self.num_envs = b.num_envs
        A = self.num_agents = b.num_agents            # 2 for the 4-actor variant: agent a of env e owns row A*e + a
        nd = self.num_dofs = A * scene.NUM_DOF
        self.obs_buf = _np_view(b.obs_buf, (n * A, scene.NUM_OBS), np.float32)
        self.rew_buf = _np_view(b.rew_buf, (n * A,), np.float32)
        self.reset_buf = _np_view(b.reset_buf, (n * A,), np.int64)
        self.progress_buf = _np_view(b.progress_buf, (n * A,), np.int64)
        self.dof_pos = _np_view(b.dof_pos, (nd, n), np.float32)
        self.dof_vel = _np_view(b.dof_vel, (nd, n), np.float32)
        self.dof_force = _np_view(b.dof_force, (nd, n), np.float32)
        self.ball = _np_view(b.ball, (13, n), np.float32)
        self.flags = _np_view(b.flags, (n,) if A == 1 else (A, n), np.uint32)
        self.episode = _np_view(b.episode, (n,), np.uint32)

    def close(self):
        if self.h:
            self.L.ppo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, t):
        self.L.ppo_set_threads(self.h, t)

    def set_randomization(self, dof_stiffness_scale=None, dof_damping_scale=None, link_mass_scale=None, restitution_scale=None,
                          friction_scale=None, action_noise_sigma=0.0, observation_noise_sigma=0.0, on=True):
        tabs = [None if t is None else np.ascontiguousarray(t, np.float32) for t in
                (dof_stiffness_scale, dof_damping_scale, link_mass_scale, restitution_scale, friction_scale)]
        self.L.ppo_set_randomization.argtypes = [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_int]
        rc = self.L.ppo_set_randomization(self.h, *[t.ctypes.data_as(C.c_void_p) if t is not None else None for t in tabs],
                                          C.c_float(action_noise_sigma), C.c_float(observation_noise_sigma), int(on))
        assert rc == 0, rc

    def set_gravity(self, gz):
        self.L.ppo_set_gravity.argtypes = [C.c_void_p, C.c_float]
        self.L.ppo_set_gravity.restype = None
        self.L.ppo_set_gravity(self.h, C.c_float(gz))

    def reset_idx(self, env_ids, refresh_obs=True):
        ids = np.ascontiguousarray(env_ids, np.int64).reshape(-1)
        self.L.ppo_reset_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        if self.L.ppo_reset_idx(self.h, ids.ctypes.data_as(C.c_void_p), ids.size, int(bool(refresh_obs))) != 0:
            raise IndexError("env id out of range")

    def step(self, actions):
        a = _f32(actions)
        assert a.shape == (self.num_envs * self.num_agents, scene.NUM_DOF)
        self.L.ppo_step(self.h, a.ctypes.data)

    def reset_all(self):
        self.L.ppo_reset_all(self.h)

    def set_serve_override(self, serve, on=True):
        if serve is None:
            self.L.ppo_set_serve_override(self.h, None, int(on))
        else:
            s = _f32(serve)
            assert s.shape == (self.num_envs, 3)
            self.L.ppo_set_serve_override(self.h, s.ctypes.data, int(on))

    def post_physics_step(self, rb_states, root_states, dof_states, dof_force, pre_ball_vx):
        """root_states / dof_states must be C-contiguous float32 arrays; they are updated in place."""
        for a in (rb_states, root_states, dof_states, dof_force, pre_ball_vx):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        self.L.ppo_post_physics_step(self.h, rb_states.ctypes.data, root_states.ctypes.data, dof_states.ctypes.data,
                                     dof_force.ctypes.data, pre_ball_vx.ctypes.data)

    def refresh_root_states(self):
        out = np.empty((self.num_envs, self.num_agents + 2, 13), np.float32)
        self.L.ppo_refresh_root_states(self.h, out.ctypes.data)
        return out

    def refresh_dof_states(self):
        out = np.empty((self.num_envs, self.num_dofs, 2), np.float32)
        self.L.ppo_refresh_dof_states(self.h, out.ctypes.data)
        return out

    def refresh_dof_force(self):
        out = np.empty((self.num_envs, self.num_dofs), np.float32)
        self.L.ppo_refresh_dof_force(self.h, out.ctypes.data)
        return out

    def refresh_rigid_body_states(self):
        out = np.empty((self.num_envs, self.num_agents * scene.NUM_HUMANOID_BODIES + 2, 13), np.float32)
        self.L.ppo_refresh_rigid_body_states(self.h, out.ctypes.data)
        return out

    def get_state(self):
        n = self.L.ppo_state_bytes(self.h)
        buf = np.empty(n, np.uint8)
        rc = self.L.ppo_get_state(self.h, buf.ctypes.data, n)
        assert rc == 0
        return buf

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        rc = self.L.ppo_set_state(self.h, blob.ctypes.data, blob.size)
        if rc != 0:
            raise ValueError("state blob size mismatch")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def arm_qdd(config, q, qd, tau, arm_eff):
    q, qd, tau, arm_eff = (np.ascontiguousarray(x, np.float64) for x in (q, qd, tau, arm_eff))
    out = np.zeros(scene.NUM_DOF)
    lib().ppo_arm_qdd(C.byref(config), _dp(q), _dp(qd), _dp(tau), _dp(arm_eff), _dp(out))
    return out


def arm_inverse_dynamics(config, q, qd, qdd):
    q, qd, qdd = (np.ascontiguousarray(x, np.float64) for x in (q, qd, qdd))
    out = np.zeros(scene.NUM_DOF)
    lib().ppo_arm_inverse_dynamics(C.byref(config), _dp(q), _dp(qd), _dp(qdd), _dp(out))
    return out


def arm_body_states(config, q, qd):
    q, qd = (np.ascontiguousarray(x, np.float64) for x in (q, qd))
    out = np.zeros((scene.NUM_OBS_BODIES, 13), np.float32)
    lib().ppo_arm_body_states(C.byref(config), _dp(q), _dp(qd), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def arm_energy(config, q, qd):
    q, qd = (np.ascontiguousarray(x, np.float64) for x in (q, qd))
    return lib().ppo_arm_energy(C.byref(config), _dp(q), _dp(qd))


def serve_velocity(config, gid, episode):
    out = (C.c_float * 3)()
    lib().ppo_serve_velocity(C.byref(config), gid, episode, out)
    return np.array(list(out), np.float32)


def serve_from_draws(form, draws):
    """generate_random_speed_for_ball of variant `form` (scene.VARIANT_IDS value) on [M,3] draws (speed, tilt deg, tilt_z deg)."""
    d = np.ascontiguousarray(draws, np.float64).reshape(-1, 3)
    out = np.zeros((d.shape[0], 3), np.float32)
    lib().ppo_serve_from_draws(int(form), d.shape[0], d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def pd_targets(actions, lo, hi, clip):
    """pre_physics_step's PD targets for [M,D] actions and joint limits lo / hi [D] (after the clipActions clamp)."""
    a = _f32(actions)
    lo, hi = _f32(lo), _f32(hi)
    out = np.zeros_like(a)
    lib().ppo_pd_targets(a.shape[0], a.shape[1], a.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                         C.c_float(clip), out.ctypes.data_as(C.c_void_p))
    return out


def compute_obs(bodies, dof_pos, dof_vel, ball):
    bodies, dof_pos, dof_vel, ball = (_f32(x) for x in (bodies, dof_pos, dof_vel, ball))
    out = np.zeros(scene.NUM_OBS, np.float32)
    lib().ppo_compute_obs(bodies.ctypes.data, dof_pos.ctypes.data, dof_vel.ctypes.data, ball.ctypes.data, out.ctypes.data)
    return out


# ---- 27-DoF variant, tensor-API mode -------------------------------------------------------------
def ta_post_physics_step(params, rb, irb, root, dof, dof_force, pre_vx, reset_override, flags, episode, progress):
    """numpy in / numpy out wrapper of ppo_ta_post_physics_step.  root, dof, flags, episode, progress are updated in place."""
    L = lib()
    n = params.num_envs
    obs = np.zeros((n, scene.TA_NUM_OBS), np.float32)
    rew = np.zeros(n, np.float32)
    reset = np.zeros(n, np.int64)
    for a in (rb, irb, root, dof, dof_force, pre_vx):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    ov = None if reset_override is None else np.ascontiguousarray(reset_override, np.float32)
    L.ppo_ta_post_physics_step.argtypes = [C.POINTER(scene.TAParams)] + [C.c_void_p] * 13
    L.ppo_ta_post_physics_step.restype = None
    L.ppo_ta_post_physics_step(C.byref(params), rb.ctypes.data, irb.ctypes.data, root.ctypes.data, dof.ctypes.data,
                               dof_force.ctypes.data, pre_vx.ctypes.data, ov.ctypes.data if ov is not None else None,
                               flags.ctypes.data, episode.ctypes.data, progress.ctypes.data, obs.ctypes.data, rew.ctypes.data,
                               reset.ctypes.data)
    return obs, rew, reset


# ---- 4-actor variant: the two reward functions -----------------------------------------------------
def t4_rewards(params, rb, root, dof, dof_force, pre_vx, progress, flags1, flags2):
    """Returns rew1, rew2, reset1, reset2, flags1_out, flags2_out (inputs are not modified)."""
    L = lib()
    n = params.num_envs
    f1o, f2o = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    rew1, rew2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    reset1, reset2 = np.zeros(n, np.int64), np.zeros(n, np.int64)
    L.ppo_t4_rewards.argtypes = [C.POINTER(scene.T4Params)] + [C.c_void_p] * 14
    L.ppo_t4_rewards.restype = None
    L.ppo_t4_rewards(C.byref(params), rb.ctypes.data, root.ctypes.data, dof.ctypes.data, dof_force.ctypes.data, pre_vx.ctypes.data,
                     progress.ctypes.data, flags1.ctypes.data, flags2.ctypes.data, f1o.ctypes.data, f2o.ctypes.data, rew1.ctypes.data,
                     rew2.ctypes.data, reset1.ctypes.data, reset2.ctypes.data)
    return rew1, rew2, reset1, reset2, f1o, f2o


# ---- 27-DoF variant: the rigid-body step ------------------------------------------------------------
def ta_simulate(scene_cfg, model, actions, root, dof, threads=1):
    """One pre_physics_step + simulate + refresh of the oracle.  root [N,3,13] and dof [N,27,2] are updated in place;
    returns rb_states [N,42,13], dof_force [N,27], pre_ball_vx [N]."""
    L = lib()
    n = root.shape[0]
    for a in (actions, root, dof):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    rb = np.zeros((n, 42, 13), np.float32)
    frc = np.zeros((n, 27), np.float32)
    pvx = np.zeros(n, np.float32)
    L.ppo_ta_simulate.argtypes = [C.POINTER(scene.Config), C.POINTER(scene.TAModel), C.c_int, C.c_int] + [C.c_void_p] * 6
    L.ppo_ta_simulate.restype = None
    L.ppo_ta_simulate(C.byref(scene_cfg), C.byref(model), n, threads, actions.ctypes.data, root.ctypes.data, dof.ctypes.data,
                      rb.ctypes.data, frc.ctypes.data, pvx.ctypes.data)
    return rb, frc, pvx


def ta_simulate_dr(scene_cfg, model, actions, root, dof, episode, progress, seed, env_id_offset=0, dof_stiffness_scale=None, dof_damping_scale=None,
                   link_mass_scale=None, restitution_scale=None, friction_scale=None, action_noise_sigma=0.0, threads=1):
    """ta_simulate with the domain-randomisation tables of ppenv_ta_randomization ([27, N] / [28, N] / [N] float32, None = not randomised) and the
    action noise keyed by (seed, env id, episode, progress at the step's start)."""
    L = lib()
    n = root.shape[0]
    for a in (actions, root, dof):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    tabs = [None if t is None else np.ascontiguousarray(t, np.float32) for t in (dof_stiffness_scale, dof_damping_scale, link_mass_scale, restitution_scale, friction_scale)]
    for t, rows in zip(tabs, (27, 27, 28, 0, 0)):
        assert t is None or t.shape == ((rows, n) if rows else (n,))
    ep, pr = np.ascontiguousarray(episode, np.uint32), np.ascontiguousarray(progress, np.int64)
    rb, frc, pvx = np.zeros((n, 42, 13), np.float32), np.zeros((n, 27), np.float32), np.zeros(n, np.float32)
    L.ppo_ta_simulate_dr.argtypes = [C.POINTER(scene.Config), C.POINTER(scene.TAModel), C.c_int, C.c_int] + [C.c_void_p] * 11 + [C.c_float, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    L.ppo_ta_simulate_dr.restype = None
    L.ppo_ta_simulate_dr(C.byref(scene_cfg), C.byref(model), n, threads, actions.ctypes.data, root.ctypes.data, dof.ctypes.data, rb.ctypes.data, frc.ctypes.data,
                         pvx.ctypes.data, *[t.ctypes.data if t is not None else None for t in tabs], float(action_noise_sigma), int(seed), int(env_id_offset),
                         ep.ctypes.data, pr.ctypes.data)
    return rb, frc, pvx


def ta_add_obs_noise(obs, sigma, seed, episode0, progress0, env_id_offset=0):
    """The observation noise of the 27-dof task's randomisation, in place on obs [N, 313]: keys = episode / progress at the step's start."""
    L = lib()
    assert obs.dtype == np.float32 and obs.flags.c_contiguous
    ep, pr = np.ascontiguousarray(episode0, np.uint32), np.ascontiguousarray(progress0, np.int64)
    L.ppo_ta_add_obs_noise.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    L.ppo_ta_add_obs_noise.restype = None
    L.ppo_ta_add_obs_noise(obs.ctypes.data, obs.shape[0], float(sigma), int(seed), int(env_id_offset), ep.ctypes.data, pr.ctypes.data)


def ta_forward_kinematics(model, root, dof):
    L = lib()
    n = root.shape[0]
    rb = np.zeros((n, 42, 13), np.float32)
    L.ppo_ta_forward_kinematics.argtypes = [C.POINTER(scene.TAModel), C.c_int] + [C.c_void_p] * 3
    L.ppo_ta_forward_kinematics.restype = None
    L.ppo_ta_forward_kinematics(C.byref(model), n, root.ctypes.data, dof.ctypes.data, rb.ctypes.data)
    return rb
