"""ctypes binding of tests/csrc/libppenv_hostshim.so — the HIP kernels' per-env arithmetic compiled for the host.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from isaacgym_amd import scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "host_shim.cpp")
_HDR = os.path.join(_HERE, "..", "isaacgym_amd", "csrc", "ppenv_device.h")
_HDR2 = os.path.join(_HERE, "..", "isaacgym_amd", "csrc", "ppenv_ta_device.h")
_LIB = os.path.join(_HERE, "csrc", "libppenv_hostshim.so")
_lib = None


def lib_for_model(header, out):
    """The host shim with ANOTHER arm model compiled in (-DPPENV_MODEL_HEADER, as isaacgym_amd._lib.build_for_arm_model does for
    the HIP library): `header` = modelgen.generate(config) written to a file; -> a CDLL."""
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=fast", "-fno-signed-zeros", "-ffinite-math-only", "-Wno-unknown-pragmas",
                    f'-DPPENV_MODEL_HEADER="{header}"', "-o", out, _SRC], check=True, capture_output=True)
    return C.CDLL(out)


def lib():
    global _lib
    if _lib is None:
        if (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR), os.path.getmtime(_HDR2)):
            subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=fast", "-fno-signed-zeros", "-ffinite-math-only", "-Wno-unknown-pragmas",
                            "-o", _LIB, _SRC], check=True, capture_output=True)
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ShimEnv:
    """Same SoA state layout as the oracle / the HIP handle; stepped by the kernel arithmetic on the CPU."""

    def __init__(self, config, L=None):
        self.L = L if L is not None else lib()
        self.config = config
        assert self.L.shim_model_matches(C.byref(config)) == 1, "the config's arm model is not the one compiled into this shim"
        n = self.num_envs = config.num_envs
        A = self.num_agents = config.num_humanoids
        nd = A * scene.NUM_DOF
        self.obs_buf = np.zeros((n * A, scene.NUM_OBS), np.float32)
        self.rew_buf = np.zeros(n * A, np.float32)
        self.reset_buf = np.zeros(n * A, np.int64)
        self.progress_buf = np.zeros(n * A, np.int64)
        self.dof_pos = np.zeros((nd, n), np.float32)
        self.dof_vel = np.zeros((nd, n), np.float32)
        self.dof_force = np.zeros((nd, n), np.float32)
        self.ball = np.zeros((13, n), np.float32)
        self.flags = np.zeros(n if A == 1 else (A, n), np.uint32)
        self.episode = np.zeros(n, np.uint32)
        self.serve = None
        self.bodies = np.zeros((n, A * scene.NUM_OBS_BODIES, 13), np.float32)

    def copy_state_from(self, other):
        for name in ("dof_pos", "dof_vel", "dof_force", "ball", "flags", "episode", "progress_buf", "reset_buf"):
            getattr(self, name)[...] = getattr(other, name)

    def set_serve_override(self, serve, on=True):
        self.serve = np.ascontiguousarray(np.asarray(serve, np.float32).T) if (on and serve is not None) else None

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32)
        self.L.shim_step(C.byref(self.config), _p(a), _p(self.dof_pos), _p(self.dof_vel), _p(self.dof_force), _p(self.ball),
                         _p(self.progress_buf), _p(self.flags), _p(self.episode),
                         _p(self.serve) if self.serve is not None else None,
                         _p(self.obs_buf), _p(self.rew_buf), _p(self.reset_buf), _p(self.bodies))


def arm_qdd(config, q, qd, tau, arm_eff):
    q, qd, tau, arm_eff = (np.ascontiguousarray(x, np.float32) for x in (q, qd, tau, arm_eff))
    out = np.zeros(scene.NUM_DOF, np.float32)
    lib().shim_arm_qdd(C.byref(config), _p(q), _p(qd), _p(tau), _p(arm_eff), _p(out))
    return out


def serve_velocity(config, gid, episode):
    out = np.zeros(3, np.float32)
    lib().shim_serve_velocity(C.byref(config), C.c_uint32(gid), C.c_uint32(episode), _p(out))
    return out


def ta_simulate(scene_cfg, model, actions, root, dof):
    """One step of the 27-DoF rigid-body kernel arithmetic on the CPU (root [N,3,13], dof [N,27,2] updated in place)."""
    n = root.shape[0]
    for a in (actions, root, dof):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    rb = np.zeros((n, 42, 13), np.float32)
    frc = np.zeros((n, 27), np.float32)
    pvx = np.zeros(n, np.float32)
    rc = lib().shim_ta_simulate(C.byref(scene_cfg), C.byref(model), n, _p(actions), _p(root), _p(dof), _p(rb), _p(frc), _p(pvx))
    assert rc == 0
    return rb, frc, pvx


def serve_from_draws(form, draws):
    """The kernels' serve_from_draws() (fp32) on [M,3] draws (speed, tilt deg, tilt_z deg)."""
    d = np.ascontiguousarray(draws, np.float32).reshape(-1, 3)
    out = np.zeros_like(d)
    lib().shim_serve_from_draws(int(form), d.shape[0], _p(d), _p(out))
    return out


def pd_targets(actions, lo, hi, clip):
    """The kernels' pd_target() on [M,D] actions."""
    a = np.ascontiguousarray(actions, np.float32)
    lo, hi = np.ascontiguousarray(lo, np.float32), np.ascontiguousarray(hi, np.float32)
    out = np.zeros_like(a)
    lib().shim_pd_targets(a.shape[0], a.shape[1], _p(a), _p(lo), _p(hi), C.c_float(clip), _p(out))
    return out
