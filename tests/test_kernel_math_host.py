"""The HIP kernels' per-env arithmetic (isaacgym_amd/csrc/ppenv_device.h, compiled for the host by
tests/csrc/host_shim.cpp) against the CPU oracle.  Runs without a GPU; the same comparisons run on
the real kernels in test_gpu_parity.py."""
import numpy as np
import pytest

import shim_binding as sb
from helpers import RTOL, ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
from isaacgym_amd import scene


def test_aba_matches_oracle_forward_dynamics(oracle_lib):
    """fp32 articulated-body algorithm (kernel) vs fp64 Newton-Euler + dense solve (oracle)."""
    cfg = scene.build_config("TT", num_envs=1)
    rng = np.random.default_rng(0)
    for _ in range(300):
        q = rng.uniform(-1.5, 1.5, 7).astype(np.float32)
        qd = rng.uniform(-10, 10, 7).astype(np.float32)
        tau = rng.uniform(-25, 25, 7).astype(np.float32)
        arm = rng.uniform(0, 0.01, 7).astype(np.float32)
        want = oracle_lib.arm_qdd(cfg, q.astype(np.float64), qd.astype(np.float64), tau.astype(np.float64), arm.astype(np.float64))
        got = sb.arm_qdd(cfg, q, qd, tau, arm)
        assert_close(got, want, "qdd", rtol=RTOL, atol=RTOL * np.abs(want).max())


def test_serve_velocity_matches_oracle(oracle_lib):
    for variant in ("T3", "TT", "TN"):
        cfg = scene.build_config(variant, num_envs=1, seed=1234)
        for gid in (0, 1, 77, 16383, 65535):
            for ep in (0, 1, 2, 1000):
                assert_close(sb.serve_velocity(cfg, gid, ep), oracle_lib.serve_velocity(cfg, gid, ep), "serve", atol=2e-6)


@pytest.mark.parametrize("variant", ["TT", "TN", "T3"])
def test_single_step_parity_vs_oracle(oracle_lib, variant):
    """Every step starts both implementations from the oracle's state, so errors do not compound."""
    n = 256
    cfg = scene.build_config(variant, num_envs=n, seed=7)
    o = oracle_lib.OracleEnv(cfg)
    s = sb.ShimEnv(cfg)
    rng = np.random.default_rng(1)
    oa, ra = obs_atol(), reward_atol(cfg)
    resets = 0
    log = ExclusionLog(f"host shim (kernel arithmetic) vs oracle [{variant}]", bound=0.005)
    steps = 180 if variant == "TN" else 120   # TN only ever resets on its 170-step time-out (TN:1317)
    probe = SensitivityProbe(oracle_lib, cfg)
    for t in range(steps):
        actions = rng.uniform(-1.2, 1.2, (n, 7)).astype(np.float32)   # beyond +-1: exercises clipActions
        s.copy_state_from(o)
        st = o.get_state()
        o.step(actions)
        s.step(actions)
        keep = ~probe.sensitive(st, actions, o)   # envs within rounding of a contact switch this step (helpers.SensitivityProbe)
        log.add(keep)
        sm, om = mask_envs(s, keep), mask_envs(o, keep)
        np.testing.assert_array_equal(sm.reset_buf, om.reset_buf, err_msg=f"reset step {t}")
        np.testing.assert_array_equal(sm.progress_buf, om.progress_buf, err_msg=f"progress step {t}")
        np.testing.assert_array_equal(sm.flags, om.flags, err_msg=f"flags step {t}")
        np.testing.assert_array_equal(sm.episode, om.episode, err_msg=f"episode step {t}")
        assert_state_close(sm, om, f"step {t}")
        assert_close(sm.obs_buf, om.obs_buf, f"obs step {t}", atol=oa)
        assert_close(sm.rew_buf, om.rew_buf, f"rew step {t}", atol=ra)
        resets += int(o.reset_buf.sum())
    assert resets > 100
    log.close()


def test_gentle_policy_single_step_is_tight(oracle_lib):
    """With small smooth actions (no flailing at the limits) plain rtol 1e-4 / atol 1e-5 holds."""
    n = 128
    cfg = scene.build_config("TT", num_envs=n, seed=3)
    o = oracle_lib.OracleEnv(cfg)
    s = sb.ShimEnv(cfg)
    rng = np.random.default_rng(2)
    a = np.zeros((n, 7), np.float32)
    for t in range(100):
        a = np.clip(a + rng.normal(0, 0.02, (n, 7)), -0.3, 0.3).astype(np.float32)
        s.copy_state_from(o)
        o.step(a)
        s.step(a)
        np.testing.assert_array_equal(s.reset_buf, o.reset_buf)
        assert_close(s.dof_pos, o.dof_pos, "dof_pos", atol=1e-5)
        assert_close(s.dof_vel, o.dof_vel, "dof_vel", atol=1e-4)
        assert_close(s.ball[0:3], o.ball[0:3], "ball pos", atol=1e-5)
        # a ball on the table's edge: the normal is (centre - edge) / 0.02 m, so the 1.2e-7 m quantum of an fp32 x ~ 1.4 m
        # turns into 1e-5 of normal and, at 8 m/s, 2e-4 m/s of rebound velocity
        assert_close(s.ball[7:10], o.ball[7:10], "ball vel", atol=5e-4)
        oa = np.full(80, 1e-4)
        oa[77:80] = 5e-4   # ball velocity columns (TT:1660), as above
        assert_close(s.obs_buf, o.obs_buf, "obs", atol=oa)


def test_free_running_rollout_statistics(oracle_lib):
    """Trajectories diverge chaotically after contacts, so a free-running comparison is statistical."""
    n = 512
    cfg = scene.build_config("TT", num_envs=n, seed=11)
    o = oracle_lib.OracleEnv(cfg)
    s = sb.ShimEnv(cfg)
    s.copy_state_from(o)
    rng = np.random.default_rng(5)
    tot = np.zeros(2)
    resets = np.zeros(2)
    for t in range(300):
        actions = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        o.step(actions)
        s.step(actions)
        tot += [o.rew_buf.mean(), s.rew_buf.mean()]
        resets += [o.reset_buf.sum(), s.reset_buf.sum()]
    assert abs(resets[0] - resets[1]) <= 0.05 * resets[0] + 5
    assert abs(tot[0] - tot[1]) <= 0.10 * abs(tot[0]) + 5.0
