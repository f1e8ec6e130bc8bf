"""Fused step of the 4-actor variant (PPENV_VARIANT_T4: two humanoids, one ball; agent a of env e owns row 2e + a).

Reference pieces it is built from: the two reward functions T4:1113-1439 (pinned by tests/golden/rewards_T4.npz through
ppenv_t4_rewards, test_t4_golden.py), _reset_idx T4:853-912, the per-humanoid observation functions T4:770-803.  The
class leaves the two-agent wiring open (T4:743,786-803); the wiring tested here is the build's completion of it
(include/ppenv.h).  CPU tests run the kernels' arithmetic through the host shim; GPU tests run the three-wave kernel."""
import ctypes as C

import numpy as np
import pytest

import shim_binding as sb
from helpers import RTOL, ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
from isaacgym_amd import scene


def _check_step(got, o, t, oa, ra):
    np.testing.assert_array_equal(got.reset_buf, o.reset_buf, err_msg=f"reset step {t}")
    np.testing.assert_array_equal(got.progress_buf, o.progress_buf, err_msg=f"progress step {t}")
    np.testing.assert_array_equal(got.flags, o.flags, err_msg=f"flags step {t}")
    np.testing.assert_array_equal(got.episode, o.episode, err_msg=f"episode step {t}")
    assert_state_close(got, o, f"step {t}")
    assert_close(got.obs_buf, o.obs_buf, f"obs step {t}", atol=oa)
    assert_close(got.rew_buf, o.rew_buf, f"rew step {t}", atol=ra)


def test_t4_config_and_layout(oracle_lib):
    cfg = scene.build_config("T4", num_envs=8, seed=0)
    assert cfg.num_humanoids == 2
    # humanoid 2 stands at x = 3.5, turned by pi about z, facing humanoid 1 (T4:555-556)
    np.testing.assert_allclose(list(cfg.humanoid2_root_pos), [3.5, 0.0, 1.0])
    np.testing.assert_allclose(list(cfg.humanoid2_root_quat), [0, 0, 1, 0])
    o = oracle_lib.OracleEnv(cfg)
    assert o.obs_buf.shape == (16, 80) and o.rew_buf.shape == (16,) and o.dof_pos.shape == (14, 8) and o.flags.shape == (2, 8)
    # mirrored scene: in its own heading frame humanoid 2 sees its arm where humanoid 1 sees its own
    np.testing.assert_allclose(o.obs_buf[0, :60], o.obs_buf[1, :60], atol=2e-5)
    # the ball starts 3.15 m in front of humanoid 1 and 0.35 m in front of humanoid 2 (T4:625), on opposite sides in y
    np.testing.assert_allclose(o.obs_buf[0, 74:77], [3.15, -0.28, 0.1], atol=1e-5)
    np.testing.assert_allclose(o.obs_buf[1, 74:77], [0.35, 0.28, 0.1], atol=1e-5)
    # rigid-body tensor of the 4-actor scene: [N, 82, 13], paddles at 39 and 79, table 80, ball 81 (T4:169-172)
    rb = o.refresh_rigid_body_states()
    assert rb.shape == (8, 82, 13)
    np.testing.assert_allclose(rb[:, 81, 0:3], np.tile([3.15, -0.28, 1.1], (8, 1)), atol=1e-6)
    np.testing.assert_allclose(rb[:, 40, 0:3], np.tile([3.5, 0.0, 1.0], (8, 1)), atol=1e-6)
    assert o.refresh_root_states().shape == (8, 4, 13) and o.refresh_dof_states().shape == (8, 14, 2)
    # the two paddles mirror each other through the table centre line x = 1.75
    np.testing.assert_allclose(rb[:, 39, 0] + rb[:, 79, 0], 3.5, atol=1e-5)
    np.testing.assert_allclose(rb[:, 39, 1] + rb[:, 79, 1], 0.0, atol=1e-5)


def test_t4_kernel_arithmetic_matches_oracle_single_steps(oracle_lib):
    """ppenv_device.h compiled for the host (the code the HIP kernel runs) vs the oracle, restarted from the oracle's
    state every step."""
    n = 192
    cfg = scene.build_config("T4", num_envs=n, seed=7)
    o = oracle_lib.OracleEnv(cfg)
    s = sb.ShimEnv(cfg)
    rng = np.random.default_rng(1)
    oa, ra = obs_atol(), 2 * reward_atol(cfg)   # the power term sums 14 dofs
    resets, side2_events = 0, 0
    for t in range(200):
        actions = rng.uniform(-1.2, 1.2, (2 * n, 7)).astype(np.float32)
        s.copy_state_from(o)
        o.step(actions)
        s.step(actions)
        _check_step(s, o, t, oa, ra)
        resets += int(o.reset_buf.sum())
        side2_events += int((np.abs(o.rew_buf[1::2]) > 100).sum())
    assert resets > 100 and side2_events > 0
    # one shared reset / progress per env
    np.testing.assert_array_equal(o.reset_buf[0::2], o.reset_buf[1::2])
    np.testing.assert_array_equal(o.progress_buf[0::2], o.progress_buf[1::2])


def test_t4_humanoid1_side_reduces_to_tt_when_humanoid2_is_out_of_reach(oracle_lib):
    """With the ball never near humanoid 2 the physics of side 1 is TT's: same ball, same arm 1, same obs row."""
    n = 64
    c4 = scene.build_config("T4", num_envs=n, seed=3)
    ct = scene.build_config("TT", num_envs=n, seed=3)
    o4, ot = oracle_lib.OracleEnv(c4), oracle_lib.OracleEnv(ct)
    # start both from a state where the ball has left humanoid 2's bound (serve is towards humanoid 1)
    rng = np.random.default_rng(0)
    compared = 0
    for t in range(40):
        a1 = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        a4 = np.zeros((2 * n, 7), np.float32)
        a4[0::2] = a1
        far = o4.ball[0] < 2.6    # > 0.7 m from humanoid 2's shoulder: outside its broad-phase bound all step long
        ot.ball[:] = o4.ball
        ot.dof_pos[:], ot.dof_vel[:], ot.dof_force[:] = o4.dof_pos[:7], o4.dof_vel[:7], o4.dof_force[:7]
        ot.flags[:], ot.episode[:], ot.progress_buf[:] = o4.flags[0], o4.episode, o4.progress_buf[0::2]
        o4.step(a4)
        ot.step(a1)
        keep = far & (o4.ball[0] < 2.6) & (ot.reset_buf == 0)
        compared += int(keep.sum())
        np.testing.assert_array_equal(o4.ball[:, keep], ot.ball[:, keep])
        np.testing.assert_array_equal(o4.dof_pos[:7, keep], ot.dof_pos[:, keep])
        np.testing.assert_array_equal(o4.obs_buf[0::2][keep], ot.obs_buf[keep])
    assert compared > 500


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


class DevView:
    def __init__(self, env):
        self.dof_pos, self.dof_vel, self.dof_force = (t.cpu().numpy() for t in (env.dof_pos, env.dof_vel, env.dof_force))
        self.ball = env.ball.cpu().numpy()
        self.flags = env.flags.cpu().numpy().view(np.uint32)
        self.episode = env.episode.cpu().numpy().view(np.uint32)
        self.progress_buf = env.progress_buf.cpu().numpy()
        self.reset_buf = env.reset_buf.cpu().numpy()
        self.rew_buf = env.rew_buf.cpu().numpy()
        self.obs_buf = env.obs_buf.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule,n", [("split", 1024), ("split", 200), ("split3", 200), ("split", 8192)])   # 8192: BASELINE config 4's per-GPU size, as named
def test_t4_fused_step_matches_oracle_single_steps(torch_cuda, oracle_lib, monkeypatch, schedule, n):
    """The three-wave kernel (arm 1 / arm 2 / ball) in both hand-off forms — the arm waves sweep the collision geometry
    (default), or the ball wave does from the published (q, qd) — vs the oracle, restarted from the oracle's state every
    step; n = 200 leaves a ragged last workgroup."""
    torch = torch_cuda
    from isaacgym_amd.env import PPEnv
    monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)
    cfg = scene.build_config("T4", num_envs=n, seed=7)
    o = oracle_lib.OracleEnv(cfg, threads=8)
    env = PPEnv(scene.build_config("T4", num_envs=n, seed=7), device="cuda:0")
    assert env.obs_buf.shape == (2 * n, 80) and env.num_agents == 2
    v0 = DevView(env)
    assert_close(v0.obs_buf, o.obs_buf, "initial obs", atol=obs_atol())
    np.testing.assert_array_equal(v0.reset_buf, o.reset_buf)
    rng = np.random.default_rng(1)
    oa, ra = obs_atol(), 2 * reward_atol(cfg)
    probe = SensitivityProbe(oracle_lib, cfg)
    resets = 0
    log = ExclusionLog(f"gpu 4-actor step vs oracle [{schedule}]", bound=0.01)
    for t in range(160 if n < 4096 else 40):
        actions = rng.uniform(-1.2, 1.2, (2 * n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        o.step(actions)
        env.step(torch.from_numpy(actions).cuda())
        keep = ~probe.sensitive(st, actions, o)      # envs sitting on a switch of the physics spec this step are skipped
        log.add(keep)
        _check_step(mask_envs(DevView(env), keep, 2), mask_envs(o, keep, 2), t, oa, ra)
        resets += int(o.reset_buf.sum())
    assert resets > 50 or n >= 4096       # (the 40 steps of the full-size case end before the first episode does)
    log.close()
    # gym.refresh_* equivalents in the 4-actor layouts
    env.set_state(o.get_state())
    assert_close(env.refresh_rigid_body_states().cpu().numpy(), o.refresh_rigid_body_states(), "rb states", atol=2e-3)
    assert_close(env.refresh_root_states().cpu().numpy(), o.refresh_root_states(), "root states", atol=1e-6)
    assert_close(env.refresh_dof_states().cpu().numpy(), o.refresh_dof_states(), "dof states", atol=1e-6)
    assert_close(env.refresh_dof_force().cpu().numpy(), o.refresh_dof_force(), "dof force", atol=1e-6)
    env.close()


@pytest.mark.gpu
def test_t4_fused_rewards_equal_the_pinned_reward_kernel(torch_cuda):
    """The fused step's rew / reset / flags are what ppenv_t4_rewards (pinned to the reference's TorchScript functions by
    rewards_T4.npz) returns on the refreshed simulator tensors of the same step."""
    torch = torch_cuda
    from isaacgym_amd import _lib
    from isaacgym_amd.env import PPEnv
    n = 2048
    cfg = scene.build_config("T4", num_envs=n, seed=5)
    env = PPEnv(cfg, device="cuda:0")
    p = scene.build_t4_params(n, episode_length=cfg.max_episode_length, alpha=cfg.alpha_velocity_reward,
                              power_coefficient=cfg.power_coefficient, penalty=cfg.penalty,
                              hit_table_reward=cfg.hit_table_reward, not_hit_table_penalty=cfg.not_hit_table_penalty)
    L = _lib.lib()
    gen = torch.Generator(device="cuda").manual_seed(3)
    i64 = lambda: torch.zeros(n, dtype=torch.int64, device="cuda")
    f32 = lambda: torch.zeros(n, dtype=torch.float32, device="cuda")
    u32 = lambda: torch.zeros(n, dtype=torch.int32, device="cuda")
    checked = 0
    for t in range(150):
        a = torch.rand(2 * n, 7, device="cuda", generator=gen) * 2 - 1
        pre_vx = env.ball[7].clone()
        prog = env.progress_buf[0::2] + 1                                  # T4:1029
        f1_in, f2_in = env.flags[0].clone(), env.flags[1].clone()
        env.step(a)
        keep = env.reset_buf[0::2] == 0          # a reset env's tensors already show the next episode
        rb, root, dof, frc = env.refresh_rigid_body_states(), env.refresh_root_states(), env.refresh_dof_states(), env.refresh_dof_force()
        f1, f2, r1, r2, s1, s2 = u32(), u32(), f32(), f32(), i64(), i64()
        _lib.check(L.ppenv_t4_rewards(C.byref(p), rb.data_ptr(), root.data_ptr(), dof.data_ptr(), frc.data_ptr(), pre_vx.data_ptr(),
                                      prog.contiguous().data_ptr(), f1_in.data_ptr(), f2_in.data_ptr(), f1.data_ptr(), f2.data_ptr(),
                                      r1.data_ptr(), r2.data_ptr(), s1.data_ptr(), s2.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        # the refreshed paddle position goes through one more FK evaluation than the fused step's: compare with the fp32 bar
        assert_close(env.rew_buf[0::2][keep].cpu().numpy(), r1[keep].cpu().numpy(), f"rew1 step {t}", atol=1e-3)
        assert_close(env.rew_buf[1::2][keep].cpu().numpy(), r2[keep].cpu().numpy(), f"rew2 step {t}", atol=1e-3)
        assert torch.equal(env.flags[0][keep], f1[keep]) and torch.equal(env.flags[1][keep], f2[keep])
        assert torch.equal(env.reset_buf[0::2][keep], (s1 | s2)[keep])   # (envs that did reset are covered by the oracle test)
        checked += int(keep.sum())
    assert checked > 100000
    env.close()


@pytest.mark.gpu
def test_t4_long_run_and_determinism(torch_cuda):
    torch = torch_cuda
    from isaacgym_amd.env import PPEnv
    n = 8192   # BASELINE config 4: 8192 envs per GPU
    envs = [PPEnv(scene.build_config("T4", num_envs=n, seed=9), device="cuda:0") for _ in range(2)]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for t in range(600):
        a = torch.rand(2 * n, 7, device="cuda", generator=gen) * 2.4 - 1.2
        for e in envs:
            e.step(a)
    torch.cuda.synchronize()
    a_, b_ = envs
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "dof_pos", "dof_vel", "ball", "flags", "episode"):
        assert torch.equal(getattr(a_, name), getattr(b_, name)), name
    assert torch.isfinite(a_.obs_buf).all() and torch.isfinite(a_.rew_buf).all()
    assert int(a_.episode.sum()) > n                      # every env has been through resets
    assert float(a_.ball[2].min()) > -0.05 and float(a_.ball[0].abs().max()) < 20.0
    assert torch.equal(a_.progress_buf[0::2], a_.progress_buf[1::2])
    for e in envs:
        e.close()


@pytest.mark.gpu
def test_t4_vec_task_surface(torch_cuda):
    """`isaacgym_amd.make("Humanoid12PingpongTiltG1")`: two agents per env on the VecTask / rl_games surface."""
    torch = torch_cuda
    import isaacgym_amd
    n = 256
    task = isaacgym_amd.make(task="Humanoid12PingpongTiltG1", num_envs=n, seed=1)
    assert task.get_number_of_agents() == 2 and task.num_actions == 7 and task.num_obs == 80
    assert task.actors_per_env == 4 and task.dofs_per_env == 14 and task.rigid_bodies_per_env == 82   # T4:125-127
    obs = task.reset()["obs"]
    assert obs.shape == (2 * n, 80)
    a = task.zero_actions()
    assert a.shape == (2 * n, 7)
    for _ in range(5):
        od, rew, done, extras = task.step(torch.rand_like(a) * 2 - 1)
    assert od["obs"].shape == (2 * n, 80) and rew.shape == (2 * n,) and done.shape == (2 * n,) and done.dtype == torch.int64
    assert "time_outs" in extras and extras["time_outs"].shape == (2 * n,)
    task.refresh_sim_tensors()
    assert task.body_states.shape == (n, 82, 13) and task.humanoid2_paddle_rb_states.shape == (n, 13)
    assert task.reward_calculated.shape == (2, n)
    assert torch.equal(task.progress_buf[0::2], task.progress_buf[1::2])


@pytest.mark.gpu
def test_t4_fused_step_with_randomisation_matches_oracle(torch_cuda, oracle_lib):
    """Domain randomisation on the 4-actor task (round 3): an env's [7][N] table entries apply to both humanoids (two instances of the yaml's one "humanoid"
    actor), the 14 action draws and the two agents' observation rows have their own noise indices — the three-wave table-reading kernel against the oracle,
    restarted from the oracle's state every step; cleared, the plain kernel runs again bit for bit."""
    torch = torch_cuda
    from isaacgym_amd.env import PPEnv
    n = 512
    cfg = scene.build_config("T4", num_envs=n, seed=9)
    o = oracle_lib.OracleEnv(cfg, threads=8)
    env = PPEnv(scene.build_config("T4", num_envs=n, seed=9), device="cuda:0")
    ref = PPEnv(scene.build_config("T4", num_envs=n, seed=9), device="cuda:0")
    rng = np.random.default_rng(3)
    u = lambda lo, hi, shape: rng.uniform(lo, hi, shape).astype(np.float32)
    tabs = dict(dof_stiffness_scale=u(0.5, 1.5, (7, n)), dof_damping_scale=u(0.5, 1.5, (7, n)), link_mass_scale=u(0.5, 1.5, (7, n)),
                restitution_scale=u(0.0, 0.7, n), friction_scale=u(0.7, 1.3, n))
    kw = dict(action_noise_sigma=0.02, observation_noise_sigma=0.002)
    probe = SensitivityProbe(oracle_lib, cfg)
    for x in (o, probe.o2, env):
        x.set_randomization(**tabs, **kw)
    oa, ra = obs_atol() + 2e-6, 2 * reward_atol(cfg)
    log = ExclusionLog("gpu 4-actor step with domain randomisation vs oracle", bound=0.01)
    resets, moved = 0, 0.0
    for t in range(120):
        actions = rng.uniform(-1.2, 1.2, (2 * n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        if t == 5:
            ref.set_state(st)
            ref.step(torch.from_numpy(actions).cuda())
        o.step(actions)
        env.step(torch.from_numpy(actions).cuda())
        if t == 5:
            moved = float((ref.dof_vel - env.dof_vel).abs().max())
        keep = ~probe.sensitive(st, actions, o)
        log.add(keep)
        _check_step(mask_envs(DevView(env), keep, 2), mask_envs(o, keep, 2), t, oa, ra)
        resets += int(o.reset_buf.sum())
    log.close()
    assert resets > 30 and moved > 1e-2
    # the two agents of an env do not share their observation noise
    g = env.obs_buf.cpu().numpy().reshape(n, 2, 80)
    assert np.abs(g[:, 0, 60:67] - g[:, 1, 60:67]).max() > 1e-4
    env.clear_randomization()
    st = o.get_state()
    env.set_state(st); ref.set_state(st)
    a = torch.from_numpy(actions).cuda()
    env.step(a); ref.step(a)
    for name in ("obs_buf", "rew_buf", "dof_pos", "dof_vel", "ball"):
        assert torch.equal(getattr(env, name), getattr(ref, name)), name
    env.close(); ref.close()
