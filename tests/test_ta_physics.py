"""Rigid-body step of the 27-DoF variant (ppenv_ta_simulate: free-floating 28-link humanoid, ground contacts, ball).

PARITY UNPINNED against the reference (its physics is the closed Isaac Gym binary; the g1_27dof.urdf asset is absent): the
tests pin the kernels' floating-base articulated-body algorithm (fp32) to the oracle's Newton-Euler + dense solve (fp64) —
two independent formulations of the build's written specification — and check physical invariants."""
import numpy as np
import pytest

import shim_binding as sb
from helpers import ExclusionLog, assert_close
from isaacgym_amd import scene


def initial_tensors(n, seed=0):
    p = scene.build_ta_params(n)
    root = np.zeros((n, 3, 13), np.float32)
    for a in range(3):
        root[:, a, :7] = np.array(list(p.init_root[a]))
    rng = np.random.default_rng(seed)
    root[:, 2, 1] = rng.uniform(-0.5, 0.1, n)
    root[:, 2, 2] = rng.uniform(0.96, 1.05, n)
    root[:, 2, 7:10] = np.stack([rng.uniform(-5.2, -4.6, n), rng.uniform(-0.5, 0.2, n), rng.uniform(1.2, 2.0, n)], axis=1)
    return root, np.zeros((n, 27, 2), np.float32)


# Tolerances of the single-step comparison: rtol 1e-4 plus a fraction of each tensor's range.  Positions hold 1e-4 of the
# range; velocities and torques get 5e-4: the ankle links (74 g, I = 1e-5 kg m^2) under the 1e5 N/m contact springs amplify
# fp32 rounding — a handful of env-steps in 10^4 reach 1.5e-2 rad/s on a joint moving at 8 rad/s.  Every switch of the TA
# physics specification is a ramp (contact fade-in, clamped implicit PD, limit spring with a toe), so no env has to be skipped.
TOL = dict(root_pos=1e-4 * 1.0, root_quat=2e-4, root_vel=5e-4 * 10.0, q=1e-4 * 3.14, qd=5e-4 * 40.0, frc=5e-4 * 139.0, rb_pos=1e-4,
           rb_vel=5e-4 * 20.0, rb_ang=5e-4 * 100.0)   # a link's angular velocity stacks up to seven joint rates of <= 37 rad/s


def ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root_after, rel=1e-6, seed=7, dof_after=None):
    """Envs whose ball sits on a switch of the contact model in this step, decided by the ORACLE alone (as helpers.SensitivityProbe
    does for the 7-dof tasks): two more oracle steps from the same state with the continuous inputs jittered by `rel`; an env whose
    oracle ball velocity moves by more than 1e-3 m/s under that jitter took a discrete decision (contact on / off, bounce threshold)
    that fp32 and fp64 may legitimately take differently.  -> bool [N], True = on a switch.
    dof_after (the oracle's stepped dof tensor; only the tests of an asset that does NOT stand level pass it): also set aside an env whose
    ORACLE joint velocities move by more than a third of their tolerance under the same jitter — a sole corner on the steep part of the
    contact ramp (round 4: the second URDF asset's left leg is 3 cm longer, its left sole starts 3 cm inside the ground; one env-step in
    57 600 had the oracle's own ankle rate move by 0.3-0.7 rad/s, 15-35 x the tolerance, under a 1e-6 jitter, and the kernel arithmetic on
    the host differed from the oracle there by MORE than the GPU did)."""
    rng = np.random.default_rng(seed)
    bad = np.zeros(root0.shape[0], bool)
    for _ in range(2):
        r2 = (root0 * (1.0 + rel * rng.uniform(-1, 1, root0.shape))).astype(np.float32)
        d2 = (dof0 * (1.0 + rel * rng.uniform(-1, 1, dof0.shape))).astype(np.float32)
        oracle_lib.ta_simulate(cfg, m, act, r2, d2, threads=8)
        bad |= np.abs(r2[:, 2, 7:10] - root_after[:, 2, 7:10]).max(axis=1) > 1e-3
        if dof_after is not None:
            bad |= np.abs(d2[..., 1] - dof_after[..., 1]).max(axis=1) > TOL["qd"] / 3.0
    return bad


def check_step(got, want, what):
    (root_g, dof_g, rb_g, frc_g), (root_w, dof_w, rb_w, frc_w) = got, want
    assert_close(root_g[..., 0:3], root_w[..., 0:3], f"{what}: root pos", atol=TOL["root_pos"])
    sign = np.sign(np.sum(root_g[..., 3:7] * root_w[..., 3:7], axis=-1, keepdims=True))
    assert_close(root_g[..., 3:7] * sign, root_w[..., 3:7], f"{what}: root quat", atol=TOL["root_quat"])
    assert_close(root_g[..., 7:13], root_w[..., 7:13], f"{what}: root vel", atol=TOL["root_vel"])
    assert_close(dof_g[..., 0], dof_w[..., 0], f"{what}: dof pos", atol=TOL["q"])
    assert_close(dof_g[..., 1], dof_w[..., 1], f"{what}: dof vel", atol=TOL["qd"])
    assert_close(frc_g, frc_w, f"{what}: dof force", atol=TOL["frc"])
    assert_close(rb_g[..., 0:3], rb_w[..., 0:3], f"{what}: body pos", atol=TOL["rb_pos"])
    assert_close(rb_g[..., 7:10], rb_w[..., 7:10], f"{what}: body vel", atol=TOL["rb_vel"])
    assert_close(rb_g[..., 10:13], rb_w[..., 10:13], f"{what}: body ang vel", atol=TOL["rb_ang"])


def test_model_tables():
    m = scene.build_ta_model()
    assert abs(sum(m.link[i].mass for i in range(28)) - 35.7) < 0.2           # a G1 weighs about 35 kg
    assert [m.link[i].parent for i in range(28)] == [-1, 0, 1, 2, 3, 4, 5, 0, 7, 8, 9, 10, 11, 0, 13, 14, 15, 16, 17, 18, 19, 20, 21, 15, 23, 24, 25, 26]
    kp = [m.link[i].kp for i in range(1, 28)]
    assert kp == scene.TA_P_GAINS and kp[10] == 80.0 and kp[4] == 20.0        # TA:757-772, the asymmetric ankle gain as written
    bodies = sorted([m.link[i].body for i in range(28)] + [m.fixed[k].body for k in range(12)])
    assert bodies == list(range(40))                                           # every rigid body of pingpong_note.txt:33 exactly once


def test_standing_pose_and_kinematics(oracle_lib):
    """All dofs at 0: the soles are on the plane z = 0.21 (TA:403) with the pelvis where the task creates it (TA:578)."""
    m = scene.build_ta_model()
    root, dof = initial_tensors(4)
    rb = oracle_lib.ta_forward_kinematics(m, root, dof)
    sole_z = rb[0, [7, 14], 2] - 0.035
    assert np.all(np.abs(sole_z - scene.TA_GROUND_Z) < 0.005)
    np.testing.assert_allclose(rb[0, 7, 1], -rb[0, 14, 1], atol=1e-6)          # mirrored legs
    np.testing.assert_allclose(rb[0, 28, [0, 2]], rb[0, 38, [0, 2]], atol=1e-5)   # mirrored hands (welded bodies 28 / 38)
    np.testing.assert_allclose(rb[0, 28, 1], -rb[0, 38, 1], atol=1e-5)
    np.testing.assert_allclose(rb[:, 40, :3], root[:, 1, :3])                  # table and ball rows are the actor roots
    np.testing.assert_allclose(rb[:, 41], root[:, 2])


def test_free_flight_conserves_momentum(oracle_lib):
    """No gravity, no ground, drives off: total linear momentum is conserved up to the integrator's O(h^2) error."""
    n = 1
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    cfg.gravity_z = 0.0
    m.ground_z = cfg.ground_z = -100.0
    for i in range(1, 28):
        L = m.link[i]
        L.kp = L.kd = 0.0
        L.lower, L.upper, L.vel_limit = -1e3, 1e3, 1e6
    root, dof = initial_tensors(n)
    rng = np.random.default_rng(0)
    dof[0, :, 0], dof[0, :, 1] = rng.uniform(-0.3, 0.3, 27), rng.uniform(-1, 1, 27)
    root[0, 0, 7:13] = [0.3, -0.2, 0.1, 0.5, 1.0, -0.7]

    def momentum(rb):
        P = np.zeros(3)
        for i in range(28):
            L = m.link[i]
            r = rb[0, L.body].astype(np.float64)
            x, y, z, w = r[3:7]
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            P += L.mass * (r[7:10] + np.cross(r[10:13], R @ np.array(list(L.com))))
        return P
    p0 = momentum(oracle_lib.ta_forward_kinematics(m, root, dof))
    act = np.zeros((n, 27), np.float32)
    for _ in range(20):
        rb, _, _ = oracle_lib.ta_simulate(cfg, m, act, root, dof)
    assert np.abs(momentum(rb) - p0).max() < 0.01 * np.abs(p0).max()


def test_kernel_arithmetic_matches_oracle_single_steps(oracle_lib):
    """ppenv_ta_device.h compiled for the host (floating-base ABA, fp32) vs the oracle (Newton-Euler + dense solve, fp64),
    both restarted from the oracle's state every step; standing, falling, lying and thrashing states all occur."""
    n = 96
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    root, dof = initial_tensors(n, seed=1)
    rng = np.random.default_rng(2)
    act = np.zeros((n, 27), np.float32)
    ball_flips, contacts = 0, 0
    for t in range(140):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1                                           # a third of the envs moves gently
        r2, d2 = root.copy(), dof.copy()
        root0, dof0 = root.copy(), dof.copy()
        rb, frc, pvx = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=8)
        rb2, frc2, pvx2 = sb.ta_simulate(cfg, m, act, r2, d2)
        contacts += int((rb[:, :40, 2].min(axis=1) < scene.TA_GROUND_Z + 0.05).sum())
        np.testing.assert_array_equal(pvx2, pvx)
        hit = ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root, seed=t)   # the ball's discrete contact decisions, by the oracle's own sensitivity
        ball_flips += int(hit.sum())
        check_step((r2[~hit], d2[~hit], rb2[~hit], frc2[~hit]), (root[~hit], dof[~hit], rb[~hit], frc[~hit]), f"step {t}")
    assert ball_flips <= 3 and contacts > n * 60
    assert np.isfinite(root).all() and np.isfinite(dof).all()


def test_long_run_stays_physical(oracle_lib):
    """Random bang-bang actions for 3 episodes' worth of steps: nothing blows up, nothing sinks through the ground."""
    n = 16
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    root, dof = initial_tensors(n, seed=3)
    rng = np.random.default_rng(4)
    for t in range(480):
        if t % 5 == 0:
            act = rng.uniform(-1, 1, (n, 27)).astype(np.float32)
        rb, frc, _ = sb.ta_simulate(cfg, m, act, root, dof)
    assert np.isfinite(root).all() and np.isfinite(dof).all()
    assert np.abs(root[:, 0, 7:10]).max() < 8.0 and np.abs(dof[..., 1]).max() < 80.0
    assert rb[:, :40, 2].min() > scene.TA_GROUND_Z - 0.15 and root[:, 0, 2].max() < 1.6
    effort = np.array([m.link[i].effort for i in range(1, 28)])
    assert (np.abs(frc) <= effort * (1 + 1e-6)).all()


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mapping,n", [("quad", 512), ("quad", 50), ("lane", 200)])
def test_ta_simulate_kernel_matches_oracle(oracle_lib, monkeypatch, mapping, n):
    """Both mappings of the rigid-body step — four lanes per env (one per limb, hub merges through quad shuffles) and one
    lane per env (any tree) — against the oracle; n = 50 / 200 leave ragged last workgroups."""
    import torch
    from isaacgym_amd.tensor_api import TASim
    monkeypatch.setenv("PPENV_TA_KERNEL", mapping)
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    sim = TASim(n, device="cuda:0")
    root, dof = initial_tensors(n, seed=1)
    rng = np.random.default_rng(2)
    dev = lambda a: torch.from_numpy(a).cuda()
    rb_d, frc_d, pvx_d = torch.zeros(n, 42, 13, device="cuda"), torch.zeros(n, 27, device="cuda"), torch.zeros(n, device="cuda")
    # forward kinematics entry (initial_body_states of TA:1152)
    sim.forward_kinematics(dev(root), dev(dof), rb_d)
    assert_close(rb_d.cpu().numpy()[..., :3], oracle_lib.ta_forward_kinematics(m, root, dof)[..., :3], "fk pos", atol=1e-5)
    log = ExclusionLog(f"gpu 27-dof rigid-body step vs oracle [{mapping}, n={n}]", bound=0.005)
    for t in range(120):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1
        root_d, dof_d = dev(root), dev(dof)
        sim.simulate(dev(act), root_d, dof_d, rb_d, frc_d, pvx_d)
        root0, dof0 = root.copy(), dof.copy()
        rb, frc, pvx = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=8)
        np.testing.assert_array_equal(pvx_d.cpu().numpy(), pvx)
        rg = root_d.cpu().numpy()
        keep = ~ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root, seed=100 + t)   # the oracle's own sensitivity, not the GPU's error
        log.add(keep)
        got = (rg[keep], dof_d.cpu().numpy()[keep], rb_d.cpu().numpy()[keep], frc_d.cpu().numpy()[keep])
        check_step(got, (root[keep], dof[keep], rb[keep], frc[keep]), f"step {t}")
    log.close()
    sim.close()


@pytest.mark.gpu
def test_ta_env_steps_end_to_end():
    """TAEnv = ppenv_ta_simulate + ppenv_ta_post_physics_step: the 27-DoF task's VecTask step, BASELINE config 5's per-GPU size."""
    import torch
    from isaacgym_amd.tensor_api import TAEnv
    n = 4096
    envs = [TAEnv(n, device="cuda:0", seed=5) for _ in range(2)]
    assert envs[0].obs_buf.shape == (n, 313)
    gen = torch.Generator(device="cuda").manual_seed(0)
    resets = 0
    for t in range(400):
        a = torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1
        for e in envs:
            e.step(a)
        resets += int(envs[0].reset_buf.sum())
    torch.cuda.synchronize()
    a_, b_ = envs
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "root_states", "dof_states"):
        assert torch.equal(getattr(a_, name), getattr(b_, name)), name      # deterministic
    assert torch.isfinite(a_.obs_buf).all() and torch.isfinite(a_.rew_buf).all() and torch.isfinite(a_.root_states).all()
    assert resets == 2 * n                                                 # no early stop: every env times out at 160 steps (TA:1688)
    assert float(a_.rb_states[:, :40, 2].min()) > scene.TA_GROUND_Z - 0.2
    for e in envs:
        e.close()


@pytest.mark.gpu
def test_ta_vec_task_surface():
    """`isaacgym_amd.make("HumanoidPingpongTiltNESSparse27DOFG1")`: the 27-dof task on the VecTask surface."""
    import torch
    import isaacgym_amd
    n = 128
    task = isaacgym_amd.make(task="HumanoidPingpongTiltNESSparse27DOFG1", num_envs=n, seed=2)
    assert task.num_obs == 313 and task.num_actions == 27 and task.get_number_of_agents() == 1
    assert (task.actors_per_env, task.dofs_per_env, task.rigid_bodies_per_env) == (3, 27, 42)          # TA:156-158
    assert task.reset()["obs"].shape == (n, 313)
    for _ in range(3):
        od, rew, done, extras = task.step(torch.rand(n, 27, device="cuda") * 2 - 1)
    assert od["obs"].shape == (n, 313) and rew.shape == (n,) and done.dtype == torch.int64 and "time_outs" in extras
    assert task.body_states.shape == (n, 42, 13) and task.dof_pos.shape == (n, 27)
    assert task.env.params.hit_table_reward == 3000.0 and task.env.params.max_episode_length == 160    # 27DOFG1.yaml:10,21
    assert torch.isfinite(od["obs"]).all()


@pytest.mark.gpu
def test_ta_fused_step_equals_the_two_launches(monkeypatch):
    """ppenv_ta_step with the four-lanes-per-env kernel (task arithmetic on the rigid-body kernel's LDS tiles) vs ppenv_ta_simulate +
    ppenv_ta_post_physics_step (pinned to the reference's post_physics_step by post_physics_TA.npz): the same arithmetic, so the same
    tensors after every step of a free-running rollout, through resets."""
    import torch
    from isaacgym_amd.tensor_api import TAEnv
    monkeypatch.setenv("PPENV_TA_KERNEL", "quad")
    n = 1000   # ragged last workgroup
    a_, b_ = TAEnv(n, device="cuda:0", seed=3, fused=True), TAEnv(n, device="cuda:0", seed=3, fused=False)
    assert a_.sim.kernel == "quad"
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(170):     # past the 160-step time-out: every env resets once
        a = torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1
        a_.step(a)
        b_.step(a)
        for name in ("reset_buf", "progress_buf"):
            assert torch.equal(getattr(a_, name), getattr(b_, name)), (name, t)
        assert torch.equal(a_.state.flags, b_.state.flags) and torch.equal(a_.state.episode, b_.state.episode), t
        for name in ("obs_buf", "rew_buf", "root_states", "dof_states", "rb_states", "dof_force_tensor", "pre_ball_vx"):
            x, y = getattr(a_, name), getattr(b_, name)
            assert torch.allclose(x, y, rtol=1e-6, atol=1e-6), (name, t, float((x - y).abs().max()))
    assert int(a_.state.episode.sum()) == n
    a_.close(); b_.close()


def _ta_obs_atol():
    """Per-column atol of the 313-wide row: body pos / vel (TA:1849-1888), dofs, ball (+ y intercept), imitation blocks (TA:1891-1927)."""
    a = np.empty(scene.TA_NUM_OBS, np.float64)
    a[0:30] = TOL["rb_pos"]
    a[30:60] = TOL["rb_vel"]
    a[60:87] = TOL["q"]
    a[87:114] = TOL["qd"] * 0.1
    a[114:117] = 3e-4
    a[117:120] = 2e-3
    a[120] = 5e-3                      # y + (vy / (-vx + 1e-6)) x: a quotient of ball velocities
    a[121:190] = 10 * TOL["rb_pos"]    # 10 x position differences
    a[190:259] = TOL["rb_vel"]
    a[259:313] = 0.0                   # constants
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("n", [640, 1000, 50])
def test_ta_chain_kernel_step_matches_oracle(oracle_lib, monkeypatch, n):
    """The chain-wave kernel (one lane per env, one wave per limb: ppenv_ta_chain.hip) through ppenv_ta_step, against the oracle's
    rigid-body step followed by its post_physics_step, restarted from the oracle's tensors every step.  n = 1000 / 50 leave a
    ragged last workgroup; episodeLength 40 makes every env reset (with the keyed draws) inside the run."""
    from isaacgym_amd.tensor_api import TAEnv
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    env = TAEnv(n, device="cuda:0", seed=11, env={"episodeLength": 40}, materialize_rb=True, share_initial_rb=(n != 50))   # n = 50: the reference's per-env [N,42,13]
    assert env.sim.kernel == "chain" and env.initial_rb_states.shape[0] == (n if n == 50 else 1)
    run_chain_step_parity(oracle_lib, env, cfg, m, f"gpu 27-dof chain-wave step vs oracle [n={n}]")
    env.close()


@pytest.mark.gpu
def test_ta_chain_kernel_step_matches_oracle_at_config_5_size(oracle_lib, monkeypatch):
    """The same comparison at BASELINE config 5's per-GPU size as named, 4096 envs = 64 full workgroups (the runs above are 640 / 1000 / 50):
    fewer steps (the oracle and its jitter probe cost ~0.3 s per step here), episodeLength 10 so that every env still resets twice inside them."""
    from isaacgym_amd.tensor_api import TAEnv
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")
    n = 4096
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    env = TAEnv(n, device="cuda:0", seed=12, env={"episodeLength": 10}, materialize_rb=True)
    assert env.sim.kernel == "chain"
    run_chain_step_parity(oracle_lib, env, cfg, m, f"gpu 27-dof chain-wave step vs oracle [n={n}, config 5's size]", steps=24, min_resets_per_env=2)
    env.close()


@pytest.mark.gpu
def test_ta_gravity_setter_reaches_every_kernel(oracle_lib, monkeypatch):
    """ppenv_ta_sim_set_gravity (the yaml's randomization_params.sim_params.gravity for this task, 27DOFG1.yaml:123-124): the chain-wave
    step — links through the by-value constants, ball through StepConsts in device memory — and the table-driven kernels follow the oracle
    under the new gravity; the VecTask hook draws and applies it (round 3 skipped the key in silence)."""
    import torch
    import isaacgym_amd
    from isaacgym_amd.tensor_api import TAEnv, TASim
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")
    n, gz = 320, -6.5
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    cfg.gravity_z = gz
    env = TAEnv(n, device="cuda:0", seed=17, env={"episodeLength": 40}, materialize_rb=True)
    env.set_gravity(gz)
    run_chain_step_parity(oracle_lib, env, cfg, m, f"gpu 27-dof chain-wave step under gravity {gz} vs oracle [n={n}]", steps=50, min_resets_per_env=1)
    env.close()
    monkeypatch.setenv("PPENV_TA_KERNEL", "quad")
    sim = TASim(64, device="cuda:0")
    sim.set_gravity(gz)
    cfg64 = scene.build_ta_scene(64)
    cfg64.gravity_z = gz
    root, dof = initial_tensors(64, seed=3)
    act = np.random.default_rng(4).uniform(-1, 1, (64, 27)).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    root_d, dof_d = dev(root), dev(dof)
    rb_d, frc_d, pvx_d = torch.zeros(64, 42, 13, device="cuda"), torch.zeros(64, 27, device="cuda"), torch.zeros(64, device="cuda")
    sim.simulate(dev(act), root_d, dof_d, rb_d, frc_d, pvx_d)
    rb, frc, _ = oracle_lib.ta_simulate(cfg64, m, act, root, dof, threads=8)
    check_step((root_d.cpu().numpy(), dof_d.cpu().numpy(), rb_d.cpu().numpy(), frc_d.cpu().numpy()), (root, dof, rb, frc), "quad kernel under the new gravity")
    assert abs(float(root_d[0, 2, 9]) - (1.2 + 0.0)) > 0 and abs((root[0, 2, 9] - initial_tensors(64, seed=3)[0][0, 2, 9]) - gz * 0.0083) < 2e-3   # the ball fell by gz dt
    sim.close()
    with pytest.raises(isaacgym_amd._lib.PPEnvError, match="<= 0"):
        TASim(8, device="cuda:0").set_gravity(1.0)
    monkeypatch.delenv("PPENV_TA_KERNEL")
    task = isaacgym_amd.make(task="HumanoidPingpongTiltNESSparse27DOFG1", num_envs=64, seed=2)
    task.randomize, task.first_randomization, task.last_step, task.last_rand_step = True, True, 0, -1
    task.apply_randomizations({"frequency": 1, "sim_params": {"gravity": {"range": [0.0, 0.4], "operation": "additive", "distribution": "gaussian"}}})
    assert task.env.sim.scene.gravity_z != np.float32(scene.TA_GRAVITY_Z) and task.env.sim.scene.gravity_z <= 0.0


def run_chain_step_parity(oracle_lib, env, cfg, m, label, steps=90, min_resets_per_env=2, joint_probe=False):
    """A TAEnv on the chain-wave kernel against the oracle's rigid-body step + post_physics_step with the same scene `cfg` and tree `m`,
    restarted from the oracle's tensors every step (also used by tests/test_urdf.py for a library built for another asset)."""
    import torch
    n = env.num_envs
    p = env.params
    root, dof = env.root_states.cpu().numpy().copy(), env.dof_states.cpu().numpy().copy()
    irb = env.initial_rb_states.cpu().numpy().copy()
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    rng = np.random.default_rng(5)
    oa = _ta_obs_atol()
    log = ExclusionLog(label, bound=0.005)
    resets = 0
    act = None
    for t in range(steps):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1
        env.root_states.copy_(torch.from_numpy(root)); env.dof_states.copy_(torch.from_numpy(dof))
        env.state.flags.copy_(torch.from_numpy(flags.view(np.int32))); env.state.episode.copy_(torch.from_numpy(episode.view(np.int32)))
        env.state.progress_buf.copy_(torch.from_numpy(progress))
        env.step(torch.from_numpy(act).cuda())
        root0, dof0 = root.copy(), dof.copy()
        rb, frc, pvx = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=8)         # root / dof: stepped in place (pre-reset)
        switch = ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root, seed=200 + t, dof_after=dof if joint_probe else None)
        obs, rew, reset = oracle_lib.ta_post_physics_step(p, rb, irb, root, dof, frc, pvx, None, flags, episode, progress)
        g_rb = env._rb_states.cpu().numpy()
        np.testing.assert_array_equal(env.pre_ball_vx.cpu().numpy(), pvx)
        # the ball's contact decisions are discrete: envs whose ORACLE result moves under a 1e-6 jitter are set aside for this step
        keep = ~switch
        log.add(keep)
        # pre-reset physics through the materialised rigid_body_states; post-reset tensors against the oracle's
        assert_close(g_rb[keep][..., 0:3], rb[keep][..., 0:3], f"step {t}: body pos", atol=TOL["rb_pos"])
        assert_close(g_rb[keep][..., 7:10], rb[keep][..., 7:10], f"step {t}: body vel", atol=TOL["rb_vel"])
        assert_close(g_rb[keep][..., 10:13], rb[keep][..., 10:13], f"step {t}: body ang vel", atol=TOL["rb_ang"])
        sign = np.sign(np.sum(g_rb[..., 3:7] * rb[..., 3:7], axis=-1, keepdims=True))
        assert_close((g_rb[..., 3:7] * sign)[keep], rb[keep][..., 3:7], f"step {t}: body quat", atol=2e-4)
        assert_close(env.dof_force_tensor.cpu().numpy()[keep], frc[keep], f"step {t}: dof force", atol=TOL["frc"])
        g_root, g_dof = env.root_states.cpu().numpy(), env.dof_states.cpu().numpy()
        np.testing.assert_array_equal(env.reset_buf.cpu().numpy(), reset)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy(), progress)
        np.testing.assert_array_equal(env.state.episode.cpu().numpy().view(np.uint32), episode)
        np.testing.assert_array_equal(env.state.flags.cpu().numpy().view(np.uint32)[keep], flags[keep])
        check_step((g_root[keep], g_dof[keep], g_rb[keep][:, :40], env.dof_force_tensor.cpu().numpy()[keep]),
                   (root[keep], dof[keep], rb[keep][:, :40], frc[keep]), f"step {t}")
        g_obs = env.obs_buf.cpu().numpy()
        assert_close(np.delete(g_obs, 120, axis=1)[keep], np.delete(obs, 120, axis=1)[keep], f"step {t}: obs", atol=np.delete(oa, 120))
        # column 120 = y + (vy / (-vx + 1e-6)) x (TA:1839) is a quotient: its tolerance is the ball-velocity tolerance times its sensitivity
        lbx, lvx, lvy = obs[:, 114].astype(np.float64), obs[:, 117].astype(np.float64), obs[:, 118].astype(np.float64)
        den = np.maximum(np.abs(-lvx + 1e-6), 1e-9)
        tol120 = oa[120] + (oa[117] + 1e-4 * np.abs(lvx)) * np.abs(lbx * lvy) / den ** 2 + (oa[118] + 1e-4 * np.abs(lvy)) * np.abs(lbx) / den + 1e-4 * np.abs(obs[:, 120])
        bad = (np.abs(g_obs[:, 120].astype(np.float64) - obs[:, 120]) > tol120) & keep
        assert not bad.any(), (t, np.nonzero(bad)[0][:5], g_obs[bad, 120][:5], obs[bad, 120][:5])
        assert_close(env.rew_buf.cpu().numpy()[keep], rew[keep], f"step {t}: rew", atol=1e-4 * 3000.0 * 0.5)   # alpha |vx| dominates (TA:1590)
        resets += int(reset.sum())
        flags[~keep] = env.state.flags.cpu().numpy().view(np.uint32)[~keep]                 # continue from a common state
    assert resets >= min_resets_per_env * n
    log.close()
    assert env.sim.status == 0


@pytest.mark.gpu
def test_ta_chain_kernel_clears_count_flags_across_workgroups_and_skips_rb(monkeypatch):
    """TA:1162-1166 inside the one launch: when any env resets, the workgroup that finishes last clears every env's count flags
    and leaves the scratch word zero.  Without a rigid_body_states buffer the step stores none, and `rb_states` is forward
    kinematics on demand."""
    import torch
    from isaacgym_amd.tensor_api import TAEnv
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")
    n = 1500
    env = TAEnv(n, device="cuda:0", seed=2, env={"episodeLength": 30})
    assert env.sim.kernel == "chain" and not env.materialize_rb
    env.state.flags.fill_(scene.TA_COUNT_MASK)            # every env carries count flags
    env.state.progress_buf[:] = 0
    env.state.progress_buf[n - 1] = 28                    # ... and one env of the LAST workgroup times out in this step
    sentinel = env._rb_states.clone().fill_(123.0)
    env._rb_states.copy_(sentinel)
    env.step(torch.zeros(n, 27, device="cuda"))
    torch.cuda.synchronize()
    assert int(env.reset_buf.sum()) == 1 and int(env.reset_buf[n - 1]) == 1
    assert int((env.state.flags & scene.TA_COUNT_MASK).abs().sum()) == 0
    assert int(env.state._any_reset.item()) == 0          # the ticket word is zero again
    assert torch.equal(env._rb_states, sentinel)          # not materialised
    rb = env.rb_states                                    # on demand
    assert float(rb[:, :40, 2].min()) > scene.TA_GROUND_Z - 0.2 and not torch.equal(rb, sentinel)
    env.step(torch.zeros(n, 27, device="cuda"))           # nobody resets: count flags may come back, the word stays zero
    torch.cuda.synchronize()
    assert int(env.reset_buf.sum()) == 0 and int(env.state._any_reset.item()) == 0
    env.close()


# ------------------------------------------------------------------------------------------- domain randomisation (round 3)
def _ta_dr_tables(n, rng):
    u = lambda lo, hi, shape: rng.uniform(lo, hi, shape).astype(np.float32)
    return dict(dof_stiffness_scale=u(0.6, 1.4, (27, n)), dof_damping_scale=u(0.6, 1.4, (27, n)), link_mass_scale=u(0.7, 1.3, (28, n)),
                restitution_scale=u(0.0, 0.7, n), friction_scale=u(0.7, 1.3, n))


def test_ta_oracle_randomisation_with_unit_tables_is_the_plain_step(oracle_lib):
    n = 24
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    root, dof = initial_tensors(n, seed=4)
    rng = np.random.default_rng(1)
    act = rng.uniform(-1, 1, (n, 27)).astype(np.float32)
    r2, d2 = root.copy(), dof.copy()
    rb, frc, _ = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=4)
    ones = {k: np.ones_like(v) for k, v in _ta_dr_tables(n, rng).items()}
    rb2, frc2, _ = oracle_lib.ta_simulate_dr(cfg, m, act, r2, d2, np.zeros(n, np.uint32), np.zeros(n, np.int64), seed=5, threads=4, **ones)
    np.testing.assert_array_equal(rb, rb2)
    np.testing.assert_array_equal(frc, frc2)
    # ... and real tables / action noise move it
    r3, d3 = root.copy(), dof.copy()
    oracle_lib.ta_simulate_dr(cfg, m, act, r3, d3, np.zeros(n, np.uint32), np.zeros(n, np.int64), seed=5, threads=4, action_noise_sigma=0.05, **_ta_dr_tables(n, rng))
    r4, d4 = root.copy(), dof.copy()
    oracle_lib.ta_simulate(cfg, m, act, r4, d4, threads=4)
    assert np.abs(d3 - d4).max() > 1e-3
    obs = np.zeros((n, 313), np.float32)
    oracle_lib.ta_add_obs_noise(obs, 0.01, 5, np.zeros(n, np.uint32), np.arange(n, dtype=np.int64))
    assert 0.008 < obs.std() < 0.012 and abs(obs.mean()) < 1e-3
    assert len(np.unique(obs)) > 0.99 * obs.size                     # no index collisions inside a row or between consecutive steps
    both = np.concatenate([obs[3], obs[4]])                          # progress 3 and 4 of (the same seed, different env ids)
    assert len(np.unique(both)) == both.size


@pytest.mark.gpu
def test_ta_chain_kernel_with_randomisation_matches_oracle(oracle_lib, monkeypatch):
    """The table-reading instantiation of the chain-wave kernel (per-env drive gains, link masses, materials, action and observation noise)
    against the oracle with the same tables (every env its own scaled copy of the model), restarted from the oracle's tensors every step; with
    the randomisation cleared the plain kernel runs again, bit for bit."""
    import torch
    from isaacgym_amd.tensor_api import TAEnv
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")
    n = 640
    cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
    env = TAEnv(n, device="cuda:0", seed=11, env={"episodeLength": 40}, materialize_rb=True)
    ref = TAEnv(n, device="cuda:0", seed=11, env={"episodeLength": 40}, materialize_rb=True)
    p = env.params
    rng = np.random.default_rng(21)
    tabs = _ta_dr_tables(n, rng)
    kw = dict(action_noise_sigma=0.02, observation_noise_sigma=0.002)
    env.set_randomization(**tabs, **kw)
    root, dof = env.root_states.cpu().numpy().copy(), env.dof_states.cpu().numpy().copy()
    irb = np.broadcast_to(env.initial_rb_states.cpu().numpy(), (1, 42, 13)).copy()
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    oa = _ta_obs_atol() + 2e-6
    log = ExclusionLog(f"gpu 27-dof chain-wave step with domain randomisation vs oracle [n={n}]", bound=0.005)
    resets, moved = 0, 0.0
    act = None

    def load(e):
        e.root_states.copy_(torch.from_numpy(root)); e.dof_states.copy_(torch.from_numpy(dof))
        e.state.flags.copy_(torch.from_numpy(flags.view(np.int32))); e.state.episode.copy_(torch.from_numpy(episode.view(np.int32)))
        e.state.progress_buf.copy_(torch.from_numpy(progress))
    for t in range(70):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1
        load(env)
        env.step(torch.from_numpy(act).cuda())
        if t == 10:                                                   # the plain kernel from the same state: the tables do something
            load(ref)
            ref.step(torch.from_numpy(act).cuda())
            moved = float((ref.dof_states - env.dof_states).abs().max())
        root0, dof0, ep0, prog0 = root.copy(), dof.copy(), episode.copy(), progress.copy()
        rb, frc, pvx = oracle_lib.ta_simulate_dr(cfg, m, act, root, dof, ep0, prog0, seed=p.seed, env_id_offset=p.env_id_offset, threads=8,
                                                 action_noise_sigma=kw["action_noise_sigma"], **tabs)
        # the ball's discrete contact decisions, by the ORACLE's own sensitivity (with this env's tables: a jittered second oracle step)
        rng2 = np.random.default_rng(500 + t)
        switch = np.zeros(n, bool)
        for _ in range(2):
            rj = (root0 * (1.0 + 1e-6 * rng2.uniform(-1, 1, root0.shape))).astype(np.float32)
            dj = (dof0 * (1.0 + 1e-6 * rng2.uniform(-1, 1, dof0.shape))).astype(np.float32)
            oracle_lib.ta_simulate_dr(cfg, m, act, rj, dj, ep0, prog0, seed=p.seed, env_id_offset=p.env_id_offset, threads=8,
                                      action_noise_sigma=kw["action_noise_sigma"], **tabs)
            switch |= np.abs(rj[:, 2, 7:10] - root[:, 2, 7:10]).max(axis=1) > 1e-3
        obs, rew, reset = oracle_lib.ta_post_physics_step(p, rb, irb, root, dof, frc, pvx, None, flags, episode, progress)
        oracle_lib.ta_add_obs_noise(obs, kw["observation_noise_sigma"], p.seed, ep0, prog0, env_id_offset=p.env_id_offset)
        keep = ~switch
        log.add(keep)
        g_rb = env._rb_states.cpu().numpy()
        g_root, g_dof, g_frc = env.root_states.cpu().numpy(), env.dof_states.cpu().numpy(), env.dof_force_tensor.cpu().numpy()
        np.testing.assert_array_equal(env.reset_buf.cpu().numpy(), reset)
        np.testing.assert_array_equal(env.progress_buf.cpu().numpy(), progress)
        np.testing.assert_array_equal(env.state.flags.cpu().numpy().view(np.uint32)[keep], flags[keep])
        check_step((g_root[keep], g_dof[keep], g_rb[keep][:, :40], g_frc[keep]), (root[keep], dof[keep], rb[keep][:, :40], frc[keep]), f"DR step {t}")
        g_obs = env.obs_buf.cpu().numpy()
        assert_close(np.delete(g_obs, 120, axis=1)[keep], np.delete(obs, 120, axis=1)[keep], f"DR step {t}: obs", atol=np.delete(oa, 120))
        assert_close(env.rew_buf.cpu().numpy()[keep], rew[keep], f"DR step {t}: rew", atol=1e-4 * 3000.0 * 0.5)
        resets += int(reset.sum())
        flags[~keep] = env.state.flags.cpu().numpy().view(np.uint32)[~keep]
    log.close()
    assert resets >= n and moved > 1e-2
    # cleared: the plain kernel again, bit for bit
    env.clear_randomization()
    load(env); load(ref)
    a = torch.from_numpy(act).cuda()
    env.step(a); ref.step(a)
    for name in ("root_states", "dof_states", "dof_force_tensor", "obs_buf", "rew_buf"):
        assert torch.equal(getattr(env, name), getattr(ref, name)), name
    assert env.sim.status == 0
    env.close(); ref.close()


@pytest.mark.gpu
def test_ta_vec_task_randomize_true_draws_tables_of_the_tree(monkeypatch):
    """task.randomize = True on the 27-dof VecTask: apply_randomizations fills tables shaped for the tree ([27, N] gains, [28, N] masses) and the
    steps stay finite; a table-driven kernel refuses the randomisation."""
    import torch
    from isaacgym_amd import _lib
    from isaacgym_amd.tasks import isaacgym_task_map
    from isaacgym_amd.tensor_api import TASim
    cfg = scene.default_task_cfg("TA")
    cfg["env"]["numEnvs"] = 256
    cfg["task"] = {"randomize": True, "randomization_params": {
        "frequency": 8, "observations": {"range": [0, 0.002], "operation": "additive", "distribution": "gaussian"},
        "actions": {"range": [0.0, 0.02], "operation": "additive", "distribution": "gaussian"},
        "actor_params": {"humanoid": {"rigid_body_properties": {"mass": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform"}},
                                      "rigid_shape_properties": {"restitution": {"range": [0.0, 0.7], "operation": "scaling", "distribution": "uniform"}},
                                      "dof_properties": {"stiffness": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform"},
                                                         "damping": {"range": [0.8, 1.2], "operation": "scaling", "distribution": "uniform"}}}}}}
    task = isaacgym_task_map["HumanoidPingpongTiltNESSparse27DOFG1"](cfg, "cuda:0", "cuda:0", -1, True, False, False)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(20):
        obs, rew, done, _ = task.step(torch.rand(256, 27, device="cuda", generator=gen) * 2 - 1)
    dr = task.env.sim._dr
    assert tuple(dr[0].shape) == (27, 256) and tuple(dr[2].shape) == (28, 256) and dr[3].shape == (256,) and dr[4] is None
    assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all() and task.env.sim.status == 0
    monkeypatch.setenv("PPENV_TA_KERNEL", "quad")
    sim = TASim(64, device="cuda:0")
    with pytest.raises(_lib.PPEnvError, match="chain-wave"):
        sim.set_randomization(action_noise_sigma=0.1)
    sim.close()
