"""Parity tests proper: the HIP kernels, called through the C ABI, against (i) the reference's own
outputs (tests/golden) and (ii) the CPU oracle on the same seeded inputs.  Need a real MI355X."""
import numpy as np
import pytest

from helpers import (RTOL, ExclusionLog, SensitivityProbe, assert_close, assert_state_close, expand_bodies, golden_config, load_golden,
                     mask_envs, obs_atol, reward_atol)
from isaacgym_amd import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def make_env(config):
    from isaacgym_amd.env import PPEnv
    return PPEnv(config, device="cuda:0")


class DevView:
    """numpy snapshot of a PPEnv's state with the oracle's attribute names."""

    def __init__(self, env):
        self.dof_pos, self.dof_vel, self.dof_force = (t.cpu().numpy() for t in (env.dof_pos, env.dof_vel, env.dof_force))
        self.ball = env.ball.cpu().numpy()
        self.flags = env.flags.cpu().numpy().view(np.uint32)
        self.episode = env.episode.cpu().numpy().view(np.uint32)
        self.progress_buf = env.progress_buf.cpu().numpy()
        self.reset_buf = env.reset_buf.cpu().numpy()
        self.rew_buf = env.rew_buf.cpu().numpy()
        self.obs_buf = env.obs_buf.cpu().numpy()


@pytest.mark.parametrize("variant", ["TT", "TN", "T3"])
def test_post_physics_matches_reference_golden(torch_cuda, variant):
    """ppenv_post_physics_step (tensor-API mode) on the scripted sequences vs the reference's post_physics_step."""
    torch = torch_cuda
    g = load_golden(variant)
    env = make_env(golden_config(variant, g))
    T = g["out_rew"].shape[0]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for t in range(T):
        env.set_serve_override(np.nan_to_num(g["serve"][t], nan=0.0), on=True)
        rb, root, dof = dev(expand_bodies(g["in_bodies"][t])), dev(g["in_root"][t]), dev(g["in_dof"][t])
        env.post_physics_step(rb, root, dof, dev(g["in_dof_force"][t]), dev(g["in_pre_vx"][t]))
        v = DevView(env)
        np.testing.assert_array_equal(v.reset_buf, g["out_reset"][t], err_msg=f"reset_buf, step {t}")
        np.testing.assert_array_equal(v.progress_buf, g["out_progress"][t], err_msg=f"progress_buf, step {t}")
        np.testing.assert_array_equal(v.flags, g["out_flags"][t], err_msg=f"flags, step {t}")
        assert_close(v.rew_buf, g["out_rew"][t], f"rew_buf, step {t}")
        assert_close(v.obs_buf, g["out_obs"][t], f"obs_buf, step {t}")
        assert_close(root.cpu().numpy(), g["out_root"][t], f"root_states after reset, step {t}", rtol=0, atol=1e-7)
        assert_close(dof.cpu().numpy(), g["out_dof"][t], f"dof_states after reset, step {t}", rtol=0, atol=0)
    env.close()


@pytest.mark.parametrize("variant,n", [("TT", 1024), ("TN", 1024), ("T3", 1024), ("TT", 4096), ("T3", 4096), ("TT", 16384)])
def test_fused_step_matches_oracle_single_steps(torch_cuda, oracle_lib, variant, n):
    """The fused step kernel vs the oracle, both restarted from the oracle's state every step.  (TT / T3 at 4096 envs: BASELINE.json
    configs[1] as named — "3-actor, num_envs=4096 on 1 MI355X, fp32, random actions, obs/reward parity"; TT at 16 384 envs: configs[2], the bandwidth run's size, one full
    round of workgroups on the chip.)"""
    torch = torch_cuda
    cfg = scene.build_config(variant, num_envs=n, seed=7)
    o = oracle_lib.OracleEnv(cfg, threads=8)
    env = make_env(scene.build_config(variant, num_envs=n, seed=7))
    rng = np.random.default_rng(1)
    oa, ra = obs_atol(), reward_atol(cfg)
    steps = 180 if variant == "TN" else (120 if n <= 1024 else (90 if n <= 4096 else 60))
    resets = 0
    probe = SensitivityProbe(oracle_lib, cfg)
    log = ExclusionLog(f"gpu fused step vs oracle [{variant}, n={n}]", bound=0.005)
    for t in range(steps):
        actions = rng.uniform(-1.2, 1.2, (n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        o.step(actions)
        env.step(torch.from_numpy(actions).cuda())
        keep = ~probe.sensitive(st, actions, o)      # envs sitting on a switch of the physics spec this step are skipped
        log.add(keep)
        v, o_all = mask_envs(DevView(env), keep), o
        o = mask_envs(o_all, keep)
        np.testing.assert_array_equal(v.reset_buf, o.reset_buf, err_msg=f"reset step {t}")
        np.testing.assert_array_equal(v.progress_buf, o.progress_buf, err_msg=f"progress step {t}")
        np.testing.assert_array_equal(v.flags, o.flags, err_msg=f"flags step {t}")
        np.testing.assert_array_equal(v.episode, o.episode, err_msg=f"episode step {t}")
        assert_state_close(v, o, f"step {t}")
        assert_close(v.obs_buf, o.obs_buf, f"obs step {t}", atol=oa)
        assert_close(v.rew_buf, o.rew_buf, f"rew step {t}", atol=ra)
        o = o_all
        resets += int(o.reset_buf.sum())
    assert resets > (50 if n <= 4096 else 10)   # the masked-reset path was exercised (the full-size case runs fewer steps)
    log.close()          # prints excluded count + worst retained error, asserts the bound
    env.close()


@pytest.mark.parametrize("schedule,n", [("split", 1000), ("fused", 1000), ("fused", 130), ("split", 63), ("split", 1), ("fused", 1),
                                        ("quad", 1000), ("quad", 63), ("quad", 1),
                                        ("split:1", 1000), ("split:2", 1000), ("split:2", 95), ("split:2", 1), ("split:4", 1000), ("split:4", 81), ("split:4", 17)])
def test_both_schedules_on_ragged_sizes(torch_cuda, oracle_lib, monkeypatch, schedule, n):
    """The two-wave (default), the one-wave and the four-wave (round-2 experiment) step kernels run the same arithmetic;
    sizes that are not a multiple of 64.  "split:B": the two-wave schedule with B ball waves of 64 / B envs each (round 3)."""
    torch = torch_cuda
    if ":" in schedule:
        schedule, bw = schedule.split(":")
        monkeypatch.setenv("PPENV_BALL_WAVES", bw)
    monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)
    cfg = scene.build_config("TT", num_envs=n, seed=21)
    o = oracle_lib.OracleEnv(cfg)
    env = make_env(scene.build_config("TT", num_envs=n, seed=21))
    rng = np.random.default_rng(4)
    oa, ra = obs_atol(), reward_atol(cfg)
    for t in range(60):
        actions = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        env.set_state(o.get_state())
        o.step(actions)
        env.step(torch.from_numpy(actions).cuda())
        v = DevView(env)
        np.testing.assert_array_equal(v.reset_buf, o.reset_buf, err_msg=f"reset step {t}")
        np.testing.assert_array_equal(v.flags, o.flags, err_msg=f"flags step {t}")
        assert_state_close(v, o, f"{schedule} step {t}")
        assert_close(v.obs_buf, o.obs_buf, f"{schedule} obs step {t}", atol=oa)
        assert_close(v.rew_buf, o.rew_buf, f"{schedule} rew step {t}", atol=ra)
    env.close()


def test_schedules_agree_step_by_step(torch_cuda, monkeypatch):
    """Two compilations of the same per-env arithmetic (FMA contraction may differ in the last bits): restarted from a
    common state every step they must agree to a few ulp, and take identical discrete decisions."""
    torch = torch_cuda
    n = 4096
    envs = {}
    for schedule in ("split", "fused", "quad"):
        monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)
        envs[schedule] = make_env(scene.build_config("TN", num_envs=n, seed=2))
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(100):
        a = torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1
        envs["fused"].set_state(envs["split"].get_state())
        envs["quad"].set_state(envs["split"].get_state())
        for e in envs.values():
            e.step(a)
        for other in ("fused", "quad"):
            for name in ("reset_buf", "progress_buf", "flags", "episode"):
                assert torch.equal(getattr(envs["split"], name), getattr(envs[other], name)), (other, name, t)
            # rew: alpha = 1000 (TN) times the ball-velocity agreement of ~1e-4 m/s right after a paddle hit
            for name, atol in (("obs_buf", 2e-4), ("rew_buf", 1e-1), ("dof_pos", 1e-5), ("dof_vel", 1e-3), ("ball", 5e-3)):
                x, y = getattr(envs["split"], name), getattr(envs[other], name)
                assert torch.allclose(x, y, rtol=1e-5, atol=atol), (other, name, t, float((x - y).abs().max()))
    for e in envs.values():
        e.close()


def test_initial_state_and_reset_all_match_oracle(torch_cuda, oracle_lib):
    n = 777   # ragged: not a multiple of the 64-lane workgroup
    for variant in ("TT", "T3", "TN"):
        cfg = scene.build_config(variant, num_envs=n, seed=99, env_id_offset=12345)
        o = oracle_lib.OracleEnv(cfg)
        env = make_env(scene.build_config(variant, num_envs=n, seed=99, env_id_offset=12345))
        for phase in ("create", "reset_all"):
            if phase == "reset_all":
                o.reset_all()
                env.reset_all()
            v = DevView(env)
            assert_state_close(v, o, f"{variant} {phase}")
            assert_close(v.obs_buf, o.obs_buf, f"{variant} {phase} obs", atol=obs_atol())
            np.testing.assert_array_equal(v.flags, o.flags)
            np.testing.assert_array_equal(v.episode, o.episode)
            np.testing.assert_array_equal(v.progress_buf, o.progress_buf)
            np.testing.assert_array_equal(v.reset_buf, o.reset_buf)
        env.close()


def test_refresh_tensors_match_oracle(torch_cuda, oracle_lib):
    torch = torch_cuda
    n = 300
    cfg = scene.build_config("TT", num_envs=n, seed=5)
    o = oracle_lib.OracleEnv(cfg)
    env = make_env(scene.build_config("TT", num_envs=n, seed=5))
    rng = np.random.default_rng(3)
    for t in range(20):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        o.step(a)
    env.set_state(o.get_state())
    assert_close(env.refresh_root_states().cpu().numpy(), o.refresh_root_states(), "root_states", atol=1e-6)
    assert_close(env.refresh_dof_states().cpu().numpy(), o.refresh_dof_states(), "dof_states", atol=0, rtol=0)
    assert_close(env.refresh_dof_force().cpu().numpy(), o.refresh_dof_force(), "dof_force", atol=0, rtol=0)
    got, want = env.refresh_rigid_body_states().cpu().numpy(), o.refresh_rigid_body_states()
    sign = np.sign(np.sum(got[..., 3:7] * want[..., 3:7], axis=-1, keepdims=True))
    got[..., 3:7] *= np.where(sign == 0, 1, sign)
    assert_close(got[..., 0:7], want[..., 0:7], "rigid body poses", atol=2e-6)
    assert_close(got[..., 7:13], want[..., 7:13], "rigid body velocities", atol=RTOL * 20.0)
    env.close()


def test_state_blob_roundtrip_and_errors(torch_cuda):
    from isaacgym_amd import _lib
    n = 130
    env = make_env(scene.build_config("TT", num_envs=n, seed=1))
    blob = env.get_state()
    assert blob.size == n * ((7 * 3 + 13) * 4 + 4 + 4 + 8 + 8)
    env.step(torch_cuda.zeros(n, 7, device="cuda"))
    changed = env.get_state()
    assert not np.array_equal(blob, changed)
    env.set_state(blob)
    np.testing.assert_array_equal(env.get_state(), blob)
    with pytest.raises(_lib.PPEnvError):
        env.set_state(blob[:-8])
    env.close()
    bad = scene.build_config("TT", num_envs=n)
    bad.joint[2].axis = 0   # a chain this build has no kernel instantiation for
    with pytest.raises(_lib.PPEnvError, match="differs from the one compiled"):
        make_env(bad)


def test_full_size_determinism_sharding_and_invariants(torch_cuda):
    """BASELINE config 3 size (N=16384): size-independent properties of the fused step."""
    torch = torch_cuda
    n, steps = 16384, 160
    full = make_env(scene.build_config("TT", num_envs=n, seed=3))
    twin = make_env(scene.build_config("TT", num_envs=n, seed=3))
    half = [make_env(scene.build_config("TT", num_envs=n // 2, seed=3, env_id_offset=k * (n // 2))) for k in range(2)]
    gen = torch.Generator(device="cuda").manual_seed(0)
    total_resets = 0
    lo = torch.tensor([full.config.joint[j].lower for j in range(7)], device="cuda")[:, None]
    hi = torch.tensor([full.config.joint[j].upper for j in range(7)], device="cuda")[:, None]
    for t in range(steps):
        a = torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1
        full.step(a)
        twin.step(a)
        half[0].step(a[: n // 2].contiguous())
        half[1].step(a[n // 2:].contiguous())
        total_resets += int(full.reset_buf.sum())
        # reset => progress restarted; otherwise progress advanced
        assert bool(((full.reset_buf == 1) == (full.progress_buf == 0)).all())
    # determinism: same seed, same actions -> bit-identical
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "dof_pos", "dof_vel", "ball", "flags", "episode"):
        assert torch.equal(getattr(full, name), getattr(twin, name)), name
    # shard invariance: env i's trajectory does not depend on how envs are split over handles
    assert torch.equal(full.obs_buf, torch.cat([half[0].obs_buf, half[1].obs_buf]))
    assert torch.equal(full.ball, torch.cat([half[0].ball, half[1].ball], dim=1))
    assert torch.equal(full.episode, torch.cat([half[0].episode, half[1].episode]))
    # physical invariants
    assert bool(torch.isfinite(full.obs_buf).all()) and bool(torch.isfinite(full.ball).all())
    assert bool((full.dof_pos >= lo - 1e-6).all()) and bool((full.dof_pos <= hi + 1e-6).all())
    assert float((full.ball[3:7].pow(2).sum(0) - 1).abs().max()) < 1e-4
    assert float(full.ball[2].min()) > -0.05          # never tunnels through the ground
    assert total_resets > n                           # every env finished at least one episode on average
    for e in [full, twin] + half:
        e.close()


def test_largest_size_is_the_same_env_by_env(torch_cuda):
    """Maximum size (the top of tools/gpu_sizes.sh's sweep: 131 072 envs, 8 x BASELINE config 3, two rounds of workgroups per CU): the first
    16 384 envs of the big handle go through the same trajectory, bit for bit, as a 16 384-env handle — an env's result depends on its global
    id and its actions only, never on the launch's size or on which workgroup it rides in — and the whole batch stays finite and in range."""
    torch = torch_cuda
    big_n, n, steps = 131072, 16384, 48
    big = make_env(scene.build_config("TT", num_envs=big_n, seed=11))
    ref = make_env(scene.build_config("TT", num_envs=n, seed=11))
    gen = torch.Generator(device="cuda").manual_seed(4)
    for t in range(steps):
        a = torch.rand(big_n, 7, device="cuda", generator=gen) * 2 - 1
        big.step(a)
        ref.step(a[:n].contiguous())
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf"):
        assert torch.equal(getattr(big, name)[:n], getattr(ref, name)), name
    assert torch.equal(big.ball[:, :n], ref.ball) and torch.equal(big.episode[:n], ref.episode)
    assert bool(torch.isfinite(big.obs_buf).all()) and bool(torch.isfinite(big.rew_buf).all())
    assert float((big.ball[3:7].pow(2).sum(0) - 1).abs().max()) < 1e-4 and float(big.ball[2].min()) > -0.05
    assert big.status == 0
    big.close(); ref.close()


def test_step_sequence_is_the_same_steps(torch_cuda):
    """ppenv_step_sequence: K steps launched by one native call end in the same state, bit for bit, as K ppenv_step calls on the same actions
    (a ragged env count; also the empty sequence and a bad shape)."""
    torch = torch_cuda
    n = 1000
    a_env, b_env = make_env(scene.build_config("TT", num_envs=n, seed=5)), make_env(scene.build_config("TT", num_envs=n, seed=5))
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = [(torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1) for _ in range(37)]
    for a in acts:
        a_env.step(a)
    b_env.step_sequence(acts[:20])
    b_env.step_sequence([])
    b_env.step_sequence(acts[20:])
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "dof_pos", "dof_vel", "ball", "flags", "episode"):
        assert torch.equal(getattr(a_env, name), getattr(b_env, name)), name
    with pytest.raises(AssertionError):
        b_env.step_sequence([acts[0][:10]])
    a_env.close(); b_env.close()


def test_vec_task_surface(torch_cuda):
    torch = torch_cuda
    import isaacgym_amd
    task = isaacgym_amd.make(seed=1, task="HumanoidPingpongTiltG1", num_envs=512, sim_device="cuda:0", rl_device="cuda:0")
    assert task.num_envs == 512 and task.num_obs == 80 and task.num_actions == 7
    assert task.obs_buf.shape == (512, 80) and task.obs_buf.dtype == torch.float32
    assert task.reset_buf.dtype == torch.int64 and task.progress_buf.dtype == torch.int64
    obs0 = task.reset()["obs"]
    assert obs0.data_ptr() == task.obs_buf.data_ptr()      # zero-copy: the dict aliases the native buffer
    assert float(obs0.abs().sum()) > 0
    for _ in range(10):
        obs, rew, reset, extras = task.step(torch.rand(512, 7, device="cuda") * 2 - 1)
    from isaacgym_amd import distributed as D
    got = D.env_horizon_stats(task.env).cpu().numpy()
    want = D.horizon_stats(task.rew_buf, task.progress_buf, task.env.episode).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)
    assert obs["obs"].shape == (512, 80) and rew.shape == (512,) and reset.shape == (512,)
    assert "time_outs" in extras
    assert int(task.progress_buf.max()) == 10
    task.refresh_sim_tensors()
    assert task.root_states.shape == (512, 3, 13) and task.body_states.shape == (512, 42, 13)
    assert task.vec_dof_states.shape == (512, 7, 2) and task.dof_force_tensor.shape == (512, 7)
    assert task.no_bounce_before_half_mask.dtype == torch.bool
    with pytest.raises(ValueError):
        isaacgym_amd.make(task="HumanoidPingpongTiltG1", num_envs=4, sim_device="cpu", rl_device="cpu")


@pytest.mark.parametrize("variant", ["TT", "TN", "T3"])
def test_long_run_stays_physical(torch_cuda, variant):
    """3000 steps of random actions: nothing blows up, limits hold, the ball stays in the scene, episodes turn over."""
    torch = torch_cuda
    n = 4096
    env = make_env(scene.build_config(variant, num_envs=n, seed=17))
    lo = torch.tensor([env.config.joint[j].lower for j in range(7)], device="cuda")[:, None]
    hi = torch.tensor([env.config.joint[j].upper for j in range(7)], device="cuda")[:, None]
    vmax = torch.tensor([env.config.joint[j].vel_limit for j in range(7)], device="cuda")[:, None]
    gen = torch.Generator(device="cuda").manual_seed(5)
    rew_sum = torch.zeros(n, device="cuda", dtype=torch.float64)
    max_speed = torch.zeros((), device="cuda")
    for t in range(3000):
        env.step(torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1)
        rew_sum += env.rew_buf
        if t % 50 == 0:
            max_speed = torch.maximum(max_speed, env.ball[7:10].norm(dim=0).max())
    assert bool(torch.isfinite(env.obs_buf).all()) and bool(torch.isfinite(rew_sum).all()) and bool(torch.isfinite(env.ball).all())
    assert bool((env.dof_pos >= lo - 1e-6).all()) and bool((env.dof_pos <= hi + 1e-6).all())
    assert bool((env.dof_vel.abs() <= vmax + 1e-4).all())
    # dof_force is reported within the actuator's effort limit
    effort = torch.tensor([env.config.joint[j].effort for j in range(7)], device="cuda")[:, None]
    assert bool((env.dof_force.abs() <= effort * (1 + 1e-6)).all()), float((env.dof_force.abs() / effort).max())
    assert float((env.ball[3:7].pow(2).sum(0) - 1).abs().max()) < 1e-4
    assert float(env.ball[2].min()) > -0.05 and float(env.ball[0:3].abs().max()) < 50.0
    assert float(max_speed) < 60.0                           # restitution <= 1: the ball cannot keep gaining energy
    episodes = env.episode.to(torch.int64)
    assert int(episodes.min()) >= (3000 // env.config.max_episode_length)   # every env hit at least its time-outs
    env.close()


def test_step_is_graph_capturable(torch_cuda):
    """ppenv_step only enqueues work on the caller's stream (no sync, no allocation), so a rollout step can be captured
    into a HIP graph and replayed; the replayed steps must equal eager ones bit for bit."""
    torch = torch_cuda
    n = 2048
    eager = make_env(scene.build_config("TT", num_envs=n, seed=9))
    graphed = make_env(scene.build_config("TT", num_envs=n, seed=9))
    actions = torch.zeros(n, 7, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step(actions)                 # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    eager.step(actions)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step(actions)                 # one captured launch reading `actions` in place
    eager.step(actions)                       # capture does not execute: keep the twins in lockstep by ...
    g.replay()                                # ... replaying once here
    gen = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(50):
        a = torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1
        actions.copy_(a)
        g.replay()
        eager.step(a)
    torch.cuda.synchronize()
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "ball", "dof_pos", "flags"):
        assert torch.equal(getattr(eager, name), getattr(graphed, name)), name
    eager.close()
    graphed.close()


def test_rlgames_adapter_drives_the_native_task(torch_cuda):
    """The train.py wiring (reference train.py:122-150,167): env creator thunk + RLGPUEnv, with a yaml-shaped task cfg."""
    torch = torch_cuda
    from isaacgym_amd.rlgames_utils import RLGPUEnv, get_rlgames_env_creator
    yaml_like = {"name": "HumanoidPingpongTiltNoEarlyStopG1",   # keys as in cfg/task/HumanoidPingpongTiltNoEarlyStopG1.yaml
                 "env": {"numEnvs": 256, "episodeLength": 170, "alphaVelocityReward": 1000, "powerCoefficient": 0.002, "penalty": -600,
                         "hitTableReward": 2000, "nothitTablePenalty": -1000, "clipActions": 1.0,
                         "bodyStatesId": [0, 31, 32, 33, 34, 35, 36, 37, 38, 39],
                         "plane": {"staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0}},
                 "task": {"randomize": False}}
    thunk = get_rlgames_env_creator(seed=3, task_config=yaml_like, task_name=yaml_like["name"], sim_device="cuda:0", rl_device="cuda:0",
                                    graphics_device_id=-1, headless=True)
    venv = RLGPUEnv("rlgpu", 256, env_creator=thunk)
    info = venv.get_env_info()
    assert info["observation_space"].shape == (80,) and info["action_space"].shape == (7,) and venv.get_number_of_agents() == 1
    obs = venv.reset()["obs"]
    total = torch.zeros(256, device="cuda")
    for _ in range(200):
        obs_d, rew, done, extras = venv.step(torch.rand(256, 7, device="cuda") * 2 - 1)
        total += rew
    assert obs_d["obs"].shape == (256, 80) and bool(torch.isfinite(total).all()) and int(done.sum()) >= 0
    assert int(venv.env.env.episode.sum()) >= 256          # 170-step episodes: everyone timed out once


@pytest.mark.parametrize("variant,n", [("TT", 1000), ("TT", 40000), ("T4", 700)])
def test_reduce_stats_matches_torch_sums(torch_cuda, variant, n):
    """ppenv_reduce_stats (what TT:763-766 prints, on the device): the one-workgroup kernel (<= 32768 rows) and the
    many-workgroup one give torch's sums; rows = agents * envs for the two-agent task."""
    torch = torch_cuda
    env = make_env(scene.build_config(variant, num_envs=n, seed=4))
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(70):
        env.step(torch.rand(n * env.num_agents, 7, device="cuda", generator=gen) * 2 - 1)
    s = env.reduce_stats().cpu().numpy()
    want = [float(env.rew_buf.double().sum()), float(env.progress_buf.double().sum()), float(env.episode.double().sum()) * env.num_agents,
            float(n * env.num_agents)]
    np.testing.assert_allclose(s, want, rtol=1e-9, atol=1e-6)
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# reset_idx(env_ids), the device status word, controlFrequencyInv


@pytest.mark.parametrize("variant", ["TT", "TN", "T4"])
def test_reset_idx_resets_only_the_listed_envs(torch_cuda, oracle_lib, variant):
    """ppenv_reset_idx = reset_idx(env_ids) -> _reset_idx (TT:809-812, 847-906) against the oracle's: the listed envs get their
    next episode's serve, progress 0 and the initial flags (TN keeps its dof state, TN:888-901), every other env is untouched."""
    torch = torch_cuda
    n = 300
    cfg = scene.build_config(variant, num_envs=n, seed=13)
    o = oracle_lib.OracleEnv(cfg, threads=8)
    env = make_env(scene.build_config(variant, num_envs=n, seed=13))
    A = env.num_agents
    rng = np.random.default_rng(3)
    for t in range(25):                                   # get away from the initial state
        o.step(rng.uniform(-1, 1, (n * A, 7)).astype(np.float32))
    env.set_state(o.get_state())
    before = DevView(env)
    ids = np.unique(rng.integers(0, n, 70))
    ids_dup = np.concatenate([ids, ids[:5]])              # duplicates are harmless
    o.reset_idx(ids)
    env.reset_idx(torch.from_numpy(ids_dup).cuda())       # device ids, as the reference's reset_buf.nonzero() result
    v = DevView(env)
    for name in ("dof_pos", "dof_vel", "ball"):
        np.testing.assert_allclose(getattr(v, name), getattr(o, name), rtol=1e-6, atol=1e-6, err_msg=name)
    np.testing.assert_array_equal(v.progress_buf, o.progress_buf)
    np.testing.assert_array_equal(v.flags, o.flags)
    np.testing.assert_array_equal(v.episode, o.episode)
    rows = np.repeat(np.isin(np.arange(n), ids), A)
    assert_close(v.obs_buf[rows], o.obs_buf[rows], "obs rows of the reset envs", atol=obs_atol())
    other = ~np.isin(np.arange(n), ids)
    for name in ("dof_pos", "dof_vel", "dof_force", "ball"):
        np.testing.assert_array_equal(getattr(v, name)[:, other], getattr(before, name)[:, other], err_msg=f"{name} of untouched envs")
    np.testing.assert_array_equal(v.episode[other], before.episode[other])
    np.testing.assert_array_equal(v.progress_buf[np.repeat(other, A)], before.progress_buf[np.repeat(other, A)])
    assert (v.episode[ids] == before.episode[ids] + 1).all() and (v.progress_buf[rows] == 0).all()
    if variant == "TN":
        np.testing.assert_array_equal(v.dof_pos[:, ids], before.dof_pos[:, ids])     # TN:888-901
    with pytest.raises(IndexError):
        env.reset_idx([n])
    env.close()


def test_vectask_reset_idx_and_control_frequency(torch_cuda):
    """VecTask.reset_idx(env_ids) resets the subset only; controlFrequencyInv = 2 is two simulate calls ahead of ONE post_physics_step."""
    torch = torch_cuda
    import isaacgym_amd
    n = 256
    task = isaacgym_amd.make(seed=4, task="HumanoidPingpongTiltG1", num_envs=n)
    a = torch.zeros(n, 7, device="cuda")
    for _ in range(10):
        task.step(a)
    ep0, prog0 = task.env.episode.clone(), task.progress_buf.clone()
    ids = torch.tensor([3, 77, 200], device="cuda")
    task.reset_idx(ids)
    assert (task.env.episode[ids] == ep0[ids] + 1).all() and (task.progress_buf[ids] == 0).all()
    mask = torch.ones(n, dtype=torch.bool, device="cuda")
    mask[ids] = False
    assert torch.equal(task.env.episode[mask], ep0[mask]) and torch.equal(task.progress_buf[mask], prog0[mask])
    assert int(task.reset_buf_force.abs().sum()) == 0

    # controlFrequencyInv: one step of a k = 2 task == two substep-pairs of physics, progress + 1, one reward
    cfg = scene.default_task_cfg("TT")
    cfg["env"]["numEnvs"], cfg["env"]["controlFrequencyInv"], cfg["seed"] = n, 2, 4
    from isaacgym_amd.tasks import isaacgym_task_map
    t2 = isaacgym_task_map["HumanoidPingpongTiltG1"](cfg, "cuda:0", "cuda:0", -1, True, False, False)
    assert t2.native_config.substeps == 4 and abs(t2.native_config.dt - 2 * 0.0083) < 1e-7
    t1 = isaacgym_amd.make(seed=4, task="HumanoidPingpongTiltG1", num_envs=n)
    ball0 = t1.env.ball.clone()
    t2.step(a)
    assert int(t2.progress_buf.max()) == 1
    # free flight over two sim steps: x advances by ~ 2 dt vx (no contact in the first steps of a serve)
    dx = (t2.env.ball[0] - ball0[0]).cpu().numpy()
    np.testing.assert_allclose(dx, 2 * 0.0083 * ball0[7].cpu().numpy(), rtol=1e-3)


@pytest.mark.parametrize("schedule", ["split", "quad"])
def test_handoff_timeout_is_reported_not_stored(torch_cuda, monkeypatch, schedule):
    """A wave of the multi-wave step kernel that never receives its partner's LDS hand-off must not store plausible garbage:
    it sets the handle's status word and every later call fails with PPENV_EDEVICE.  PPENV_DEBUG_DROP_HANDOFF withholds the
    arm wave's last hand-off so that the ball wave's bounded wait runs out."""
    torch = torch_cuda
    from isaacgym_amd import _lib
    monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)
    monkeypatch.setenv("PPENV_DEBUG_DROP_HANDOFF", "1")
    env = make_env(scene.build_config("TT", num_envs=64, seed=2))
    monkeypatch.delenv("PPENV_DEBUG_DROP_HANDOFF")
    assert env.status == 0
    ball0, rew0 = env.ball.clone(), env.rew_buf.clone()
    env.step(torch.zeros(64, 7, device="cuda"))          # the launch itself succeeds; the fault is reported by the device
    torch.cuda.synchronize()
    assert env.status & scene.STATUS_HANDOFF_TIMEOUT
    assert torch.equal(env.ball, ball0) and torch.equal(env.rew_buf, rew0)     # the timed-out wave stored nothing
    with pytest.raises(_lib.PPEnvError, match="hand-off"):
        env.step(torch.zeros(64, 7, device="cuda"))
    with pytest.raises(_lib.PPEnvError):
        env.get_state()
    with pytest.raises(_lib.PPEnvError):
        env.reduce_stats()
    env.close()
    ok = make_env(scene.build_config("TT", num_envs=64, seed=2))               # a healthy handle is unaffected
    ok.step(torch.zeros(64, 7, device="cuda"))
    torch.cuda.synchronize()
    assert ok.status == 0
    ok.close()


# ---------------------------------------------------------------------------------------------------------------------
# N4: domain randomisation tables + extras


def _dr_tables(n, rng):
    return dict(dof_stiffness_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32), dof_damping_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32),
                link_mass_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32), restitution_scale=rng.uniform(0.0, 0.7, n).astype(np.float32),
                friction_scale=rng.uniform(0.7, 1.3, n).astype(np.float32))


def test_randomization_off_is_bit_identical_and_unit_tables_change_nothing(torch_cuda):
    """ppenv_set_randomization(NULL) / never set: the step kernel that ran before runs, bit for bit.  And the table-reading instantiation
    with every scale 1 and no noise reproduces the one-wave kernel it is an instantiation of."""
    torch = torch_cuda
    n = 1000
    mk = lambda: make_env(scene.build_config("TT", num_envs=n, seed=9))
    a, b = mk(), mk()
    b.set_randomization(**{k: np.full_like(v, 1.7) for k, v in _dr_tables(n, np.random.default_rng(0)).items()}, action_noise_sigma=0.5)
    b.clear_randomization()
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(150):
        act = torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1
        a.step(act)
        b.step(act)
    for name in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "dof_pos", "dof_vel", "dof_force", "ball", "flags", "episode"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    a.close(); b.close()


@pytest.mark.parametrize("schedule", ["split", "fused"])
def test_unit_randomization_tables_reproduce_the_plain_kernel(torch_cuda, monkeypatch, schedule):
    torch = torch_cuda
    n = 512
    monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)      # the DR instantiation of either schedule = that schedule with table reads
    a, b = make_env(scene.build_config("TT", num_envs=n, seed=9)), make_env(scene.build_config("TT", num_envs=n, seed=9))
    b.set_randomization(**{k: np.ones_like(v) for k, v in _dr_tables(n, np.random.default_rng(0)).items()})
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(40):
        act = torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1
        b.set_state(a.get_state())
        a.step(act)
        b.step(act)
        for name in ("obs_buf", "rew_buf", "dof_pos", "dof_vel", "ball"):
            x, y = getattr(a, name), getattr(b, name)
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-5), (name, t, float((x - y).abs().max()))
        assert torch.equal(a.reset_buf, b.reset_buf)
    a.close(); b.close()


@pytest.mark.parametrize("schedule", ["split", "fused"])
def test_randomized_step_matches_oracle(torch_cuda, oracle_lib, monkeypatch, schedule):
    """Randomisation on: per-env stiffness / damping / mass / restitution / friction tables, action and observation noise, another
    gravity — the kernel against the oracle with the same tables, restarted from the oracle's state every step.  Both table-reading
    instantiations: the two-wave schedule (round 3, the default) and the one-wave kernel."""
    torch = torch_cuda
    monkeypatch.setenv("PPENV_STEP_KERNEL", schedule)
    n = 768
    cfg = scene.build_config("TT", num_envs=n, seed=17)
    o = oracle_lib.OracleEnv(cfg, threads=8)
    env = make_env(scene.build_config("TT", num_envs=n, seed=17))
    rng = np.random.default_rng(8)
    tabs = _dr_tables(n, rng)
    kw = dict(action_noise_sigma=0.02, observation_noise_sigma=0.002)          # yaml:106-113
    probe = SensitivityProbe(oracle_lib, cfg)
    for x in (o, probe.o2):
        x.set_randomization(**tabs, **kw)
        x.set_gravity(-9.8 - 0.3)
    env.set_randomization(**tabs, **kw)
    env.set_gravity(-9.8 - 0.3)
    oa, ra = obs_atol() + 2e-6, reward_atol(cfg)
    log = ExclusionLog(f"gpu randomised fused step vs oracle [TT, {schedule}]", bound=0.01)
    resets = 0
    for t in range(120):
        actions = rng.uniform(-1.2, 1.2, (n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        o.step(actions)
        env.step(torch.from_numpy(actions).cuda())
        keep = ~probe.sensitive(st, actions, o)
        log.add(keep)
        v, om = mask_envs(DevView(env), keep), mask_envs(o, keep)
        np.testing.assert_array_equal(v.reset_buf, om.reset_buf, err_msg=f"reset step {t}")
        np.testing.assert_array_equal(v.flags, om.flags, err_msg=f"flags step {t}")
        assert_state_close(v, om, f"step {t}")
        assert_close(v.obs_buf, om.obs_buf, f"obs step {t}", atol=oa)
        assert_close(v.rew_buf, om.rew_buf, f"rew step {t}", atol=ra)
        resets += int(o.reset_buf.sum())
    assert resets > 30
    log.close()
    # the randomisation did something: a second oracle without it ends up elsewhere
    plain = oracle_lib.OracleEnv(cfg, threads=8)
    plain.set_state(st)
    plain.step(actions)
    assert np.abs(plain.dof_vel - o.dof_vel).max() > 1e-2 and np.abs(plain.obs_buf - o.obs_buf).max() > 1e-3
    env.close()


def test_vectask_randomize_true_and_extras(torch_cuda):
    """cfg task.randomize = True with the reference yaml's randomization_params block (yaml:102-169) steps; extras carries
    reward_mean / progress_mean (TT:767-768) as device scalars."""
    torch = torch_cuda
    import json
    import os
    from helpers import GOLDEN_DIR
    from isaacgym_amd.tasks import isaacgym_task_map
    snap = json.load(open(os.path.join(GOLDEN_DIR, "task_cfgs.json")))["HumanoidPingpongTiltG1"]["task"]
    assert snap["task"]["randomize"] is False and snap["task"]["randomization_params"]["frequency"] == 600
    n = 256
    cfg = scene.default_task_cfg("TT")
    cfg["env"]["numEnvs"], cfg["seed"], cfg["stats_every"] = n, 3, 4
    cfg["task"] = dict(randomize=True, randomization_params=dict(snap["task"]["randomization_params"], frequency=5))
    task = isaacgym_task_map["HumanoidPingpongTiltG1"](cfg, "cuda:0", "cuda:0", -1, True, False, False)
    plain = isaacgym_task_map["HumanoidPingpongTiltG1"](dict(scene.default_task_cfg("TT"), seed=3, env=dict(cfg["env"])), "cuda:0", "cuda:0", -1, True, False, False)
    a = torch.zeros(n, 7, device="cuda")
    for t in range(12):
        od, rew, done, extras = task.step(a)
        plain.step(a)
    assert torch.isfinite(od["obs"]).all() and torch.isfinite(rew).all()
    assert task.env._dr[0].shape == (7, n) and float(task.env._dr[0].min()) > 0.49 and float(task.env._dr[2].max()) < 1.51
    assert not torch.equal(task.obs_buf, plain.obs_buf)                      # observation noise at the very least
    assert extras["reward_mean"].device.type == "cuda" and extras["reward_mean"].dim() == 0
    np.testing.assert_allclose(float(extras["reward_mean"]), float(task.rew_buf.mean()), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(extras["progress_mean"]), 12.0, rtol=1e-6)
