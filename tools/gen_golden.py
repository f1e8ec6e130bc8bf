#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code on scripted inputs.

Runs only in the build container (needs /root/reference).  For each 7-DoF task
variant it instantiates the reference task class without its Isaac Gym
constructor (object.__new__), gives it CPU torch tensors laid out exactly like
the simulator tensors it wraps (TT:153-214), and calls the reference's own
`post_physics_step()` (TT:1022-1052) once per scripted step.  That executes the
reference's `compute_reward` -> `compute_pingpong_reward*`, `reset_idx` ->
`_reset_idx` (with Python `random` driving `generate_random_speed_for_ball`)
and `compute_observations`, with sticky flags and progress carried from step to
step.  `gym.simulate` does not exist offline, so the state between steps comes
from a small scripted ball/arm generator below — it only has to visit every
branch of the reward, not be physical.

Outputs per variant (arrays over T steps x N envs), all float32 unless noted:
  in_bodies   [T,N,10,13]  rigid-body rows bodyStatesId=[0,31..39] fed to the reference
  in_root     [T,N,3,13]   actor root states before post_physics_step
  in_dof      [T,N,7,2]    dof states before
  in_dof_force[T,N,7]
  in_pre_vx   [T,N]        ball vx captured by pre_physics_step
  serve       [T,N,3]      serve velocity the reference drew for envs it reset (NaN elsewhere)
  out_rew [T,N], out_reset [T,N] i64, out_obs [T,N,80], out_progress [T,N] i64,
  out_flags [T,N] u32 (PPENV_FLAG_* packing), out_root [T,N,3,13], out_dof [T,N,7,2]
plus the scalar task constants used.  The fixture is data only; no reference
source text is stored.
"""
import contextlib
import io
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_loader  # noqa: E402
from isaacgym_amd import scene  # noqa: E402

BODY_IDS = [0, 31, 32, 33, 34, 35, 36, 37, 38, 39]
N_ENVS, T_STEPS = 48, 40


class Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class FakeGym:
    """Stands in for the gym handle: refresh_* are no-ops (tensors are injected), set_* succeed."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: True


VARIANTS = {
    "TT": dict(file="humanoid_pingpong_3_actor_tilt.py", cls="HumanoidPingpongTilt"),
    "TN": dict(file="humanoid_pingpong_3_actor_tilt_no_earlystop.py", cls=None),
    "T3": dict(file="humanoid_interos_edit_pingpong_only_3_actor.py", cls=None),
}


def find_task_class(mod):
    """The task class is the one defining post_physics_step."""
    for name, obj in vars(mod).items():
        if isinstance(obj, type) and "post_physics_step" in vars(obj):
            return obj
    raise RuntimeError("no task class found")


def make_instance(variant, mod, cfg, n, episode_length):
    cls = find_task_class(mod)
    sys.modules["isaacgym.gymapi"].Vec3 = Vec3
    mod.gymapi.Vec3 = Vec3
    o = object.__new__(cls)
    env = cfg["env"]
    o.cfg = cfg
    o.num_envs, o.device, o.headless, o.randomize = n, "cpu", True, False
    o.gym, o.sim, o.viewer = FakeGym(), None, None
    o.num_steps = 1
    o.max_episode_length = episode_length
    o.alpha = env["alphaVelocityReward"]
    o.power_coefficient = env["powerCoefficient"]
    o.penalty = env["penalty"]
    o.hit_table_reward = env["hitTableReward"]
    o.not_hit_table_penalty = env["nothitTablePenalty"]
    sc = cfg["scene"]
    o.initial_speed_range = tuple(sc["serve_speed"])
    o.tilt_angle_range = tuple(sc["serve_tilt"])
    o.tilt_z_angle_range = tuple(sc["serve_tilt_z"])
    o.actors_per_env, o.dofs_per_env = 3, 7

    o.progress_buf = torch.zeros(n, dtype=torch.long)
    o.randomize_buf = torch.zeros(n, dtype=torch.long)
    o.reset_buf = torch.ones(n, dtype=torch.long)
    o.reset_buf_force = torch.zeros(n, dtype=torch.long)
    o.rew_buf = torch.zeros(n)
    o.obs_buf = torch.zeros(n, 80)
    o.actions = torch.zeros(n, 7)

    c = scene.build_config(variant, cfg=cfg, num_envs=n)
    init_root = torch.from_numpy(scene.initial_root_states(c))
    o.root_states = init_root.repeat(n, 1).clone()                      # [N*3, 13]
    o.vec_root_states = o.root_states.view(n, 3, 13)
    o.initial_vec_root_states = o.vec_root_states.clone()
    o.initial_pos = o.initial_vec_root_states[:, :, 0:3]
    o.initial_rot = o.initial_vec_root_states[:, :, 3:7]
    o.humanoid1_root_states = o.vec_root_states[:, 0, :]
    o.table_root_states = o.vec_root_states[:, 1, :]
    o.ball2_root_states = o.vec_root_states[:, 2, :]
    o.pre_ball2_root_states = o.ball2_root_states.clone()

    o.rb_states = torch.zeros(n * 42, 13)
    o.body_states = o.rb_states.view(n, 42, 13)
    o.vec_rb_states = o.rb_states.view(n, -1, 13)
    o.humanoid1_paddle_rb_states = o.vec_rb_states[:, 39, :]
    o.body_states_id = torch.tensor(BODY_IDS, dtype=torch.long)

    o.dof_states = torch.zeros(n * 7, 2)
    o.vec_dof_states = o.dof_states.view(n, 7, 2)
    o.dof_pos = o.vec_dof_states[..., 0]
    o.dof_vel = o.vec_dof_states[..., 1]
    o.initial_dof_states = o.vec_dof_states.clone()
    o.dof_force_tensor = torch.zeros(n, 7)
    o.actor_indices = torch.arange(n * 3, dtype=torch.long)
    o.dof_indices = torch.arange(n, dtype=torch.long)

    o.reward_calculated = torch.zeros(n, dtype=torch.bool)
    o.condition_calculated = torch.zeros(n, dtype=torch.bool)
    o.no_bounce_before_half_mask = torch.ones(n, dtype=torch.bool)
    o.paddle_condition_calculated = torch.zeros(n, dtype=torch.bool)
    o.missed_ball_calculated = torch.zeros(n, dtype=torch.bool)
    o.net_condition_calculated = torch.zeros(n, dtype=torch.bool)
    return o, c


def pack_flags(variant, o):
    f = np.zeros(o.num_envs, dtype=np.uint32)
    if variant == "TT":
        f |= o.reward_calculated.numpy().astype(np.uint32) * scene.FLAG_REWARD_CALC
        f |= o.condition_calculated.numpy().astype(np.uint32) * scene.FLAG_COND_CALC
        f |= o.no_bounce_before_half_mask.numpy().astype(np.uint32) * scene.FLAG_NO_BOUNCE
    elif variant == "TN":
        f |= o.paddle_condition_calculated.numpy().astype(np.uint32) * scene.FLAG_COND_CALC
        f |= o.missed_ball_calculated.numpy().astype(np.uint32) * scene.FLAG_MISSED_CALC
        f |= scene.FLAG_NO_BOUNCE  # TN never touches this mask in its active reward; it stays set
    else:
        f |= scene.FLAG_NO_BOUNCE  # T3 has no flags; the native state keeps the initial value
    return f


def random_unit_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)


def yaw_quats(angles):
    q = np.zeros((len(angles), 4), np.float32)
    q[:, 2] = np.sin(0.5 * angles)
    q[:, 3] = np.cos(0.5 * angles)
    return q


def generate(variant, seed):
    rng = np.random.default_rng(seed)
    cfg = scene.default_task_cfg(variant)
    n, T = N_ENVS, T_STEPS
    episode_length = 12  # short so that the progress time-out branch (TT:1265) fires inside T steps
    mod = ref_loader.load_task(VARIANTS[variant]["file"])
    o, c = make_instance(variant, mod, cfg, n, episode_length)

    drawn = []
    orig = o.generate_random_speed_for_ball

    def recording(*a, **k):
        v = orig(*a, **k)
        drawn.append((v.x, v.y, v.z))
        return v
    o.generate_random_speed_for_ball = recording

    # pelvis orientation per env: identity, the T3 yaw, pure yaws, and general tilts
    pelvis_q = random_unit_quats(rng, n)
    pelvis_q[: n // 4] = np.array(c.humanoid_root_quat[:], np.float32)
    pelvis_q[n // 4: n // 2] = yaw_quats(rng.uniform(-np.pi, np.pi, n // 2 - n // 4))
    pelvis_p = np.tile(np.array(c.humanoid_root_pos[:], np.float32), (n, 1))
    pelvis_p[n // 2:] += rng.uniform(-0.2, 0.2, (n - n // 2, 3)).astype(np.float32)

    # scripted ball: half the envs fly ballistically with table bounces / paddle returns, half are i.i.d. samples
    ball = o.vec_root_states[:, 2, :].numpy().copy()
    ball[:, 7:10] = rng.uniform([-9, -0.8, -1], [-5, 0.8, 1], (n, 3))
    kin = np.arange(n) < n // 2
    dt = cfg["sim"]["dt"] * 3.0  # coarse steps so a rally fits into the short episodes
    lo = np.array([c.joint[j].lower for j in range(7)], np.float32)
    hi = np.array([c.joint[j].upper for j in range(7)], np.float32)
    dof = np.zeros((n, 7, 2), np.float32)

    keys = ["in_bodies", "in_root", "in_dof", "in_dof_force", "in_pre_vx", "serve", "out_rew", "out_reset",
            "out_obs", "out_progress", "out_flags", "out_root", "out_dof"]
    rec = {k: [] for k in keys}
    random.seed(seed)
    for t in range(T):
        pre_vx = ball[:, 7].copy()
        # --- scripted "physics"
        nb = ball.copy()
        nb[kin, 9] -= 9.8 * dt
        nb[kin, 0:3] += nb[kin, 7:10] * dt
        on_table = kin & (nb[:, 2] < 0.78) & (nb[:, 9] < 0) & (nb[:, 0] > 0.38) & (nb[:, 0] < 3.12) & (np.abs(nb[:, 1]) < 0.76)
        nb[on_table, 2] = 0.78 + (0.78 - nb[on_table, 2])
        nb[on_table, 9] *= -0.9
        at_paddle = kin & (nb[:, 0] < 0.35) & (nb[:, 7] < 0) & (rng.uniform(size=n) < 0.7)
        nb[at_paddle, 7] = rng.uniform(0.5, 7.0, at_paddle.sum())
        nb[at_paddle, 9] = rng.uniform(0.5, 3.5, at_paddle.sum())
        iid = ~kin
        nb[iid, 0:3] = rng.uniform([-0.4, -0.9, 0.0], [3.5, 0.9, 1.4], (iid.sum(), 3))
        nb[iid, 7:10] = rng.uniform([-9, -2, -4], [9, 2, 4], (iid.sum(), 3))
        # put some i.i.d. samples right inside the narrow windows (net 1.7<x<1.8, table zones)
        pick = iid & (rng.uniform(size=n) < 0.3)
        nb[pick, 0] = rng.uniform(1.68, 1.82, pick.sum())
        nb[pick, 2] = rng.uniform(0.95, 1.17, pick.sum())
        pick = iid & (rng.uniform(size=n) < 0.3)
        nb[pick, 2] = rng.uniform(0.05, 0.9, pick.sum())
        nb[:, 10:13] = rng.uniform(-20, 20, (n, 3))
        ball = nb.astype(np.float32)
        dof[:, :, 1] = rng.uniform(-6, 6, (n, 7))
        dof[:, :, 0] = np.clip(dof[:, :, 0] + dof[:, :, 1] * dt, lo, hi)
        dof_force = rng.uniform(-25, 25, (n, 7)).astype(np.float32)
        bodies = np.zeros((n, 10, 13), np.float32)
        bodies[:, :, 0:3] = rng.uniform([-0.2, -0.6, 0.7], [0.7, 0.3, 1.6], (n, 10, 3))
        bodies[:, :, 3:7] = random_unit_quats(rng, n * 10).reshape(n, 10, 4)
        bodies[:, :, 7:13] = rng.uniform(-5, 5, (n, 10, 6))
        bodies[:, 0, 0:3] = pelvis_p
        bodies[:, 0, 3:7] = pelvis_q
        bodies[:, 0, 7:13] = 0
        near = rng.uniform(size=n) < 0.4   # paddle close to the ball: exercises the proximity terms
        bodies[near, 9, 0:3] = ball[near, 0:3] + rng.normal(0, 0.05, (near.sum(), 3))

        # --- inject into the reference object's simulator tensors
        o.body_states.zero_()
        o.body_states[:, BODY_IDS, :] = torch.from_numpy(bodies)
        o.vec_root_states[:, 2, :] = torch.from_numpy(ball)
        o.vec_dof_states[:] = torch.from_numpy(dof)
        o.dof_force_tensor[:] = torch.from_numpy(dof_force)
        o.pre_ball2_root_states = o.ball2_root_states.clone()
        o.pre_ball2_root_states[:, 7] = torch.from_numpy(pre_vx)
        rec["in_bodies"].append(bodies.copy())
        rec["in_root"].append(o.vec_root_states.numpy().copy())
        rec["in_dof"].append(dof.copy())
        rec["in_dof_force"].append(dof_force.copy())
        rec["in_pre_vx"].append(pre_vx.astype(np.float32))

        # --- the reference's own post_physics_step
        drawn.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            o.post_physics_step()
        serve = np.full((n, 3), np.nan, np.float32)
        ids = np.nonzero(o.reset_buf.numpy())[0]
        assert len(ids) == len(drawn)
        for k, env_id in enumerate(ids):  # _reset_idx draws in ascending env order (TT:857-862)
            serve[env_id] = np.array(drawn[k], np.float32)
        rec["serve"].append(serve)
        rec["out_rew"].append(o.rew_buf.numpy().copy())
        rec["out_reset"].append(o.reset_buf.numpy().copy())
        rec["out_obs"].append(o.obs_buf.numpy().copy())
        rec["out_progress"].append(o.progress_buf.numpy().copy())
        rec["out_flags"].append(pack_flags(variant, o))
        rec["out_root"].append(o.vec_root_states.numpy().copy())
        rec["out_dof"].append(o.vec_dof_states.numpy().copy())
        # continue the script from whatever the reference left (reset rows included)
        ball = o.vec_root_states[:, 2, :].numpy().copy()
        dof = o.vec_dof_states.numpy().copy()

    out = {k: np.stack(v) for k, v in rec.items()}
    env = cfg["env"]
    out.update(
        variant=np.array(variant), episode_length=np.array(episode_length),
        alpha=np.array(env["alphaVelocityReward"], np.float32), power_coefficient=np.array(env["powerCoefficient"], np.float32),
        penalty=np.array(env["penalty"], np.float32), hit_table_reward=np.array(env["hitTableReward"], np.float32),
        not_hit_table_penalty=np.array(env["nothitTablePenalty"], np.float32), seed=np.array(seed),
    )
    return out


TA_BAL_IDS = [0, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 27]
TA_PARAMS = dict(episode_length=14, alpha=3000.0, power_coefficient=0.002, hit_paddle_reward=200.0,
                 miss_paddle_penalty_coefficient=-100.0, cross_net_reward=1000.0, hit_table_reward=3000.0,
                 not_hit_table_penalty=-1000.0, die_penalty=-3000.0)   # HumanoidPingpongTiltNESSparse27DOFG1.yaml:15-30


def pack_ta_flags(o):
    f = np.zeros(o.num_envs, dtype=np.uint32)
    for bit, name in ((1, "paddle_condition_calculated"), (2, "hit_table_calculated"), (4, "die_penalty_calculated"),
                      (8, "humanoid_die_calculated"), (16, "closer_to_paddle_count"), (32, "hit_paddle_count"),
                      (64, "cross_net_count"), (128, "hit_table_count"), (256, "fall_down_count")):
        f |= getattr(o, name).numpy().astype(bool).astype(np.uint32) * bit
    return f


def generate_ta(seed):
    """27-DoF variant: the reference's own post_physics_step (TA:1145-1192) on scripted [N,42,13] / [N,27,2] tensors."""
    rng = np.random.default_rng(seed)
    n, T = 32, 36      # [T,N,42,13] inputs: keep the fixture small
    mod = ref_loader.load_task("humanoid_pingpong_3_actor_all_dof.py")
    cls = find_task_class(mod)
    sys.modules["isaacgym.gymapi"].Vec3 = Vec3
    mod.gymapi.Vec3 = Vec3
    o = object.__new__(cls)
    P = TA_PARAMS
    o.num_envs, o.device, o.headless, o.randomize = n, "cpu", True, False
    o.gym, o.sim, o.viewer, o.num_steps = FakeGym(), None, None, 1
    o.max_episode_length = P["episode_length"]
    o.alpha, o.power_coefficient = P["alpha"], P["power_coefficient"]
    o.hit_paddle_reward, o.miss_paddle_penalty_coefficient = P["hit_paddle_reward"], P["miss_paddle_penalty_coefficient"]
    o.cross_net_reward_float, o.hit_table_reward = P["cross_net_reward"], P["hit_table_reward"]
    o.not_hit_table_penalty, o.die_penalty_float = P["not_hit_table_penalty"], P["die_penalty"]
    o.is_g1, o.is_train = True, True
    o.initial_speed_range, o.tilt_angle_range, o.tilt_z_angle_range = (5.0, 5.4), (-8.0, 3.0), (14.0, 24.0)   # TA:129-131
    o.initial_pos_y_range, o.initial_pos_z_range = (-0.5, 0.1), (0.96, 1.05)                                   # TA:133-134
    o.actors_per_env, o.dofs_per_env = 3, 27
    o.progress_buf = torch.zeros(n, dtype=torch.long)
    o.randomize_buf = torch.zeros(n, dtype=torch.long)
    o.reset_buf = torch.ones(n, dtype=torch.long)
    o.reset_buf_force = torch.zeros(n, dtype=torch.long)
    o.rew_buf, o.obs_buf, o.actions = torch.zeros(n), torch.zeros(n, 313), torch.zeros(n, 27)
    init_root = np.zeros((3, 13), np.float32)
    init_root[0, 0:3], init_root[0, 6] = (0.0, 0.0, 1.0), 1.0          # TA:578-579
    init_root[1, 0:3], init_root[1, 6] = (1.75, 0.0, 0.0), 1.0
    init_root[2, 0:3], init_root[2, 6] = (2.9, -0.2, 1.0), 1.0         # TA:678-680 (y, z are re-drawn at every reset)
    o.root_states = torch.from_numpy(init_root).repeat(n, 1).clone()
    o.vec_root_states = o.root_states.view(n, 3, 13)
    o.initial_vec_root_states = o.vec_root_states.clone()
    o.initial_pos, o.initial_rot = o.initial_vec_root_states[:, :, 0:3], o.initial_vec_root_states[:, :, 3:7]
    o.humanoid1_root_states, o.ball2_root_states = o.vec_root_states[:, 0, :], o.vec_root_states[:, 2, :]
    o.pre_ball2_root_states = o.ball2_root_states.clone()
    o.rb_states = torch.zeros(n * 42, 13)
    o.body_states = o.rb_states.view(n, 42, 13)
    o.vec_rb_states = o.body_states
    o.humanoid1_paddle_rb_states = o.vec_rb_states[:, 39, :]
    o.humanoid1_pelvis_rb_states = o.vec_rb_states[:, 0, :]
    o.body_states_id = torch.tensor(BODY_IDS, dtype=torch.long)              # bodyStatesIdPingpong, yaml:56
    o.body_balance_states_id = torch.tensor(TA_BAL_IDS, dtype=torch.long)    # bodyStatesIdBalance, yaml:57
    # a standing pose: the initial body states every later state is compared with (TA:200)
    init_bodies = np.zeros((n, 42, 13), np.float32)
    init_bodies[:, :, 0:3] = rng.uniform([-0.15, -0.25, 0.05], [0.15, 0.25, 1.35], (1, 42, 3))
    init_bodies[:, 0, 0:3] = (0.0, 0.0, 1.0)
    init_bodies[:, :, 6] = 1.0
    o.initial_body_states = torch.from_numpy(init_bodies.copy())
    o.dof_states = torch.zeros(n * 27, 2)
    o.vec_dof_states = o.dof_states.view(n, 27, 2)
    o.dof_pos, o.dof_vel = o.vec_dof_states[..., 0], o.vec_dof_states[..., 1]
    o.initial_dof_states = o.vec_dof_states.clone()
    o.initial_dof_pos, o.initial_dof_vel = o.initial_dof_states[..., 0], o.initial_dof_states[..., 1]
    o.dof_force_tensor = torch.zeros(n, 27)
    o.actor_indices = torch.arange(n * 3, dtype=torch.long)
    o.dof_indices = torch.arange(n, dtype=torch.long)
    for name in ("paddle_condition_calculated", "die_penalty_calculated", "humanoid_die_calculated", "hit_table_calculated",
                 "closer_to_paddle_count", "hit_paddle_count", "cross_net_count", "hit_table_count", "fall_down_count"):
        setattr(o, name, torch.zeros(n, dtype=torch.bool))

    pelvis_q = random_unit_quats(rng, n)
    pelvis_q[: n // 2] = yaw_quats(rng.uniform(-0.4, 0.4, n // 2))
    ball = o.vec_root_states[:, 2, :].numpy().copy()
    ball[:, 7:10] = rng.uniform([-6, -0.5, 1], [-4, 0.5, 2.5], (n, 3))
    kin = np.arange(n) < n // 2
    dt = 0.0083 * 4.0
    dof = np.zeros((n, 27, 2), np.float32)
    keys = ["in_bodies42", "in_root", "in_dof", "in_dof_force", "in_pre_vx", "reset_override", "out_rew", "out_reset", "out_obs",
            "out_progress", "out_flags", "out_root", "out_dof"]
    rec = {k: [] for k in keys}
    random.seed(seed)
    for t in range(T):
        pre_vx = ball[:, 7].copy()
        nb = ball.copy()
        nb[kin, 9] -= 9.8 * dt
        nb[kin, 0:3] += nb[kin, 7:10] * dt
        on_table = kin & (nb[:, 2] < 0.78) & (nb[:, 9] < 0) & (nb[:, 0] > 0.38) & (nb[:, 0] < 3.12)
        nb[on_table, 2] = 0.78 + (0.78 - nb[on_table, 2])
        nb[on_table, 9] *= -0.9
        at_paddle = kin & (nb[:, 0] < 0.45) & (nb[:, 7] < 0) & (rng.uniform(size=n) < 0.7)
        nb[at_paddle, 7] = rng.uniform(0.5, 7.0, at_paddle.sum())
        nb[at_paddle, 9] = rng.uniform(0.5, 3.5, at_paddle.sum())
        iid = ~kin
        nb[iid, 0:3] = rng.uniform([-0.4, -0.9, 0.5], [3.5, 0.9, 1.5], (iid.sum(), 3))
        nb[iid, 7:10] = rng.uniform([-6, -2, -4], [6, 2, 4], (iid.sum(), 3))
        pick = iid & (rng.uniform(size=n) < 0.25)      # net window 1.72 < x < 1.78
        nb[pick, 0] = rng.uniform(1.70, 1.80, pick.sum())
        nb[pick, 2] = rng.uniform(0.90, 1.32, pick.sum())
        pick = iid & (rng.uniform(size=n) < 0.25)      # landing window 0.82 <= z <= 0.83 (TA:1263)
        nb[pick, 2] = rng.uniform(0.815, 0.835, pick.sum())
        pick = iid & (rng.uniform(size=n) < 0.15)      # ball dropped below 0.78 (TA:1479)
        nb[pick, 2] = rng.uniform(0.5, 0.79, pick.sum())
        nb[:, 10:13] = rng.uniform(-20, 20, (n, 3))
        ball = nb.astype(np.float32)
        dof[:, :, 1] = rng.normal(0, 1.0, (n, 27))
        dof[:, :, 0] = np.clip(dof[:, :, 0] * 0.8 + rng.normal(0, 0.02, (n, 27)), -1.5, 1.5)
        dof_force = rng.uniform(-25, 25, (n, 27)).astype(np.float32)
        bodies = init_bodies.copy()
        bodies[:, :, 0:3] += rng.normal(0, 0.03, (n, 42, 3))          # small sway around the standing pose
        fallen = rng.uniform(size=n) < 0.08                             # some envs far from it (has_fallen, TA:1415)
        bodies[fallen, :, 0:3] += rng.normal(0, 0.4, (fallen.sum(), 42, 3))
        low = rng.uniform(size=n) < 0.1                                 # pelvis below 0.97 (TA:1683)
        bodies[low, 0, 2] = rng.uniform(0.7, 0.969, low.sum())
        bodies[:, :, 3:7] = random_unit_quats(rng, n * 42).reshape(n, 42, 4)
        bodies[:, 0, 3:7] = pelvis_q
        bodies[:, :, 7:13] = rng.normal(0, 0.5, (n, 42, 6))
        bodies[:, 39, 0:3] = rng.uniform([0.1, -0.6, 0.8], [0.6, 0.2, 1.4], (n, 3))
        near = rng.uniform(size=n) < 0.4
        bodies[near, 39, 0:3] = ball[near, 0:3] + rng.normal(0, 0.08, (near.sum(), 3))
        bodies = bodies.astype(np.float32)

        o.body_states[:] = torch.from_numpy(bodies)
        o.vec_root_states[:, 2, :] = torch.from_numpy(ball)
        o.vec_dof_states[:] = torch.from_numpy(dof)
        o.dof_force_tensor[:] = torch.from_numpy(dof_force)
        o.pre_ball2_root_states = o.ball2_root_states.clone()
        o.pre_ball2_root_states[:, 7] = torch.from_numpy(pre_vx)
        rec["in_bodies42"].append(bodies.copy())
        rec["in_root"].append(o.vec_root_states.numpy().copy())
        rec["in_dof"].append(dof.copy())
        rec["in_dof_force"].append(dof_force.copy())
        rec["in_pre_vx"].append(pre_vx.astype(np.float32))
        with contextlib.redirect_stdout(io.StringIO()):
            o.post_physics_step()
        out_root = o.vec_root_states.numpy().copy()
        ov = np.full((n, 5), np.nan, np.float32)
        ids = np.nonzero(o.reset_buf.numpy())[0]
        ov[ids, 0:2] = out_root[ids, 2, 1:3]       # the y, z the reference drew (TA:976-979)
        ov[ids, 2:5] = out_root[ids, 2, 7:10]      # and its serve velocity
        rec["reset_override"].append(ov)
        rec["out_rew"].append(o.rew_buf.numpy().copy())
        rec["out_reset"].append(o.reset_buf.numpy().copy())
        rec["out_obs"].append(o.obs_buf.numpy().copy())
        rec["out_progress"].append(o.progress_buf.numpy().copy())
        rec["out_flags"].append(pack_ta_flags(o))
        rec["out_root"].append(out_root)
        rec["out_dof"].append(o.vec_dof_states.numpy().copy())
        ball = out_root[:, 2, :].copy()
        dof = o.vec_dof_states.numpy().copy()
    out = {k: np.stack(v) for k, v in rec.items()}
    out["initial_bodies42"] = init_bodies[0]          # identical for every env
    out["init_root"] = init_root
    out.update({k: np.array(v, np.float32 if isinstance(v, float) else np.int64) for k, v in P.items()})
    out["seed"] = np.array(seed)
    return out


def generate_t4(seed):
    """4-actor variant: direct calls of the two TorchScript reward functions (T4:1113-1439) over a scripted
    sequence, sticky flags carried in place by the functions themselves (the class never calls them)."""
    rng = np.random.default_rng(seed)
    n, T, L = N_ENVS, T_STEPS, 12
    mod = ref_loader.load_task("humanoid_pingpong_4_actor_tilt.py")
    P = dict(alpha=50.0, power_coefficient=0.0005, penalty=-200.0, hit_table_reward=2000.0, not_hit_table_penalty=-1000.0)
    # Under @torch.jit.script the functions' `flag |= ...` statements are out-of-place: the caller's flag tensors are
    # never written, so the flags are pure inputs.  Feed every combination.
    fl = {side: None for side in (1, 2)}
    root = np.zeros((n, 4, 13), np.float32)
    root[:, 0, 0:3], root[:, 1, 0:3], root[:, 2, 0:3] = (0, 0, 1), (3.5, 0, 1), (1.75, 0, 0)   # T4:525,555,~583
    root[:, :, 6] = 1.0
    root[:, 1, 3:7] = (0, 0, 1, 0)                                                             # T4:556
    ball = np.zeros((n, 13), np.float32)
    ball[:, 0:3], ball[:, 6] = (3.15, -0.28, 1.1), 1.0                                         # T4:625
    ball[:, 7:10] = rng.uniform([-9, -0.8, -1], [-5, 0.8, 1], (n, 3))
    kin = np.arange(n) < n // 2
    dt = 0.0083 * 3.0
    progress = np.zeros(n, np.int64)
    keys = ["in_rb82", "in_root", "in_dof", "in_dof_force", "in_pre_vx", "in_progress", "in_flags1", "in_flags2", "out_rew1", "out_rew2",
            "out_reset1", "out_reset2"]
    rec = {k: [] for k in keys}

    def pack(d):
        return (d["calc"].numpy().astype(np.uint32) * scene.FLAG_REWARD_CALC | d["cond"].numpy().astype(np.uint32) * scene.FLAG_COND_CALC
                | d["nob"].numpy().astype(np.uint32) * scene.FLAG_NO_BOUNCE)
    for t in range(T):
        pre_vx = ball[:, 7].copy()
        nb = ball.copy()
        nb[kin, 9] -= 9.8 * dt
        nb[kin, 0:3] += nb[kin, 7:10] * dt
        on_table = kin & (nb[:, 2] < 0.78) & (nb[:, 9] < 0) & (nb[:, 0] > 0.38) & (nb[:, 0] < 3.12) & (np.abs(nb[:, 1]) < 0.76)
        nb[on_table, 2] = 0.78 + (0.78 - nb[on_table, 2])
        nb[on_table, 9] *= -0.9
        back1 = kin & (nb[:, 0] < 0.35) & (nb[:, 7] < 0)          # returned by humanoid 1 ...
        nb[back1, 7] = rng.uniform(3.0, 8.0, back1.sum()); nb[back1, 9] = rng.uniform(0.5, 3.0, back1.sum())
        back2 = kin & (nb[:, 0] > 3.15) & (nb[:, 7] > 0)          # ... and by humanoid 2
        nb[back2, 7] = -rng.uniform(3.0, 8.0, back2.sum()); nb[back2, 9] = rng.uniform(0.5, 3.0, back2.sum())
        iid = ~kin
        nb[iid, 0:3] = rng.uniform([-0.4, -0.9, 0.0], [3.9, 0.9, 1.4], (iid.sum(), 3))
        nb[iid, 7:10] = rng.uniform([-9, -2, -4], [9, 2, 4], (iid.sum(), 3))
        pick = iid & (rng.uniform(size=n) < 0.3)
        nb[pick, 0] = rng.uniform(1.68, 1.82, pick.sum()); nb[pick, 2] = rng.uniform(0.95, 1.17, pick.sum())
        pick = iid & (rng.uniform(size=n) < 0.3)
        nb[pick, 2] = rng.uniform(0.05, 0.9, pick.sum())
        ball = nb.astype(np.float32)
        progress = progress + 1
        root[:, 3, :] = ball
        rb = np.zeros((n, 82, 13), np.float32)
        for row, side_x in ((39, 0.3), (79, 3.2)):
            rb[:, row, 0:3] = rng.uniform([side_x - 0.3, -0.6, 0.8], [side_x + 0.3, 0.2, 1.4], (n, 3))
            near = rng.uniform(size=n) < 0.3
            rb[near, row, 0:3] = ball[near, 0:3] + rng.normal(0, 0.05, (near.sum(), 3))
        dof = rng.uniform(-6, 6, (n, 14, 2)).astype(np.float32)
        dof_force = rng.uniform(-25, 25, (n, 14)).astype(np.float32)
        tr, tb, td, tf = (torch.from_numpy(x.copy()) for x in (root, rb, dof, dof_force))
        pre = tr[:, 3, :].clone(); pre[:, 7] = torch.from_numpy(pre_vx)
        tp = torch.from_numpy(progress.copy())
        rec["in_rb82"].append(rb[:, [39, 79], :].copy())       # only the two paddle rows are read
        rec["in_root"].append(root.copy()); rec["in_dof"].append(dof.copy()); rec["in_dof_force"].append(dof_force.copy())
        rec["in_pre_vx"].append(pre_vx.astype(np.float32)); rec["in_progress"].append(progress.copy())
        for side in (1, 2):
            w = rng.integers(0, 8, n)
            w[rng.uniform(size=n) < 0.5] = 4          # half the envs in the fresh-episode state
            fl[side] = dict(calc=torch.from_numpy((w & 1) != 0), cond=torch.from_numpy((w & 2) != 0), nob=torch.from_numpy((w & 4) != 0))
            rec[f"in_flags{side}"].append(pack(fl[side]))
        outs = {}
        for side, fn, hroot, paddle in ((1, mod.compute_humanoid1_pingpong_reward, tr[:, 0, :], tb[:, 39, :]),
                                        (2, mod.compute_humanoid2_pingpong_reward, tr[:, 1, :], tb[:, 79, :])):
            d = fl[side]
            with contextlib.redirect_stdout(io.StringIO()):
                rew, reset = fn(hroot, paddle, pre, tr[:, 3, :], tf, td[..., 1], torch.zeros(n, dtype=torch.long), tp, float(L), P["alpha"],
                                P["power_coefficient"], P["penalty"], d["cond"], P["hit_table_reward"], P["not_hit_table_penalty"],
                                d["calc"], d["nob"])
            assert np.array_equal(pack(d), rec[f"in_flags{side}"][-1])   # the scripted function did not touch the caller's flags
            outs[side] = (rew.numpy().copy(), reset.numpy().copy())
        rec["out_rew1"].append(outs[1][0]); rec["out_rew2"].append(outs[2][0])
        rec["out_reset1"].append(outs[1][1]); rec["out_reset2"].append(outs[2][1])
        done = (outs[1][1] != 0) | (outs[2][1] != 0)            # the script's own episode boundary
        progress[done] = 0
        ball[done, 0:3] = (3.15, -0.28, 1.1)
        ball[done, 7:10] = rng.uniform([-9, -0.8, -1], [-5, 0.8, 1], (done.sum(), 3))
        rec.setdefault("episode_end", []).append(done.copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(episode_length=np.array(L), **{k: np.array(v, np.float32) for k, v in P.items()})
    return out


SERVE_FILES = {   # variant -> (task file, default ranges of its own class attributes: TT:111-113 etc. come from the cfg; TA:129-131 are literals)
    "T3": "humanoid_interos_edit_pingpong_only_3_actor.py",
    "TT": "humanoid_pingpong_3_actor_tilt.py",
    "TN": "humanoid_pingpong_3_actor_tilt_no_earlystop.py",
    "T4": "humanoid_pingpong_4_actor_tilt.py",
    "TA": "humanoid_pingpong_3_actor_all_dof.py",
}


def generate_serve(seed):
    """serve_draws.npz: every variant's own generate_random_speed_for_ball (T3:289-305, TT:296-323, TN:301-328, T4:299-326,
    TA:346-377) with `random.uniform` scripted: call k returns the k-th scripted draw, so the function's three (T3: two) draws
    are known inputs.  Stored per variant: draws [M,3] float64 = (speed, tilt deg, tilt_z deg) as `random.uniform` returned them,
    ranges [3,2] the function was called with, out [M,3] float64 = the Vec3 it returned, ndraws = how many draws a call made."""
    rng = np.random.default_rng(seed)
    out = {}
    for variant, fname in SERVE_FILES.items():
        mod = ref_loader.load_task(fname)
        sys.modules["isaacgym.gymapi"].Vec3 = Vec3
        mod.gymapi.Vec3 = Vec3
        cls = find_task_class(mod)
        o = object.__new__(cls)
        if variant == "TA":
            ranges = [(5.0, 5.4), (-8.0, 3.0), (14.0, 24.0)]                 # TA:129-131
        else:
            sc = scene.default_task_cfg(variant)["scene"]
            ranges = [tuple(sc["serve_speed"]), tuple(sc["serve_tilt"]), tuple(sc["serve_tilt_z"])]
        M = 96
        draws = np.stack([rng.uniform(lo, hi, M) for lo, hi in ranges], axis=1)
        # range corners and zero angles as well
        corners = np.array([[r[i] for r, i in zip(ranges, idx)] for idx in np.ndindex(2, 2, 2)], np.float64)
        draws = np.concatenate([draws, corners, [[ranges[0][0], 0.0, 0.0], [ranges[0][1], 0.0, ranges[2][0]]]])
        res, ncalls = [], []
        real_uniform = mod.random.uniform
        try:
            for row in draws:
                script = list(row)
                calls = []

                def scripted(a, b, _s=script, _c=calls):
                    _c.append((a, b))
                    return _s[len(_c) - 1]
                mod.random.uniform = scripted
                nargs = o.generate_random_speed_for_ball.__func__.__code__.co_argcount - 1   # T3's takes (speed, tilt) only (T3:289)
                v = o.generate_random_speed_for_ball(*ranges[:nargs])
                assert calls == list(ranges[:len(calls)]), (calls, ranges)           # draw order: speed, tilt, tilt_z
                ncalls.append(len(calls))
                res.append((v.x, v.y, v.z))
        finally:
            mod.random.uniform = real_uniform
        assert len(set(ncalls)) == 1
        out[f"{variant}_draws"] = draws
        out[f"{variant}_ranges"] = np.array(ranges, np.float64)
        out[f"{variant}_out"] = np.array(res, np.float64)
        out[f"{variant}_ndraws"] = np.array(ncalls[0])
    out["seed"] = np.array(seed)
    return out


def generate_pre_physics(seed):
    """pre_physics.npz: the reference's own pre_physics_step (TT:1002-1020, T3:977-995, TN:1013-1031, T4:1008-1026, TA:1124-1143)
    with `gymtorch.unwrap_tensor` stubbed to record the tensor handed to gym.set_dof_position_target_tensor.
    `_pd_action_offset / _pd_action_scale` are built the way TT:649-671 / TA:720-733 build them (float32 numpy 0.5 * (hi +- lo))
    from the joint limits of this build's model tables (the URDF limits themselves are UNVERIFIED placeholders: what is pinned
    here is the MAPPING).  Actions are clamped to +-clipActions first, as upstream VecTask.step does before calling the hook
    (SURVEY.md App. D); raw values beyond +-1 are in the fixture.  Stored per variant: actions [M,D] (raw), clip, lo / hi [D],
    pd_tar [M,D] float32 (recorded), ball [M,13] and pre_ball [M,13] (the snapshot the hook takes, TT:1020)."""
    rng = np.random.default_rng(seed)
    out = {}
    M = 64
    for variant, fname in SERVE_FILES.items():
        mod = ref_loader.load_task(fname)
        cls = find_task_class(mod)
        o = object.__new__(cls)
        if variant == "TA":
            model = scene.build_ta_model()
            lo = np.array([model.link[d + 1].lower for d in range(27)], np.float32)
            hi = np.array([model.link[d + 1].upper for d in range(27)], np.float32)
            clip = float(scene.default_task_cfg("TA")["env"].get("clipActions", 1.0))
        else:
            c = scene.build_config(variant, num_envs=1)
            lo = np.array([c.joint[j].lower for j in range(7)], np.float32)
            hi = np.array([c.joint[j].upper for j in range(7)], np.float32)
            clip = float(c.clip_actions)
        D = lo.size
        lim_low, lim_high = lo.copy(), hi.copy()
        o._pd_action_offset = ref_loader.to_torch(0.5 * (lim_high + lim_low))        # TT:664-668
        o._pd_action_scale = ref_loader.to_torch(0.5 * (lim_high - lim_low))
        o.device = "cpu"
        o.gym, o.sim = FakeGym(), None
        captured = []

        class _GT:
            @staticmethod
            def unwrap_tensor(t):
                captured.append(t.detach().clone())
                return t
        mod.gymtorch = _GT
        ball = np.zeros((M, 13), np.float32)
        ball[:, 0:3] = rng.uniform([0, -0.8, 0.1], [3.3, 0.8, 1.5], (M, 3))
        ball[:, 6] = 1.0
        ball[:, 7:13] = rng.uniform(-9, 9, (M, 6))
        o.ball2_root_states = torch.from_numpy(ball.copy())
        actions = rng.uniform(-1.6, 1.6, (M, D)).astype(np.float32)
        actions[0] = 0.0
        actions[1] = 1.0
        actions[2] = -1.0
        a = torch.clamp(torch.from_numpy(actions), -clip, clip)                        # upstream VecTask.step
        with contextlib.redirect_stdout(io.StringIO()):
            o.pre_physics_step(a)
        assert len(captured) == 1 and captured[0].dtype == torch.float32
        out[f"{variant}_actions"] = actions
        out[f"{variant}_clip"] = np.array(clip, np.float32)
        out[f"{variant}_lo"], out[f"{variant}_hi"] = lo, hi
        out[f"{variant}_pd_tar"] = captured[0].numpy().reshape(M, D)
        out[f"{variant}_ball"] = ball
        out[f"{variant}_pre_ball"] = o.pre_ball2_root_states.numpy().copy()
        assert np.array_equal(o.actions.numpy(), a.numpy())
    out["seed"] = np.array(seed)
    return out


def branch_report(variant, g):
    rew, reset, fl = g["out_rew"], g["out_reset"], g["out_flags"]
    print(f"[{variant}] steps x envs = {rew.shape}, resets {int(reset.sum())}, "
          f"timeouts {(g['out_progress'] == 0).sum() - int(0)}, "
          f"rew range [{rew.min():.1f}, {rew.max():.1f}], distinct flag words {sorted(set(fl.ravel().tolist()))}")
    for thr in (1000.0, 300.0, -150.0, -700.0):
        print(f"    rewards {'>' if thr > 0 else '<'} {thr}: {int(((rew > thr) if thr > 0 else (rew < thr)).sum())}")


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for i, variant in enumerate(("TT", "TN", "T3")):
        g = generate(variant, seed=20250 + i)
        branch_report(variant, g)
        np.savez_compressed(os.path.join(outdir, f"post_physics_{variant}.npz"), **g)
    g = generate_t4(seed=20270)
    print(f"[T4] {g['out_rew1'].shape}: side-1 rewards in [{g['out_rew1'].min():.0f}, {g['out_rew1'].max():.0f}], side-2 in "
          f"[{g['out_rew2'].min():.0f}, {g['out_rew2'].max():.0f}], side-2 hit-table rewards {int((g['out_rew2'] > 1500).sum())}, "
          f"input flag words {sorted(set(g['in_flags2'].ravel().tolist()))}")
    np.savez_compressed(os.path.join(outdir, "rewards_T4.npz"), **g)
    g = generate_ta(seed=20260)
    rew, fl = g["out_rew"], g["out_flags"]
    print(f"[TA] {rew.shape}, resets {int(g['out_reset'].sum())}, rew range [{rew.min():.1f}, {rew.max():.1f}], "
          f"flag bits seen {sorted({b for w in set(fl.ravel().tolist()) for b in range(9) if w >> b & 1})}, "
          f"rew>2500: {int((rew > 2500).sum())}, rew<-2500: {int((rew < -2500).sum())}, ref=-50 steps: {int((np.abs(rew + 50) < 30).sum())}")
    np.savez_compressed(os.path.join(outdir, "post_physics_TA.npz"), **g)
    g = generate_serve(seed=20280)
    np.savez_compressed(os.path.join(outdir, "serve_draws.npz"), **g)
    print("[serve]", {v: (g[f"{v}_draws"].shape[0], int(g[f"{v}_ndraws"])) for v in SERVE_FILES})
    g = generate_pre_physics(seed=20290)
    np.savez_compressed(os.path.join(outdir, "pre_physics.npz"), **g)
    print("[pre_physics]", {v: g[f"{v}_pd_tar"].shape for v in SERVE_FILES})
    print("[task cfgs]", generate_task_cfgs())
    print("wrote", sorted(os.listdir(outdir)))


def generate_task_cfgs():
    """tests/golden/task_cfgs.json: what isaacgym_amd.cfgyaml.compose() yields for each of the reference's cfg/task/*.yaml (+ the train
    yaml where one exists) — resolved VALUES only (nested dicts of numbers / strings / lists), so that the tests that drive the tasks
    "from their reference yaml" also run on the GPU box, where /root/reference does not exist."""
    import json
    from isaacgym_amd import cfgyaml
    out = {}
    for name in ("HumanoidPingpongG1", "HumanoidPingpongTiltG1", "HumanoidPingpongTiltNoEarlyStopG1", "HumanoidPingpongTiltNESSparse27DOFG1"):
        c = cfgyaml.compose(name, os.path.join(ref_loader.REF_ROOT, "cfg"))
        out[name] = {"task": c["task"], "train": c.get("train")}
    path = os.path.join(ROOT, "tests", "golden", "task_cfgs.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    return path


if __name__ == "__main__":
    main()
