"""Scene specification -> `ppenv_config` (include/ppenv.h).

The reference builds its scene imperatively inside `_create_envs`
(tasks/humanoid_pingpong_3_actor_tilt.py:387-691, "TT") from three URDF files
that are not part of the reference (TT:415,496,502 load them from absolute
paths on the author's machine).  This module is the data the native step needs
instead: the constants the reference *does* state (poses, materials, PD gains,
sim params, reward constants, serve distributions) taken verbatim with their
file:line, and a hand-authored kinematic / inertial table for the G1 right arm,
paddle, table, net and ball flagged UNVERIFIED because no
asset is available offline.

Everything here is host-side, float64 numpy; the result is the plain C struct.
"""
import copy
import ctypes as C
import math

import numpy as np

NUM_DOF = 7
NUM_OBS_BODIES = 10
NUM_OBS = 80
NUM_BODIES = 42
NUM_HUMANOID_BODIES = 40
NUM_ACTORS = 3
MAX_SHAPES = 8
ABI_VERSION = 2

VARIANT_T3, VARIANT_TT, VARIANT_TN, VARIANT_T4 = 0, 1, 2, 3
VARIANT_IDS = {"T3": VARIANT_T3, "TT": VARIANT_TT, "TN": VARIANT_TN, "T4": VARIANT_T4}

FLAG_REWARD_CALC, FLAG_COND_CALC, FLAG_NO_BOUNCE, FLAG_MISSED_CALC = 1, 2, 4, 8

# reference task name (tasks/__init__.py:49-53,118-120) -> variant tag
TASK_VARIANTS = {
    "HumanoidPingpongG1": "T3",
    "HumanoidPingpongTiltG1": "TT",
    "HumanoidPingpongTiltNoEarlyStopG1": "TN",
    "Humanoid12PingpongTiltG1": "T4",
}


# ---------------------------------------------------------------- ctypes mirror
class Joint(C.Structure):
    _fields_ = [
        ("origin_xyz", C.c_float * 3), ("origin_rot", C.c_float * 9), ("axis", C.c_int32),
        ("lower", C.c_float), ("upper", C.c_float), ("kp", C.c_float), ("kd", C.c_float),
        ("effort", C.c_float), ("vel_limit", C.c_float), ("armature", C.c_float),
        ("mass", C.c_float), ("com", C.c_float * 3), ("inertia", C.c_float * 6),
    ]


class Frame(C.Structure):
    _fields_ = [("link", C.c_int32), ("xyz", C.c_float * 3), ("rot", C.c_float * 9)]


class Shape(C.Structure):
    _fields_ = [("link", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("radius", C.c_float),
                ("restitution", C.c_float), ("friction", C.c_float)]


class Box(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("half", C.c_float * 3), ("restitution", C.c_float),
                ("friction", C.c_float)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("variant", C.c_int32), ("num_envs", C.c_int32),
        ("env_id_offset", C.c_int32), ("seed", C.c_uint64), ("device_id", C.c_int32),
        ("dt", C.c_float), ("substeps", C.c_int32), ("ball_substeps", C.c_int32), ("gravity_z", C.c_float),
        ("contact_offset", C.c_float), ("bounce_threshold", C.c_float),
        ("max_depenetration_velocity", C.c_float), ("clip_actions", C.c_float), ("clip_obs", C.c_float),
        ("base_pos", C.c_float * 3), ("base_rot", C.c_float * 9), ("joint", Joint * NUM_DOF),
        ("init_dof_pos", C.c_float * NUM_DOF), ("init_dof_vel", C.c_float * NUM_DOF),
        ("obs_body", Frame * NUM_OBS_BODIES), ("paddle_obs_index", C.c_int32),
        ("humanoid_root_pos", C.c_float * 3), ("humanoid_root_quat", C.c_float * 4),
        ("table_root_pos", C.c_float * 3), ("table_root_quat", C.c_float * 4),
        ("ball_init_pos", C.c_float * 3), ("ball_init_quat", C.c_float * 4),
        ("ball_radius", C.c_float), ("ball_mass", C.c_float), ("ball_inertia_factor", C.c_float),
        ("ball_restitution", C.c_float), ("ball_friction", C.c_float), ("ball_angular_damping", C.c_float),
        ("restitution_max", C.c_float),
        ("ground_z", C.c_float), ("ground_restitution", C.c_float), ("ground_friction", C.c_float),
        ("table", Box), ("net", Box),
        ("paddle_link", C.c_int32), ("paddle_center", C.c_float * 3), ("paddle_normal", C.c_float * 3),
        ("paddle_radius", C.c_float), ("paddle_half_thickness", C.c_float),
        ("paddle_restitution", C.c_float), ("paddle_friction", C.c_float),
        ("num_shapes", C.c_int32), ("shape", Shape * MAX_SHAPES),
        ("humanoid_bound_center", C.c_float * 3), ("humanoid_bound_radius", C.c_float),
        ("serve_speed_lo", C.c_float), ("serve_speed_hi", C.c_float),
        ("serve_tilt_lo_deg", C.c_float), ("serve_tilt_hi_deg", C.c_float),
        ("serve_tilt_z_lo_deg", C.c_float), ("serve_tilt_z_hi_deg", C.c_float),
        ("max_episode_length", C.c_int32), ("alpha_velocity_reward", C.c_float),
        ("power_coefficient", C.c_float), ("penalty", C.c_float), ("hit_table_reward", C.c_float),
        ("not_hit_table_penalty", C.c_float),
        ("num_humanoids", C.c_int32), ("base2_pos", C.c_float * 3), ("base2_rot", C.c_float * 9),
        ("humanoid2_root_pos", C.c_float * 3), ("humanoid2_root_quat", C.c_float * 4),
        ("shape2", Shape * MAX_SHAPES), ("humanoid2_bound_center", C.c_float * 3),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_agents", C.c_int32),
        ("obs_buf", C.c_void_p), ("rew_buf", C.c_void_p), ("reset_buf", C.c_void_p),
        ("progress_buf", C.c_void_p), ("dof_pos", C.c_void_p), ("dof_vel", C.c_void_p),
        ("dof_force", C.c_void_p), ("ball", C.c_void_p), ("flags", C.c_void_p), ("episode", C.c_void_p),
        ("serve_override", C.c_void_p),
    ]


# 27-DoF variant (tensor-API mode only): ppenv_ta_params
TA_NUM_DOF, TA_NUM_OBS, TA_NUM_BALANCE = 27, 313, 23


class TAParams(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("max_episode_length", C.c_int32), ("is_train", C.c_int32), ("env_id_offset", C.c_int32),
        ("seed", C.c_uint64),
        ("alpha_velocity_reward", C.c_float), ("power_coefficient", C.c_float), ("hit_paddle_reward", C.c_float),
        ("miss_paddle_penalty_coefficient", C.c_float), ("cross_net_reward", C.c_float), ("hit_table_reward", C.c_float),
        ("not_hit_table_penalty", C.c_float), ("die_penalty", C.c_float),
        ("init_root", (C.c_float * 7) * NUM_ACTORS), ("init_dof_pos", C.c_float * TA_NUM_DOF), ("init_dof_vel", C.c_float * TA_NUM_DOF),
        ("serve_speed_lo", C.c_float), ("serve_speed_hi", C.c_float), ("serve_tilt_lo_deg", C.c_float), ("serve_tilt_hi_deg", C.c_float),
        ("serve_tilt_z_lo_deg", C.c_float), ("serve_tilt_z_hi_deg", C.c_float),
        ("ball_y_lo", C.c_float), ("ball_y_hi", C.c_float), ("ball_z_lo", C.c_float), ("ball_z_hi", C.c_float),
    ]


def build_ta_params(num_envs, env=None, seed=0, env_id_offset=0, is_train=True):
    """ppenv_ta_params from the HumanoidPingpongTiltNESSparse27DOFG1.yaml keys (defaults = the yaml's resolve_default values)."""
    e = dict(episodeLength=160, alphaVelocityReward=3000.0, powerCoefficient=0.002, hitTableReward=3000.0, nothitTablePenalty=-1000.0,
             crossNetRewardFloat=1000.0, diePenaltyFloat=-3000.0, hitPaddleReward=200.0, missPaddlePenaltyCoefficient=-100.0)
    e.update(env or {})
    p = TAParams()
    p.num_envs, p.max_episode_length, p.is_train, p.env_id_offset, p.seed = int(num_envs), int(e["episodeLength"]), int(is_train), int(env_id_offset), int(seed)
    p.alpha_velocity_reward, p.power_coefficient = e["alphaVelocityReward"], e["powerCoefficient"]
    p.hit_paddle_reward, p.miss_paddle_penalty_coefficient = e["hitPaddleReward"], e["missPaddlePenaltyCoefficient"]
    p.cross_net_reward, p.hit_table_reward = e["crossNetRewardFloat"], e["hitTableReward"]
    p.not_hit_table_penalty, p.die_penalty = e["nothitTablePenalty"], e["diePenaltyFloat"]
    roots = [(0.0, 0.0, 1.0, 0, 0, 0, 1), (1.75, 0.0, 0.0, 0, 0, 0, 1), (2.9, -0.2, 1.0, 0, 0, 0, 1)]   # TA:578-579,  table, TA:678-680
    for a in range(NUM_ACTORS):
        for k in range(7):
            p.init_root[a][k] = roots[a][k]
    p.serve_speed_lo, p.serve_speed_hi = 5.0, 5.4                 # TA:129
    p.serve_tilt_lo_deg, p.serve_tilt_hi_deg = -8.0, 3.0          # TA:130
    p.serve_tilt_z_lo_deg, p.serve_tilt_z_hi_deg = 14.0, 24.0     # TA:131
    p.ball_y_lo, p.ball_y_hi, p.ball_z_lo, p.ball_z_hi = -0.5, 0.1, 0.96, 1.05   # TA:133-134
    return p


class T4Params(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("max_episode_length", C.c_int32), ("alpha_velocity_reward", C.c_float),
                ("power_coefficient", C.c_float), ("penalty", C.c_float), ("hit_table_reward", C.c_float),
                ("not_hit_table_penalty", C.c_float)]


def build_t4_params(num_envs, episode_length=140, alpha=50.0, power_coefficient=0.0005, penalty=-200.0, hit_table_reward=2000.0,
                    not_hit_table_penalty=-1000.0):
    """ppenv_t4_params; the 4-actor task has no yaml of its own, defaults are the Tilt yaml values."""
    p = T4Params()
    p.num_envs, p.max_episode_length = int(num_envs), int(episode_length)
    p.alpha_velocity_reward, p.power_coefficient, p.penalty = alpha, power_coefficient, penalty
    p.hit_table_reward, p.not_hit_table_penalty = hit_table_reward, not_hit_table_penalty
    return p


# ------------------------------------------------------------------ math helpers
def rpy_to_rot(r, p, y):
    """URDF fixed-axis roll-pitch-yaw -> rotation matrix (parent <- child)."""
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def quat_to_rot(q):
    """xyzw unit quaternion -> rotation matrix."""
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _inertia_mat(v6):
    xx, yy, zz, xy, xz, yz = v6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


def _inertia_vec(m):
    return [m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 2]]


def composite_inertial(parts):
    """Merge rigidly attached bodies.  parts: list of (mass, com[3], I_com[3x3], R[3x3] link<-part).
    Returns (mass, com, I about composite com in link axes)."""
    if len(parts) == 1:   # nothing to merge: keep the table values exactly (zeros stay zeros)
        mass, c, i_com, rot = parts[0]
        return mass, np.asarray(c, dtype=np.float64), rot @ np.asarray(i_com, dtype=np.float64) @ rot.T
    m = sum(p[0] for p in parts)
    com = sum(p[0] * np.asarray(p[1], dtype=np.float64) for p in parts) / m
    inertia = np.zeros((3, 3))
    for mass, c, i_com, rot in parts:
        d = np.asarray(c, dtype=np.float64) - com
        inertia += rot @ i_com @ rot.T + mass * (d.dot(d) * np.eye(3) - np.outer(d, d))
    return m, com, inertia


# ------------------------------------------------- UNVERIFIED physical model data
# Recalled from the public Unitree g1_29dof_rev_1_0.urdf and ITTF rules; NOT
# reference facts.  Replace by parsed assets when the real
# URDFs are available.
G1_RIGHT_ARM = [
    # name, origin xyz, origin rpy, axis, (lower, upper), mass, com, inertia diag, effort, vel_limit
    dict(name="right_shoulder_pitch_link", body=31, xyz=(0.0039563, -0.10021, 0.24778), rpy=(-0.27931, 0, 0),
         axis=1, limits=(-3.0892, 2.6704), mass=0.718, com=(0.0, -0.035892, -0.011628),
         inertia=(0.0004291, 0.000453, 0.000423), effort=25.0, vel=37.0),
    dict(name="right_shoulder_roll_link", body=32, xyz=(0.0, -0.038, -0.013831), rpy=(0.27925, 0, 0),
         axis=0, limits=(-2.2515, 1.5882), mass=0.643, com=(-0.000227, -0.00727, -0.063243),
         inertia=(0.0006177, 0.0006912, 0.0003894), effort=25.0, vel=37.0),
    dict(name="right_shoulder_yaw_link", body=33, xyz=(0.0, -0.00624, -0.1032), rpy=(0, 0, 0),
         axis=2, limits=(-2.618, 2.618), mass=0.734, com=(0.010773, 0.002949, -0.072009),
         inertia=(0.0009988, 0.0010605, 0.0004354), effort=25.0, vel=37.0),
    dict(name="right_elbow_link", body=34, xyz=(0.015783, 0.0, -0.080518), rpy=(0, 0, 0),
         axis=1, limits=(-1.0472, 2.0944), mass=0.6, com=(0.064956, -0.004454, -0.010062),
         inertia=(0.0002891, 0.0004152, 0.0004197), effort=25.0, vel=37.0),
    dict(name="right_wrist_roll_link", body=35, xyz=(0.100, -0.00188791, -0.010), rpy=(0, 0, 0),
         axis=0, limits=(-1.97222, 1.97222), mass=0.08544, com=(0.0171, -0.000538, 0.0),
         inertia=(5.5e-5, 5.0e-5, 3.8e-5), effort=25.0, vel=37.0),
    dict(name="right_wrist_pitch_link", body=36, xyz=(0.038, 0.0, 0.0), rpy=(0, 0, 0),
         axis=1, limits=(-1.61443, 1.61443), mass=0.48404, com=(0.023, 0.0011, -0.0011),
         inertia=(1.68e-4, 4.23e-4, 4.31e-4), effort=5.0, vel=22.0),
    dict(name="right_wrist_yaw_link", body=37, xyz=(0.046, 0.0, 0.0), rpy=(0, 0, 0),
         axis=2, limits=(-1.61443, 1.61443), mass=0.25457, com=(0.0708, -0.0001, 0.0033),
         inertia=(6.4e-5, 2.3e-4, 1.6e-4), effort=5.0, vel=22.0),
]
# rigidly attached to right_wrist_yaw_link (chain link 6)
G1_HAND = dict(name="right_rubber_hand", body=38, xyz=(0.0415, -0.003, 0.0), mass=0.457,
               com=(0.0442, -0.00015, 0.0023), inertia=(9.0e-5, 5.1e-4, 5.0e-4))
PADDLE = dict(name="pingpong_paddle", body=39, xyz_from_hand=(0.14, 0.0, 0.0), normal=(0.0, 1.0, 0.0),
              radius=0.075, half_thickness=0.005, mass=0.15)
PELVIS_TO_TORSO = (-0.0039635, 0.0, 0.044)  # waist joints are fixed in the 7-DoF asset

TABLE_GEOM = dict(length=2.74, width=1.525, top_z=0.76, slab=0.03, net_height=0.1525, net_overhang=0.1525,
                  net_half_thickness=0.002)
BALL_GEOM = dict(radius=0.02, mass=0.0027, inertia_factor=2.0 / 3.0, angular_damping=0.5)
DEFAULT_SHAPE_MATERIAL = dict(restitution=0.0, friction=1.0)  # Isaac Gym default for shapes the task never touches

# PD gains, TT:694-711 (same in T3/TN); Kd = Kp / 40
P_GAINS = [20.0, 20.0, 20.0, 20.0, 20.0, 5.0, 5.0]

# ---------------------------------------------------------- per-variant task cfg
# Keys mirror cfg/task/*.yaml (`env:` and `sim:`), values are the yaml defaults
# with OmegaConf interpolations replaced by their resolve_default values.
_SIM_DEFAULT = dict(
    dt=0.0083, substeps=2, gravity=[0.0, 0.0, -9.81],  # HumanoidPingpongTiltG1.yaml:78-83 (task forces z=-9.8, TT:331)
    physx=dict(num_position_iterations=4, num_velocity_iterations=0, contact_offset=0.0002, rest_offset=0.0,
               bounce_threshold_velocity=0.2, max_depenetration_velocity=10.0),
)

TASK_CFGS = {
    # HumanoidPingpongG1.yaml (T3).  alphaVelocityReward / powerCoefficient / penalty are read by
    # the task (T3:103-105) but absent from its yaml; the TT yaml values are the documented default.
    "T3": dict(
        name="HumanoidPingpongG1",
        env=dict(numEnvs=512, episodeLength=64, clipActions=1.0, clipObservations=float("inf"),
                 alphaVelocityReward=50.0, powerCoefficient=0.0005, penalty=-200.0,
                 hitTableReward=0.0, nothitTablePenalty=0.0,
                 bodyStatesId=[0, 31, 32, 33, 34, 35, 36, 37, 38, 39],
                 plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0)),
        sim=dict(_SIM_DEFAULT, dt=0.0166),
        scene=dict(
            humanoid_pos=(0.0, 0.0, 1.0), humanoid_quat=(0.0, 0.0, -0.2588, 0.9659),  # T3:504-506
            humanoid_material=dict(restitution=0.6, friction=0.5),                      # T3:514-516
            table_pos=(1.75, 0.0, 0.0), table_material=dict(restitution=0.7, friction=0.2),  # T3:558,563-565
            ball_pos=(3.1, -0.3, 1.3), ball_material=dict(restitution=0.9, friction=0.2),    # T3:605,611-613
            serve_speed=(6.5, 7.5), serve_tilt=(-5.0, 5.0), serve_tilt_z=(0.0, 0.0)),        # T3:289-305
    ),
    # HumanoidPingpongTiltG1.yaml (TT).  hitTableReward / nothitTablePenalty are read (TT:106-107)
    # but commented out in this yaml (line 21); the NoEarlyStop yaml values are the documented default.
    "TT": dict(
        name="HumanoidPingpongTiltG1",
        env=dict(numEnvs=4, episodeLength=140, clipActions=1.0, clipObservations=float("inf"),
                 alphaVelocityReward=50.0, powerCoefficient=0.0005, penalty=-200.0,
                 hitTableReward=2000.0, nothitTablePenalty=-1000.0,
                 bodyStatesId=[0, 31, 32, 33, 34, 35, 36, 37, 38, 39],
                 plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0)),
        sim=dict(_SIM_DEFAULT),
        scene=dict(
            humanoid_pos=(0.0, 0.0, 1.0), humanoid_quat=(0.0, 0.0, 0.0, 1.0),            # TT:522-523
            humanoid_material=dict(restitution=0.6, friction=0.5),                       # TT:531-533
            table_pos=(1.75, 0.0, 0.0), table_material=dict(restitution=1.5, friction=0.4),  # TT:575,580-582
            ball_pos=(3.15, -0.28, 1.1), ball_material=dict(restitution=1.5, friction=0.2),  # TT:622,628-630
            serve_speed=(8.0, 8.6), serve_tilt=(-5.0, 5.0), serve_tilt_z=(2.0, 10.0)),       # TT:111-113
    ),
    # HumanoidPingpongTiltNoEarlyStopG1.yaml (TN)
    "TN": dict(
        name="HumanoidPingpongTiltNoEarlyStopG1",
        env=dict(numEnvs=1024, episodeLength=170, clipActions=1.0, clipObservations=float("inf"),
                 alphaVelocityReward=1000.0, powerCoefficient=0.002, penalty=-600.0,
                 hitTableReward=2000.0, nothitTablePenalty=-1000.0,
                 bodyStatesId=[0, 31, 32, 33, 34, 35, 36, 37, 38, 39],
                 plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0)),
        sim=dict(_SIM_DEFAULT),
        scene=dict(
            humanoid_pos=(0.0, 0.0, 1.0), humanoid_quat=(0.0, 0.0, 0.0, 1.0),            # TN:527-528
            humanoid_material=dict(restitution=0.6, friction=0.5),                       # TN:537-539
            table_pos=(1.75, 0.0, 0.0), table_material=dict(restitution=1.5, friction=0.2),  # TN:586-588
            ball_pos=(2.9, -0.28, 1.02), ball_material=dict(restitution=1.5, friction=0.2),  # TN:628
            serve_speed=(5.4, 5.9), serve_tilt=(-5.0, 5.0), serve_tilt_z=(10.0, 17.0)),      # TN:301-328
    ),
}


# 4-actor variant (T4).  No yaml of its own in the reference (tasks/__init__.py:52,122 register the class, cfg/task has
# no file): env / sim values are the Tilt yaml's, poses and materials are the class's (T4:525-526,555-556,583-585,625-631).
TASK_CFGS["T4"] = dict(
    name="Humanoid12PingpongTiltG1",
    env=dict(TASK_CFGS["TT"]["env"], numEnvs=1024),
    sim=dict(_SIM_DEFAULT),
    scene=dict(TASK_CFGS["TT"]["scene"],
               humanoid2_pos=(3.5, 0.0, 1.0), humanoid2_quat=(0.0, 0.0, 1.0, 0.0)),   # T4:555-556
)


def default_task_cfg(variant):
    """A fresh copy of the task cfg dict (`env`, `sim`, `scene`) for variant 'T3' | 'TT' | 'TN'."""
    return copy.deepcopy(TASK_CFGS[variant])


def _combine(ball_mat, other_mat, e_max):
    """PhysX default combine mode (average) after clamping restitution to [0, e_max]."""
    e = 0.5 * (min(ball_mat["restitution"], e_max) + min(other_mat["restitution"], e_max))
    mu = 0.5 * (ball_mat["friction"] + other_mat["friction"])
    return e, mu


def _set(arr, values):
    for i, v in enumerate(values):
        arr[i] = float(v)


def build_config(variant, cfg=None, num_envs=None, seed=0, device_id=0, env_id_offset=0, ball_substeps=4,
                 restitution_max=1.0):
    """Build the C config for variant 'T3' | 'TT' | 'TN' from a task cfg dict (default: the yaml defaults)."""
    if cfg is None:
        cfg = default_task_cfg(variant)
    env, sim, scene = cfg["env"], cfg["sim"], cfg["scene"]
    c = Config()
    c.abi_version = ABI_VERSION
    c.variant = VARIANT_IDS[variant]
    c.num_envs = int(num_envs if num_envs is not None else env["numEnvs"])
    c.env_id_offset = int(env_id_offset)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.device_id = int(device_id)

    c.dt = sim["dt"]
    c.substeps = int(sim["substeps"])
    c.ball_substeps = int(ball_substeps)
    c.gravity_z = -9.8  # TT:329-331 (T3:310-312, TN:333-335) override the yaml's -9.81
    px = sim["physx"]
    c.contact_offset = px["contact_offset"]
    c.bounce_threshold = px["bounce_threshold_velocity"]
    c.max_depenetration_velocity = px["max_depenetration_velocity"]
    c.clip_actions = env.get("clipActions", 1.0)
    c.clip_obs = env.get("clipObservations", float("inf"))

    # --- articulated arm
    root_p = np.asarray(scene["humanoid_pos"], dtype=np.float64)
    root_q = np.asarray(scene["humanoid_quat"], dtype=np.float64)
    root_q = root_q / np.linalg.norm(root_q)
    root_r = quat_to_rot(root_q)
    base_p = root_p + root_r @ np.asarray(PELVIS_TO_TORSO)
    _set(c.base_pos, base_p)
    _set(c.base_rot, root_r.reshape(-1))
    for j, spec in enumerate(G1_RIGHT_ARM):
        jt = c.joint[j]
        _set(jt.origin_xyz, spec["xyz"])
        _set(jt.origin_rot, rpy_to_rot(*spec["rpy"]).reshape(-1))
        jt.axis = spec["axis"]
        lo, hi = spec["limits"]
        jt.lower, jt.upper = min(lo, hi), max(lo, hi)  # TT:639-645 swaps inverted limits
        jt.kp = P_GAINS[j]
        jt.kd = P_GAINS[j] / 40.0  # TT:711
        jt.effort = spec["effort"]
        jt.vel_limit = spec["vel"]
        jt.armature = 0.0
        parts = [(spec["mass"], spec["com"], np.diag(spec["inertia"]), np.eye(3))]
        if j == NUM_DOF - 1:
            hand_o = np.asarray(G1_HAND["xyz"])
            parts.append((G1_HAND["mass"], hand_o + np.asarray(G1_HAND["com"]), np.diag(G1_HAND["inertia"]), np.eye(3)))
            pad_o = hand_o + np.asarray(PADDLE["xyz_from_hand"])
            m, r = PADDLE["mass"], PADDLE["radius"]
            n = np.asarray(PADDLE["normal"], dtype=np.float64)
            i_disc = 0.25 * m * r * r * np.eye(3) + 0.25 * m * r * r * np.outer(n, n)  # axial 1/2, transverse 1/4
            parts.append((m, pad_o, i_disc, np.eye(3)))
        mass, com, inertia = composite_inertial(parts)
        jt.mass = mass
        _set(jt.com, com)
        _set(jt.inertia, _inertia_vec(inertia))
    _set(c.init_dof_pos, [0.0] * NUM_DOF)  # humanoid_dof_states = np.zeros, TT:471,547
    _set(c.init_dof_vel, [0.0] * NUM_DOF)

    # --- observed bodies: bodyStatesId = [0, 31..39]
    ids = list(env["bodyStatesId"])
    if ids != [0, 31, 32, 33, 34, 35, 36, 37, 38, 39]:
        raise ValueError("this build supports bodyStatesId = [0,31..39] (pelvis + right arm chain + hand + paddle)")
    f = c.obs_body[0]  # pelvis: static at the root pose
    f.link = -1
    _set(f.xyz, root_p)
    _set(f.rot, root_r.reshape(-1))
    for j in range(NUM_DOF):
        f = c.obs_body[1 + j]
        f.link = j
        _set(f.xyz, (0, 0, 0))
        _set(f.rot, np.eye(3).reshape(-1))
    f = c.obs_body[8]  # right_rubber_hand
    f.link = NUM_DOF - 1
    _set(f.xyz, G1_HAND["xyz"])
    _set(f.rot, np.eye(3).reshape(-1))
    f = c.obs_body[9]  # pingpong_paddle
    f.link = NUM_DOF - 1
    pad_o = np.asarray(G1_HAND["xyz"]) + np.asarray(PADDLE["xyz_from_hand"])
    _set(f.xyz, pad_o)
    _set(f.rot, np.eye(3).reshape(-1))
    c.paddle_obs_index = 9

    # --- actor roots
    _set(c.humanoid_root_pos, root_p)
    _set(c.humanoid_root_quat, root_q)
    _set(c.table_root_pos, scene["table_pos"])
    _set(c.table_root_quat, (0, 0, 0, 1))
    _set(c.ball_init_pos, scene["ball_pos"])
    _set(c.ball_init_quat, (0, 0, 0, 1))

    # --- ball + materials (combined coefficients are precomputed here)
    ball_mat = scene["ball_material"]
    c.ball_radius = BALL_GEOM["radius"]
    c.ball_mass = BALL_GEOM["mass"]
    c.ball_inertia_factor = BALL_GEOM["inertia_factor"]
    c.ball_restitution = ball_mat["restitution"]
    c.ball_friction = ball_mat["friction"]
    c.ball_angular_damping = BALL_GEOM["angular_damping"]
    c.restitution_max = restitution_max

    plane = env["plane"]
    c.ground_z = 0.0
    c.ground_restitution, c.ground_friction = _combine(
        ball_mat, dict(restitution=plane["restitution"], friction=plane["dynamicFriction"]), restitution_max)

    tp = np.asarray(scene["table_pos"], dtype=np.float64)
    g = TABLE_GEOM
    _set(c.table.center, (tp[0], tp[1], tp[2] + g["top_z"] - 0.5 * g["slab"]))
    _set(c.table.half, (0.5 * g["length"], 0.5 * g["width"], 0.5 * g["slab"]))
    c.table.restitution, c.table.friction = _combine(ball_mat, scene["table_material"], restitution_max)
    _set(c.net.center, (tp[0], tp[1], tp[2] + g["top_z"] + 0.5 * g["net_height"]))
    _set(c.net.half, (g["net_half_thickness"], 0.5 * g["width"] + g["net_overhang"], 0.5 * g["net_height"]))
    # only table_shape_props[0] is given the table material (TT:580-582); the net keeps the default
    c.net.restitution, c.net.friction = _combine(ball_mat, DEFAULT_SHAPE_MATERIAL, restitution_max)

    hum_e, hum_mu = _combine(ball_mat, scene["humanoid_material"], restitution_max)
    c.paddle_link = NUM_DOF - 1
    _set(c.paddle_center, pad_o)
    _set(c.paddle_normal, PADDLE["normal"])
    c.paddle_radius = PADDLE["radius"]
    c.paddle_half_thickness = PADDLE["half_thickness"]
    c.paddle_restitution, c.paddle_friction = hum_e, hum_mu

    # six capsule / sphere shapes + the paddle = the 7 collision shapes of TT:528-530 (UNVERIFIED geometry)
    shoulder_w = base_p + root_r @ np.asarray(G1_RIGHT_ARM[0]["xyz"])
    shapes = [
        dict(link=6, a=(0.07, 0.0, 0.0), b=(0.07, 0.0, 0.0), radius=0.035),          # hand
        dict(link=3, a=(0.0, 0.0, -0.01), b=(0.10, 0.0, -0.01), radius=0.03),        # forearm
        dict(link=1, a=(0.0, 0.0, -0.02), b=(0.0, -0.006, -0.17), radius=0.035),     # upper arm
        dict(link=-1, a=base_p + root_r @ np.array([0.0, 0.0, 0.05]),
             b=base_p + root_r @ np.array([0.0, 0.0, 0.30]), radius=0.09),           # torso
        dict(link=-1, a=root_p + root_r @ np.array([0.0, 0.0, -0.02]),
             b=root_p + root_r @ np.array([0.0, 0.0, -0.02]), radius=0.09),          # pelvis
        dict(link=-1, a=base_p + root_r @ np.array([0.0, 0.0, 0.45]),
             b=base_p + root_r @ np.array([0.0, 0.0, 0.45]), radius=0.07),           # head
    ]
    c.num_shapes = len(shapes)
    for k, s in enumerate(shapes):
        sh = c.shape[k]
        sh.link = s["link"]
        _set(sh.a, s["a"])
        _set(sh.b, s["b"])
        sh.radius = s["radius"]
        sh.restitution, sh.friction = hum_e, hum_mu
    _set(c.humanoid_bound_center, shoulder_w)
    c.humanoid_bound_radius = 0.95

    c.serve_speed_lo, c.serve_speed_hi = scene["serve_speed"]
    c.serve_tilt_lo_deg, c.serve_tilt_hi_deg = scene["serve_tilt"]
    c.serve_tilt_z_lo_deg, c.serve_tilt_z_hi_deg = scene["serve_tilt_z"]

    c.max_episode_length = int(env["episodeLength"])
    c.alpha_velocity_reward = env["alphaVelocityReward"]
    c.power_coefficient = env["powerCoefficient"]
    c.penalty = env["penalty"]
    c.hit_table_reward = env["hitTableReward"]
    c.not_hit_table_penalty = env["nothitTablePenalty"]

    # --- second humanoid (T4): the same arm model on another base
    c.num_humanoids = 1
    if variant == "T4":
        c.num_humanoids = 2
        root2_p = np.asarray(scene["humanoid2_pos"], dtype=np.float64)
        root2_q = np.asarray(scene["humanoid2_quat"], dtype=np.float64)
        root2_q = root2_q / np.linalg.norm(root2_q)
        root2_r = quat_to_rot(root2_q)
        base2_p = root2_p + root2_r @ np.asarray(PELVIS_TO_TORSO)
        _set(c.base2_pos, base2_p)
        _set(c.base2_rot, root2_r.reshape(-1))
        _set(c.humanoid2_root_pos, root2_p)
        _set(c.humanoid2_root_quat, root2_q)
        statics = {3: (base2_p + root2_r @ np.array([0.0, 0.0, 0.05]), base2_p + root2_r @ np.array([0.0, 0.0, 0.30])),
                   4: (root2_p + root2_r @ np.array([0.0, 0.0, -0.02]),) * 2,
                   5: (base2_p + root2_r @ np.array([0.0, 0.0, 0.45]),) * 2}
        for k in range(c.num_shapes):
            src, dst = c.shape[k], c.shape2[k]
            dst.link, dst.radius, dst.restitution, dst.friction = src.link, src.radius, src.restitution, src.friction
            if src.link >= 0:
                _set(dst.a, list(src.a))
                _set(dst.b, list(src.b))
            else:
                _set(dst.a, statics[k][0])
                _set(dst.b, statics[k][1])
        _set(c.humanoid2_bound_center, base2_p + root2_r @ np.asarray(G1_RIGHT_ARM[0]["xyz"]))
    return c


def pd_action_offset_scale(config):
    """`_pd_action_offset`, `_pd_action_scale` = 1/2 (hi +- lo) of the dof limits (TT:649-671)."""
    lo = np.array([config.joint[j].lower for j in range(NUM_DOF)], dtype=np.float32)
    hi = np.array([config.joint[j].upper for j in range(NUM_DOF)], dtype=np.float32)
    return 0.5 * (hi + lo), 0.5 * (hi - lo)


def initial_root_states(config):
    """[3, 13] initial actor root states (humanoid, table, ball) in the reference's layout (TT:173-183)."""
    out = np.zeros((NUM_ACTORS, 13), dtype=np.float32)
    out[0, 0:3] = list(config.humanoid_root_pos)
    out[0, 3:7] = list(config.humanoid_root_quat)
    out[1, 0:3] = list(config.table_root_pos)
    out[1, 3:7] = list(config.table_root_quat)
    out[2, 0:3] = list(config.ball_init_pos)
    out[2, 3:7] = list(config.ball_init_quat)
    return out
