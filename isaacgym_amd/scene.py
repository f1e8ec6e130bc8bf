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
ABI_VERSION = 3
STATUS_HANDOFF_TIMEOUT = 1       # include/ppenv.h PPENV_STATUS_*
SERVE_ANGLE_LIMIT_DEG = 57.0   # csrc/ppenv_device.h sincos_small; ppenv_create refuses wider ranges too

VARIANT_T3, VARIANT_TT, VARIANT_TN, VARIANT_T4 = 0, 1, 2, 3
VARIANT_IDS = {"T3": VARIANT_T3, "TT": VARIANT_TT, "TN": VARIANT_TN, "T4": VARIANT_T4}

FLAG_REWARD_CALC, FLAG_COND_CALC, FLAG_NO_BOUNCE, FLAG_MISSED_CALC = 1, 2, 4, 8
# 27-dof task (include/ppenv.h PPENV_TA_*): sticky flags TA:279-286, diagnostic count flags TA:289-293
TA_FLAG_PADDLE_COND, TA_FLAG_HIT_TABLE_CALC, TA_FLAG_DIE_PENALTY_CALC, TA_FLAG_HUMANOID_DIE_CALC = 1, 2, 4, 8
TA_COUNT_MASK = 0x1F0

# reference task name (tasks/__init__.py:49-53,118-120) -> variant tag
TASK_VARIANTS = {
    "HumanoidPingpongG1": "T3",
    "HumanoidPingpongTiltG1": "TT",
    "HumanoidPingpongTiltNoEarlyStopG1": "TN",
    "Humanoid12PingpongTiltG1": "T4",
    "HumanoidPingpongTiltNESSparse27DOFG1": "TA",
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


class Randomization(C.Structure):
    """ctypes mirror of ppenv_randomization (include/ppenv.h): device pointers of the per-env tables + the two noise amplitudes."""
    _fields_ = [("dof_stiffness_scale", C.c_void_p), ("dof_damping_scale", C.c_void_p), ("link_mass_scale", C.c_void_p),
                ("restitution_scale", C.c_void_p), ("friction_scale", C.c_void_p), ("action_noise_sigma", C.c_float),
                ("observation_noise_sigma", C.c_float)]


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
        ("initial_rb_shared", C.c_int32), ("pad_", C.c_int32),
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


def _table_geom(table):
    """TABLE_GEOM overridden by `table` (isaacgym_amd.urdf.table_scene of a table asset, or a hand-written dict of the same keys)."""
    g = dict(TABLE_GEOM, offset_xy=(0.0, 0.0), net_offset_xy=None, net_bottom_z=None)
    if table is not None:
        unknown = sorted(set(table) - set(g) - {"ignored"})
        if unknown:
            raise ValueError(f"table geometry has unknown keys {unknown}")
        g.update({k: v for k, v in table.items() if k != "ignored"})
    if g["net_offset_xy"] is None:
        g["net_offset_xy"] = g["offset_xy"]
    if g["net_bottom_z"] is None:
        g["net_bottom_z"] = g["top_z"]
    for k in ("length", "width", "slab", "net_height", "net_half_thickness"):
        if not g[k] > 0.0:
            raise ValueError(f"table geometry: {k} must be positive")
    return g


def _ball_geom(ball):
    """BALL_GEOM overridden by `ball` (isaacgym_amd.urdf.ball_params of a ball asset, or a dict of the same keys)."""
    g = dict(BALL_GEOM)
    if ball is not None:
        unknown = sorted(set(ball) - set(g))
        if unknown:
            raise ValueError(f"ball parameters have unknown keys {unknown}")
        g.update(ball)
    if not (g["radius"] > 0.0 and g["mass"] > 0.0 and g["inertia_factor"] > 0.0 and g["angular_damping"] >= 0.0):
        raise ValueError("ball parameters: radius, mass and inertia_factor must be positive")
    return g


def asset_geometry(scene_cfg):
    """(table, ball) for build_config from the optional asset entries of a task cfg's `scene` block: `table_urdf` / `ball_urdf` name the
    files the reference loads as pingpong_table.urdf / small_ball.urdf (TT:496,502 — absolute paths on its author's machine, not
    yaml keys, hence entries of this build's own block); `table_geom` / `ball_geom` give the same dicts directly.  Absent: (None, None),
    i.e. the UNVERIFIED placeholders."""
    from . import urdf
    table, ball = scene_cfg.get("table_geom"), scene_cfg.get("ball_geom")
    if scene_cfg.get("table_urdf"):
        table = urdf.table_scene(urdf.load(scene_cfg["table_urdf"]))
    if scene_cfg.get("ball_urdf"):
        ball = urdf.ball_params(urdf.load(scene_cfg["ball_urdf"]))
    return table, ball


def build_config(variant, cfg=None, num_envs=None, seed=0, device_id=0, env_id_offset=0, ball_substeps=4,
                 restitution_max=1.0, table=None, ball=None):
    """Build the C config for variant 'T3' | 'TT' | 'TN' | 'T4' from a task cfg dict (default: the yaml defaults).
    table / ball: the geometry of the other two assets of the scene (TT:496,502) when it comes from their URDFs
    (isaacgym_amd.urdf.table_scene / ball_params) instead of the UNVERIFIED placeholders TABLE_GEOM / BALL_GEOM; both are run-time
    constants of the step kernels, so no rebuild is involved."""
    if cfg is None:
        cfg = default_task_cfg(variant)
    env, sim, scene = cfg["env"], cfg["sim"], cfg["scene"]
    bg = _ball_geom(ball)
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
        parts = [(spec["mass"], spec["com"], _spec_inertia(spec["inertia"]), np.eye(3))]
        if j == NUM_DOF - 1:
            hand_o = np.asarray(G1_HAND["xyz"])
            parts.append((G1_HAND["mass"], hand_o + np.asarray(G1_HAND["com"]), _spec_inertia(G1_HAND["inertia"]), np.eye(3)))
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
    c.ball_radius = bg["radius"]
    c.ball_mass = bg["mass"]
    c.ball_inertia_factor = bg["inertia_factor"]
    c.ball_restitution = ball_mat["restitution"]
    c.ball_friction = ball_mat["friction"]
    c.ball_angular_damping = bg["angular_damping"]
    c.restitution_max = restitution_max

    plane = env["plane"]
    c.ground_z = 0.0
    c.ground_restitution, c.ground_friction = _combine(
        ball_mat, dict(restitution=plane["restitution"], friction=plane["dynamicFriction"]), restitution_max)

    tp = np.asarray(scene["table_pos"], dtype=np.float64)
    g = _table_geom(table)
    _set(c.table.center, (tp[0] + g["offset_xy"][0], tp[1] + g["offset_xy"][1], tp[2] + g["top_z"] - 0.5 * g["slab"]))
    _set(c.table.half, (0.5 * g["length"], 0.5 * g["width"], 0.5 * g["slab"]))
    c.table.restitution, c.table.friction = _combine(ball_mat, scene["table_material"], restitution_max)
    _set(c.net.center, (tp[0] + g["net_offset_xy"][0], tp[1] + g["net_offset_xy"][1], tp[2] + g["net_bottom_z"] + 0.5 * g["net_height"]))
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
    for name in ("serve_tilt", "serve_tilt_z"):   # the kernels' sine / cosine of a serve angle is a short series, fp32-exact up to ~1 rad
        if max(abs(float(v)) for v in scene[name]) > SERVE_ANGLE_LIMIT_DEG:
            raise ValueError(f"scene.{name} = {tuple(scene[name])}: serve angles must lie within +-{SERVE_ANGLE_LIMIT_DEG} degrees")

    c.max_episode_length = int(env["episodeLength"])
    c.alpha_velocity_reward = env["alphaVelocityReward"]
    c.power_coefficient = env["powerCoefficient"]
    c.penalty = env["penalty"]
    if variant == "T3":   # T3's reward has no table terms (T3:1080-1173); its yaml has no such keys
        env = dict(env, hitTableReward=env.get("hitTableReward", 0.0), nothitTablePenalty=env.get("nothitTablePenalty", 0.0))
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


# =====================================================================================================================
# 27-DoF variant (TA = tasks/humanoid_pingpong_3_actor_all_dof.py): model of the free-floating humanoid
# =====================================================================================================================
# UNVERIFIED physical data, like the arm tables above: g1_27dof.urdf (TA:470) is not in the reference; the numbers are
# recalled from the public Unitree g1_29dof_rev_1_0.urdf (the 27-dof asset welds right_shoulder_yaw and right_elbow,
# TA:1303-1311 lists 5 right-arm dofs).  Dof order = TA:1303-1311; rigid-body indices = pingpong_note.txt:33.
TA_NUM_LINKS, TA_NUM_FIXED, TA_MAX_CONTACTS = 28, 12, 24


def _leg(side):
    s = 1.0 if side == "left" else -1.0
    b0 = 2 if side == "left" else 9
    ms = lambda lo, hi: (lo, hi) if side == "left" else (-hi, -lo)   # mirrored x / z axes flip the limit interval
    return [
        dict(name=f"{side}_hip_pitch_link", body=b0, xyz=(0.0, s * 0.064452, -0.1027), rpy=(0, 0, 0), axis=1, limits=(-2.5307, 2.8798),
             mass=1.35, com=(0.002741, s * 0.047791, -0.02606), inertia=(0.00182, 0.00153, 0.00116), effort=88.0, vel=32.0),
        dict(name=f"{side}_hip_roll_link", body=b0 + 1, xyz=(0.0, s * 0.052, -0.030465), rpy=(0, -0.1749, 0), axis=0, limits=ms(-0.5236, 2.9671),
             mass=1.52, com=(0.029812, s * -0.001045, -0.087934), inertia=(0.00254, 0.00241, 0.00148), effort=88.0, vel=32.0),
        dict(name=f"{side}_hip_yaw_link", body=b0 + 2, xyz=(0.025001, 0.0, -0.12412), rpy=(0, 0, 0), axis=2, limits=ms(-2.7576, 2.7576),
             mass=1.702, com=(-0.057709, s * -0.010981, -0.15078), inertia=(0.00776, 0.00717, 0.00160), effort=88.0, vel=32.0),
        dict(name=f"{side}_knee_link", body=b0 + 3, xyz=(-0.078273, s * 0.0021489, -0.17734), rpy=(0, 0.1749, 0), axis=1, limits=(-0.087267, 2.8798),
             mass=1.932, com=(0.005457, s * 0.003964, -0.12074), inertia=(0.01139, 0.01127, 0.00146), effort=139.0, vel=20.0),
        dict(name=f"{side}_ankle_pitch_link", body=b0 + 4, xyz=(0.0, s * -9.4445e-05, -0.30001), rpy=(0, 0, 0), axis=1, limits=(-0.87267, 0.5236),
             mass=0.074, com=(-0.007269, 0.0, 0.011137), inertia=(8.4e-6, 1.89e-5, 1.26e-5), effort=50.0, vel=37.0),
        dict(name=f"{side}_ankle_roll_link", body=b0 + 5, xyz=(0.0, 0.0, -0.017558), rpy=(0, 0, 0), axis=0, limits=ms(-0.2618, 0.2618),
             mass=0.608, com=(0.026505, 0.0, -0.016425), inertia=(0.00022, 0.00161, 0.00167), effort=50.0, vel=37.0),
    ]


G1_WAIST = [
    dict(name="waist_yaw_link", body=15, xyz=(0.0, 0.0, 0.0), rpy=(0, 0, 0), axis=2, limits=(-2.618, 2.618),
         mass=0.214, com=(0.003494, 0.000233, 0.018034), inertia=(1.6e-4, 1.2e-4, 1.9e-4), effort=88.0, vel=32.0),
    dict(name="waist_roll_link", body=16, xyz=PELVIS_TO_TORSO, rpy=(0, 0, 0), axis=0, limits=(-0.52, 0.52),
         mass=0.086, com=(0.0, 0.0, 0.0), inertia=(7.0e-6, 6.9e-6, 3.0e-6), effort=50.0, vel=37.0),
    dict(name="torso_link", body=17, xyz=(0.0, 0.0, 0.0), rpy=(0, 0, 0), axis=1, limits=(-0.52, 0.52),
         mass=7.818, com=(0.00203158, 0.000339683, 0.184568), inertia=(0.1216, 0.1127, 0.0327), effort=50.0, vel=37.0),
]
G1_PELVIS = dict(name="pelvis", body=0, mass=3.813, com=(0.0, 0.0, -0.07605), inertia=(0.0106, 0.0093, 0.0080))
# welded to the torso (bodies 18, 19, 20, 29, 30): only the head and the lidar carry noticeable mass
G1_TORSO_WELDED = [
    dict(name="d435_link", body=18, xyz=(0.0576235, 0.01753, 0.42987), mass=0.0),
    dict(name="head_link", body=19, xyz=(0.0039635, 0.0, 0.40), mass=1.036, com=(0.005, 0.0, 0.02), inertia=(0.0045, 0.0048, 0.0030)),
    dict(name="imu_in_torso", body=20, xyz=(-0.03959, -0.00224, 0.14792), mass=0.0),
    dict(name="logo_link", body=29, xyz=(0.0039635, 0.0, 0.10), mass=0.0),
    dict(name="mid360_link", body=30, xyz=(0.0002835, 0.00003, 0.41618), mass=0.265, com=(0.0, 0.0, 0.02), inertia=(1.2e-4, 1.2e-4, 1.6e-4)),
]
G1_PELVIS_WELDED = [dict(name="imu_in_pelvis", body=1, xyz=(0.04525, 0.0, -0.08339), mass=0.0),
                    dict(name="pelvis_contour_link", body=8, xyz=(0.0, 0.0, 0.0), mass=0.001, com=(0.0, 0.0, 0.0), inertia=(1e-7, 1e-7, 1e-7))]
# 27 gains, TA:757-772 (the right ankle-pitch 80 vs the left 20 is as written there); Kd = Kp / 40, TA:774
TA_P_GAINS = [80.0, 80.0, 80.0, 160.0, 20.0, 20.0, 80.0, 80.0, 80.0, 160.0, 80.0, 20.0, 80.0, 80.0, 80.0,
              20.0, 20.0, 20.0, 20.0, 20.0, 5.0, 5.0, 20.0, 20.0, 20.0, 5.0, 5.0]
# sole corners in the ankle-roll frame; joint armature (reflected rotor inertia) — named parameters of
# the build's physics specification, see DESIGN.md "TA physics"
TA_FOOT = dict(points=[(-0.05, -0.03, -0.035), (-0.05, 0.03, -0.035), (0.12, -0.03, -0.035), (0.12, 0.03, -0.035)],
               stiffness=1.0e5, damping=1.0e3, tangent_damping=2.0e3, max_penetration=0.02,
               fade_depth=2.0e-3, fade_force=20.0)
# further (link, point) pairs tested against the ground, so that a fallen humanoid lies on the plane instead of sinking
# through it: knees, pelvis, chest / back, head top, shoulders, elbows, hands (link indices of the 28-link tree)
TA_BODY_CONTACTS = [(4, (0.03, 0.0, 0.0)), (10, (0.03, 0.0, 0.0)), (0, (0.07, 0.0, -0.08)), (0, (-0.09, 0.0, -0.08)),
                    (15, (0.10, 0.0, 0.25)), (15, (-0.10, 0.0, 0.25)), (15, (0.0, 0.0, 0.50)),
                    (16, (0.0, 0.05, 0.0)), (23, (0.0, -0.05, 0.0)), (19, (0.0, 0.0, -0.02)), (24, (0.0158, -0.0062, -0.1837)),
                    (22, (0.11, 0.0, 0.0)), (27, (0.11, 0.0, 0.0))]
TA_ARMATURE = 0.01            # every joint: without it a saturated drive spins a 50-gram wrist link up by 10^3 rad/s in one substep
TA_JOINT_LIMITS = dict(stiffness=2000.0, damping=20.0, vel_damping=50.0)   # N m / rad, N m s / rad, N m s / rad
# PlaneParams.distance = -0.21 (TA:403) with normal +z is the plane z = 0.21: the pelvis is created at z = 1.0 (TA:578)
# and the G1's soles are 0.787 m below it with straight legs, i.e. at z = 0.213 — the robot is spawned standing on it.
TA_GROUND_Z = 0.21


def _mirror_arm(spec):
    """left-arm link from the right-arm table: reflect through the xz-plane."""
    d = dict(spec)
    d["name"] = spec["name"].replace("right", "left")
    d["body"] = spec["body"] - 10
    x, y, z = spec["xyz"]
    d["xyz"] = (x, -y, z)
    r, p, yw = spec["rpy"]
    d["rpy"] = (-r, p, -yw)
    cx, cy, cz = spec["com"]
    d["com"] = (cx, -cy, cz)
    lo, hi = spec["limits"]
    d["limits"] = (lo, hi) if spec["axis"] == 1 else (-hi, -lo)
    return d


def _spec_inertia(v):
    """inertia entry of a spec: the diagonal (xx, yy, zz) or the 6-vector xx yy zz xy xz yz -> 3x3"""
    return np.diag(v) if len(v) == 3 else _inertia_mat(v)


def _diag_parts(spec, offset=(0.0, 0.0, 0.0)):
    return (spec["mass"], np.asarray(offset) + np.asarray(spec.get("com", (0, 0, 0))), _spec_inertia(spec.get("inertia", (0, 0, 0))), np.eye(3))


class TALink(C.Structure):
    _fields_ = [("parent", C.c_int32), ("axis", C.c_int32), ("body", C.c_int32), ("origin_xyz", C.c_float * 3), ("origin_rot", C.c_float * 9),
                ("mass", C.c_float), ("com", C.c_float * 3), ("inertia", C.c_float * 6), ("lower", C.c_float), ("upper", C.c_float),
                ("kp", C.c_float), ("kd", C.c_float), ("effort", C.c_float), ("vel_limit", C.c_float), ("armature", C.c_float)]


class TAFixed(C.Structure):
    _fields_ = [("body", C.c_int32), ("link", C.c_int32), ("xyz", C.c_float * 3), ("rot", C.c_float * 9)]


class TAModel(C.Structure):
    _fields_ = [("link", TALink * TA_NUM_LINKS), ("fixed", TAFixed * TA_NUM_FIXED), ("num_contacts", C.c_int32),
                ("contact_link", C.c_int32 * TA_MAX_CONTACTS), ("contact_point", (C.c_float * 3) * TA_MAX_CONTACTS),
                ("ground_z", C.c_float), ("foot_stiffness", C.c_float),
                ("foot_damping", C.c_float), ("foot_tangent_damping", C.c_float), ("foot_friction", C.c_float),
                ("contact_fade_depth", C.c_float), ("contact_fade_force", C.c_float), ("contact_max_penetration", C.c_float), ("limit_stiffness", C.c_float), ("limit_damping", C.c_float),
                ("vel_limit_damping", C.c_float), ("bound_link", C.c_int32), ("bound_center", C.c_float * 3)]


def build_ta_model():
    """ppenv_ta_model: the 28-link tree (pelvis + 27 dofs in the order of TA:1303-1311) with welded bodies merged in."""
    m = TAModel()
    right_arm = G1_RIGHT_ARM
    left_arm = [_mirror_arm(s) for s in right_arm]
    hand_r, hand_l = G1_HAND, dict(G1_HAND, xyz=(G1_HAND["xyz"][0], -G1_HAND["xyz"][1], G1_HAND["xyz"][2]),
                                   com=(G1_HAND["com"][0], -G1_HAND["com"][1], G1_HAND["com"][2]))
    # (spec, parent link, welded parts [(mass, com, I, R)] in the link frame, origin override)
    links = [(None, -1, [], None)]
    for s in _leg("left"):
        links.append((s, len(links) - 1 if len(links) > 1 else 0, [], None))
    first_right = len(links)
    for k, s in enumerate(_leg("right")):
        links.append((s, 0 if k == 0 else len(links) - 1, [], None))
    for k, s in enumerate(G1_WAIST):
        welded = [_diag_parts(w, w["xyz"]) for w in G1_TORSO_WELDED if w["mass"] > 0] if s["name"] == "torso_link" else []
        links.append((s, 0 if k == 0 else len(links) - 1, welded, None))
    torso = len(links) - 1
    for k, s in enumerate(left_arm):
        welded = [_diag_parts(hand_l, hand_l["xyz"])] if k == 6 else []
        links.append((s, torso if k == 0 else len(links) - 1, welded, None))
    # right arm: shoulder pitch, shoulder roll (+ welded shoulder-yaw and elbow links), wrist roll / pitch / yaw (+ hand, paddle)
    yaw_o = np.asarray(right_arm[2]["xyz"])
    elbow_o = yaw_o + np.asarray(right_arm[3]["xyz"])
    wrist_o = elbow_o + np.asarray(right_arm[4]["xyz"])
    links.append((right_arm[0], torso, [], None))
    links.append((right_arm[1], len(links) - 1, [_diag_parts(right_arm[2], yaw_o), _diag_parts(right_arm[3], elbow_o)], None))
    links.append((right_arm[4], len(links) - 1, [], wrist_o))
    links.append((right_arm[5], len(links) - 1, [], None))
    hand_o = np.asarray(hand_r["xyz"])
    pad_o = hand_o + np.asarray(PADDLE["xyz_from_hand"])
    pm, pr, pn = PADDLE["mass"], PADDLE["radius"], np.asarray(PADDLE["normal"], dtype=np.float64)
    i_disc = 0.25 * pm * pr * pr * np.eye(3) + 0.25 * pm * pr * pr * np.outer(pn, pn)
    links.append((right_arm[6], len(links) - 1, [_diag_parts(hand_r, hand_o), (pm, pad_o, i_disc, np.eye(3))], None))
    assert len(links) == TA_NUM_LINKS and first_right == 7 and torso == 15

    for i, (spec, parent, welded, origin) in enumerate(links):
        L = m.link[i]
        L.parent = parent
        if spec is None:   # pelvis
            parts = [_diag_parts(G1_PELVIS)] + [_diag_parts(w, w["xyz"]) for w in G1_PELVIS_WELDED if w["mass"] > 0]
            L.axis, L.body = 0, 0
            _set(L.origin_xyz, (0, 0, 0))
            _set(L.origin_rot, np.eye(3).reshape(-1))
            L.lower = L.upper = L.kp = L.kd = L.effort = L.vel_limit = L.armature = 0.0
        else:
            parts = [_diag_parts(spec)] + welded
            L.axis, L.body = spec["axis"], spec["body"]
            _set(L.origin_xyz, origin if origin is not None else spec["xyz"])
            _set(L.origin_rot, rpy_to_rot(*spec["rpy"]).reshape(-1))
            lo, hi = spec["limits"]
            L.lower, L.upper = min(lo, hi), max(lo, hi)              # TA:719-725 swaps inverted limits
            L.kp = TA_P_GAINS[i - 1]
            L.kd = TA_P_GAINS[i - 1] / 40.0                          # TA:774
            L.effort, L.vel_limit = spec["effort"], spec["vel"]
            L.armature = TA_ARMATURE
        mass, com, inertia = composite_inertial(parts)
        L.mass = mass
        _set(L.com, com)
        _set(L.inertia, _inertia_vec(inertia))

    fixed = [(w["body"], 0, w["xyz"]) for w in G1_PELVIS_WELDED] + [(w["body"], torso, w["xyz"]) for w in G1_TORSO_WELDED] + \
            [(28, 22, hand_l["xyz"]), (33, 24, yaw_o), (34, 24, elbow_o), (38, 27, hand_o), (39, 27, pad_o)]
    assert len(fixed) == TA_NUM_FIXED
    for k, (body, link, xyz) in enumerate(sorted(fixed)):
        f = m.fixed[k]
        f.body, f.link = body, link
        _set(f.xyz, xyz)
        _set(f.rot, np.eye(3).reshape(-1))
    fill_ta_contacts_and_limits(m)
    return m


def fill_ta_contacts_and_limits(m, contacts=None, bound=None):
    """The parts of ppenv_ta_model that are not in a URDF: ground-contact points, contact / limit parameters of the physics
    specification, the ball's broad-phase sphere.  contacts: [(link, point)], bound: (link, point); defaults are the G1 tree's."""
    contacts = ([(6, p) for p in TA_FOOT["points"]] + [(12, p) for p in TA_FOOT["points"]] + TA_BODY_CONTACTS) if contacts is None else list(contacts)
    assert len(contacts) <= TA_MAX_CONTACTS
    m.num_contacts = len(contacts)
    for k, (link, pnt) in enumerate(contacts):
        m.contact_link[k] = link
        _set(m.contact_point[k], pnt)
    m.ground_z = TA_GROUND_Z
    m.foot_stiffness, m.foot_damping, m.foot_tangent_damping = TA_FOOT["stiffness"], TA_FOOT["damping"], TA_FOOT["tangent_damping"]
    m.contact_max_penetration = TA_FOOT["max_penetration"]
    m.contact_fade_depth, m.contact_fade_force = TA_FOOT["fade_depth"], TA_FOOT["fade_force"]
    m.limit_stiffness, m.limit_damping, m.vel_limit_damping = TA_JOINT_LIMITS["stiffness"], TA_JOINT_LIMITS["damping"], TA_JOINT_LIMITS["vel_damping"]
    m.foot_friction = 0.5 * (1.0 + 0.5)      # PhysX average of plane 1.0 (TA yaml:78) and humanoid shapes 0.5 (TA:588)
    link, pnt = (15, G1_RIGHT_ARM[0]["xyz"]) if bound is None else bound
    m.bound_link = link
    _set(m.bound_center, pnt)
    return m


def use_arm_tables(arm_specs, hand=None, paddle=None):
    """Replace the placeholder 7-dof chain tables (e.g. by isaacgym_amd.urdf.arm_specs of the real asset).  The compiled-in
    model must then be regenerated (`python -m isaacgym_amd.modelgen`) and the library rebuilt: ppenv_create refuses a
    config whose model differs from the compiled one."""
    global G1_RIGHT_ARM, G1_HAND, PADDLE
    if len(arm_specs) != NUM_DOF:
        raise ValueError(f"the arm chain has {NUM_DOF} dofs")
    G1_RIGHT_ARM = [dict(s) for s in arm_specs]
    if hand is not None:
        G1_HAND = dict(hand)
    if paddle is not None:
        PADDLE = dict(paddle)


TASK_CFGS["TA"] = dict(
    name="HumanoidPingpongTiltNESSparse27DOFG1",
    env=dict(TASK_CFGS["TN"]["env"], numEnvs=2048, episodeLength=160, alphaVelocityReward=3000.0, powerCoefficient=0.002,      # yaml:8-17
             hitTableReward=3000.0, nothitTablePenalty=-1000.0, crossNetRewardFloat=1000.0, diePenaltyFloat=-3000.0,            # yaml:21-30
             hitPaddleReward=200.0, missPaddlePenaltyCoefficient=-100.0),
    sim=dict(_SIM_DEFAULT),                                                                 # 27DOFG1.yaml:88-90: dt 0.0083, substeps 2
    scene=dict(TASK_CFGS["TN"]["scene"],
               table_material=dict(restitution=1.5, friction=0.2),                         # TA:636-638
               ball_pos=(2.9, -0.2, 1.0), ball_material=dict(restitution=1.5, friction=0.2),   # TA:678-680,686-687
               serve_speed=(5.0, 5.4), serve_tilt=(-8.0, 3.0), serve_tilt_z=(14.0, 24.0)),  # TA:129-131
)


TA_GRAVITY_Z = -9.8      # TA:383-385 (the task overrides its yaml's -9.81 as the 7-dof tasks do)


def build_ta_scene(num_envs, device_id=0, table=None, ball=None):
    """The ppenv_config part of the 27-DoF scene (ball, table, net, contact scalars, dt ...) with the humanoid's ball-collision
    shapes re-attached to links of the 28-link tree (ppenv_ta_model).  table / ball: as in build_config (TA:551,557)."""
    c = build_config("TN", cfg=default_task_cfg("TA"), num_envs=num_envs, device_id=device_id, table=table, ball=ball)
    ra = G1_RIGHT_ARM
    yaw_o = np.asarray(ra[2]["xyz"])
    elbow_o = yaw_o + np.asarray(ra[3]["xyz"])
    c.ground_z = TA_GROUND_Z                       # the ball lands on the same plane as the feet
    c.paddle_link = 27
    re_attach = {6: (27, np.zeros(3)), 3: (24, elbow_o), 1: (24, np.zeros(3))}   # 7-dof chain link -> (tree link, frame offset)
    statics = {3: (15, (0.0, 0.0, 0.05), (0.0, 0.0, 0.30)), 4: (0, (0.0, 0.0, -0.02), (0.0, 0.0, -0.02)), 5: (15, (0.0, 0.0, 0.45), (0.0, 0.0, 0.45))}
    for k in range(c.num_shapes):
        sh = c.shape[k]
        if sh.link >= 0:
            link, off = re_attach[sh.link]
            _set(sh.a, np.asarray(list(sh.a)) + off)
            _set(sh.b, np.asarray(list(sh.b)) + off)
            sh.link = link
        else:   # torso / pelvis / head: static in the fixed-base scenes, on their links here
            sh.link = statics[k][0]
            _set(sh.a, statics[k][1])
            _set(sh.b, statics[k][2])
    return c


def _mix64(z):
    m = 0xFFFFFFFFFFFFFFFF
    z &= m
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & m
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return z


def ta_reset_draws(params, env_ids, episodes):
    """Host restatement of the 27-DoF task's reset draws (ta_post_physics_kernel, TA:976-979 + 346-377): for each (env,
    episode) the five values ball y, ball z, vx, vy, vz.  Used to lay out the state at creation (episode 0); resets during
    stepping are drawn on the device by the same keyed function."""
    import torch
    m = 0xFFFFFFFFFFFFFFFF
    out = torch.zeros(len(env_ids), 5, dtype=torch.float32)
    f32 = np.float32
    for row, (i, ep) in enumerate(zip([int(x) for x in env_ids], [int(x) for x in episodes])):
        gid = params.env_id_offset + i
        u = []
        for k in range(5):
            s = _mix64(params.seed + 0x9E3779B97F4A7C15 * (gid + 1))
            x = _mix64(s + 0x9E3779B97F4A7C15 * (ep * 8 + k + 1))
            u.append(f32(x >> 40) * f32(1.0 / 16777216.0))
        deg = f32(0.017453292519943295)
        y = f32(params.ball_y_lo) + f32(params.ball_y_hi - params.ball_y_lo) * u[0]
        z = f32(params.ball_z_lo) + f32(params.ball_z_hi - params.ball_z_lo) * u[1]
        speed = f32(params.serve_speed_lo) + f32(params.serve_speed_hi - params.serve_speed_lo) * u[2]
        a = (f32(params.serve_tilt_lo_deg) + f32(params.serve_tilt_hi_deg - params.serve_tilt_lo_deg) * u[3]) * deg
        az = (f32(params.serve_tilt_z_lo_deg) + f32(params.serve_tilt_z_hi_deg - params.serve_tilt_z_lo_deg) * u[4]) * deg
        out[row] = torch.tensor([y, z, -speed * math.cos(a) * math.cos(az), speed * math.sin(a) * math.cos(az), speed * math.sin(az)])
    return out
