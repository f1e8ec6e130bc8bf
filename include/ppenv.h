/*
 * ppenv.h — C ABI of the MI355X-native vectorised HumanoidPingpong environment.
 *
 * This is the drop-in boundary for the one hot path of mjmj531/isaacgym that
 * BASELINE.json names: the VecTask step of the HumanoidPingpong tasks
 *     pre_physics_step -> gym.simulate -> post_physics_step
 * (reference: tasks/humanoid_pingpong_3_actor_tilt.py:1002-1052 "TT", and the
 * same hooks in the T3 / TN variants).  The reference drives that path through
 * two foreign interfaces, neither of which lives in the reference repo:
 *   - upward, isaacgymenvs' VecTask buffer surface (obs_buf / rew_buf /
 *     reset_buf / progress_buf, TT:118,1023-1037), and
 *   - downward, Isaac Gym's gymapi/gymtorch tensor API (acquire_*_tensor,
 *     refresh_*, set_*_indexed, simulate; TT:131-134,801-807,881-888,1014).
 * Each entry point below names the reference call it replaces.
 *
 * Conventions: plain C, no torch types.  Every function returns 0 on success
 * or a negative PPENV_E* code and never throws; ppenv_last_error() gives a
 * thread-local message.  All pointers named *_dev are device (HBM) pointers.
 * Launches are enqueued on the caller's HIP stream (`stream`, a hipStream_t
 * passed as void*; NULL = the null stream) and never synchronise.  A handle is
 * bound to one GPU and is not re-entrant.
 *
 * Quaternions are xyzw everywhere (Isaac Gym layout, TT:173-183).
 */
#ifndef PPENV_H
#define PPENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPENV_ABI_VERSION 3

struct ppenv; /* opaque handle */
typedef struct ppenv ppenv;

#define PPENV_NUM_DOF 7          /* right-arm chain (TT:427-431) */
#define PPENV_NUM_OBS_BODIES 10  /* bodyStatesId, cfg/task/HumanoidPingpongTiltG1.yaml:47 */
#define PPENV_NUM_OBS 80         /* TT:98 : 30 + 30 + 7 + 7 + 3 + 3 */
#define PPENV_NUM_HUMANOID_BODIES 40 /* tasks/pingpong_note.txt:33 */
#define PPENV_NUM_BODIES 42      /* humanoid 40 + table + ball, TT:127 */
#define PPENV_NUM_ACTORS 3       /* humanoid, table, ball, TT:125 */
#define PPENV_MAX_SHAPES 8       /* capsule/sphere collision shapes on the humanoid */

/* error codes */
#define PPENV_OK 0
#define PPENV_EINVAL (-1)   /* bad argument / inconsistent config */
#define PPENV_ENOMEM (-2)   /* allocation failed */
#define PPENV_EHIP (-3)     /* a HIP runtime call failed */
#define PPENV_ESTATE (-4)   /* blob size / version mismatch in get/set_state */
#define PPENV_EDEVICE (-5)  /* a step kernel reported a fault (see ppenv_status): the handle's state is no longer valid */

/* Bits of the per-handle device status word (ppenv_status).  Zero = healthy. */
#define PPENV_STATUS_HANDOFF_TIMEOUT 1u  /* a wave of the multi-wave step kernel gave up waiting for its partner wave's
                                            LDS hand-off; that workgroup's envs were NOT stored for that step */

/* Task variants (reward / reset semantics). */
enum {
    PPENV_VARIANT_T3 = 0, /* HumanoidPingpongG1: tasks/humanoid_interos_edit_pingpong_only_3_actor.py */
    PPENV_VARIANT_TT = 1, /* HumanoidPingpongTiltG1: tasks/humanoid_pingpong_3_actor_tilt.py */
    PPENV_VARIANT_TN = 2, /* HumanoidPingpongTiltNoEarlyStopG1: ..._tilt_no_earlystop.py */
    PPENV_VARIANT_T4 = 3  /* Humanoid12PingpongTiltG1: tasks/humanoid_pingpong_4_actor_tilt.py — two humanoids, one ball.
                             The reference class is unfinished (T4:743,786-803); the two-agent wiring below is the build's
                             completion of it: agent a of env e owns row 2e + a of obs / rew / reset /
                             progress and of the [2N, 7] action tensor (the rl_games multi-agent convention), humanoid 2
                             stands at x = 3.5 facing humanoid 1 (T4:555-556), sticky flags per side, TT's serve and reset. */
};

/* Sticky per-env flag bits (one uint32 per env).
 * TT: reward_calculated / condition_calculated / no_bounce_before_half_mask (TT:241-243)
 * TN: paddle_condition_calculated / missed_ball_calculated (TN:245-246) */
#define PPENV_FLAG_REWARD_CALC 1u     /* TT reward_calculated */
#define PPENV_FLAG_COND_CALC 2u       /* TT condition_calculated, TN paddle_condition_calculated */
#define PPENV_FLAG_NO_BOUNCE 4u       /* TT no_bounce_before_half_mask (initially set) */
#define PPENV_FLAG_MISSED_CALC 8u     /* TN missed_ball_calculated */

/* One revolute joint + the link it moves.  Link i's frame sits at the joint;
 * E_i(q) = origin_rot * Rot(axis, q) maps child coordinates to parent coordinates. */
typedef struct ppenv_joint {
    float origin_xyz[3];  /* child origin in parent coordinates */
    float origin_rot[9];  /* row-major, parent <- child at q = 0 */
    int32_t axis;         /* 0 = x, 1 = y, 2 = z */
    float lower, upper;   /* position limits [rad] */
    float kp, kd;         /* PD drive gains (TT:694-711) */
    float effort;         /* drive torque limit [N m] */
    float vel_limit;      /* |qd| limit [rad/s] */
    float armature;       /* added joint-space inertia */
    float mass;           /* link mass (composite of everything rigidly attached) */
    float com[3];         /* centre of mass, link coordinates */
    float inertia[6];     /* about the COM, link axes: xx, yy, zz, xy, xz, yz */
} ppenv_joint;

/* A frame rigidly attached to chain link `link` (0..6), or to the world when
 * link = -1 (then xyz/rot are the world pose).  Used for the observed bodies. */
typedef struct ppenv_frame {
    int32_t link;
    float xyz[3];
    float rot[9];         /* row-major, link <- frame */
} ppenv_frame;

/* Capsule (a != b) or sphere (a == b) collision shape attached like a frame. */
typedef struct ppenv_shape {
    int32_t link;         /* -1 = static, a/b in world coordinates */
    float a[3], b[3];
    float radius;
    float restitution;    /* COMBINED ball x surface coefficients (the host clamps each material's
                             restitution to restitution_max, then averages: PhysX default combine mode) */
    float friction;
} ppenv_shape;

typedef struct ppenv_box {
    float center[3];
    float half[3];
    float restitution;    /* combined, as in ppenv_shape */
    float friction;
} ppenv_box;

/* Everything the step needs that is not per-env state.  Built by the host
 * (isaacgym_amd/scene.py) from the task cfg; the same struct feeds the HIP
 * library and the CPU oracle.  Replaces the constants the reference scatters
 * over __init__/create_sim/_create_envs (TT:60-116,325-344,387-711) and the
 * `sim:` block of cfg/task/HumanoidPingpongTiltG1.yaml:77-98. */
typedef struct ppenv_config {
    int32_t abi_version;      /* PPENV_ABI_VERSION */
    int32_t variant;          /* PPENV_VARIANT_* */
    int32_t num_envs;         /* envs owned by this handle (local shard) */
    int32_t env_id_offset;    /* global id of local env 0 (multi-GPU shard; keys the RNG) */
    uint64_t seed;
    int32_t device_id;

    /* simulation (yaml sim: block) */
    float dt;                 /* 0.0083 */
    int32_t substeps;         /* 2 */
    int32_t ball_substeps;    /* ball micro-steps per substep (design param, default 4) */
    float gravity_z;          /* -9.8 (TT:329-331 overrides the yaml's -9.81) */
    float contact_offset;     /* 2e-4 */
    float bounce_threshold;   /* 0.2 */
    float max_depenetration_velocity; /* 10 */
    float clip_actions;       /* env.clipActions = 1.0 */
    float clip_obs;           /* VecTask clipObservations (inf = off) */

    /* articulated arm */
    float base_pos[3];        /* chain base (torso frame) in the world */
    float base_rot[9];        /* row-major, world <- base */
    ppenv_joint joint[PPENV_NUM_DOF];
    float init_dof_pos[PPENV_NUM_DOF];  /* zeros (TT:471,547) */
    float init_dof_vel[PPENV_NUM_DOF];

    /* observed bodies, in bodyStatesId order; entry 0 is the pelvis (root) */
    ppenv_frame obs_body[PPENV_NUM_OBS_BODIES];
    int32_t paddle_obs_index; /* which obs_body is the paddle (9) */

    /* actor root poses (TT:522-523,575,622); xyzw */
    float humanoid_root_pos[3], humanoid_root_quat[4];
    float table_root_pos[3], table_root_quat[4];
    float ball_init_pos[3], ball_init_quat[4];

    /* ball */
    float ball_radius, ball_mass;
    float ball_inertia_factor;   /* I = k m r^2 ; 2/3 = thin shell */
    float ball_restitution, ball_friction;   /* raw ball material (informational; surfaces carry combined values) */
    float ball_angular_damping;
    float restitution_max;       /* materials are clamped to this before combining (1.0) */

    /* static scene */
    float ground_z, ground_restitution, ground_friction;   /* combined */
    ppenv_box table;             /* playing surface slab */
    ppenv_box net;
    /* paddle blade: disc attached to chain link `paddle_link` */
    int32_t paddle_link;
    float paddle_center[3];      /* link coordinates */
    float paddle_normal[3];      /* unit, link coordinates */
    float paddle_radius, paddle_half_thickness;
    float paddle_restitution, paddle_friction;   /* combined */
    int32_t num_shapes;
    ppenv_shape shape[PPENV_MAX_SHAPES];
    float humanoid_bound_center[3]; /* ball farther than this sphere skips arm/body shapes */
    float humanoid_bound_radius;

    /* serve distribution, generate_random_speed_for_ball (TT:296-323, T3:289-305, TN:301-328) */
    float serve_speed_lo, serve_speed_hi;
    float serve_tilt_lo_deg, serve_tilt_hi_deg;
    float serve_tilt_z_lo_deg, serve_tilt_z_hi_deg;

    /* reward constants (TT:84,103-107) */
    int32_t max_episode_length;
    float alpha_velocity_reward;
    float power_coefficient;
    float penalty;
    float hit_table_reward;
    float not_hit_table_penalty;

    /* second humanoid (PPENV_VARIANT_T4 only; num_humanoids = 1 otherwise).  It is the same compiled arm model on
     * another base; its link-attached shapes are humanoid 1's, only the world-frame data differ. */
    int32_t num_humanoids;
    float base2_pos[3], base2_rot[9];
    float humanoid2_root_pos[3], humanoid2_root_quat[4];
    ppenv_shape shape2[PPENV_MAX_SHAPES];
    float humanoid2_bound_center[3];
} ppenv_config;

/* Device pointers of the buffers a handle owns, for zero-copy wrapping (the
 * analogue of gymtorch.wrap_tensor, TT:153-208).  SoA state arrays are
 * [field][num_envs]; surface tensors use the reference's AoS layouts. */
typedef struct ppenv_buffers {
    int32_t num_envs;
    int32_t num_agents;    /* A = 1, or 2 for PPENV_VARIANT_T4; agent a of env e owns row A*e + a of the surface tensors */
    /* VecTask surface */
    float* obs_buf;        /* [A*N, 80] f32 */
    float* rew_buf;        /* [A*N] f32 */
    int64_t* reset_buf;    /* [A*N] i64 */
    int64_t* progress_buf; /* [A*N] i64 */
    /* SoA simulation state */
    float* dof_pos;        /* [7*A][N]  (humanoid 1's seven rows first) */
    float* dof_vel;        /* [7*A][N] */
    float* dof_force;      /* [7*A][N]  drive torque of the last substep */
    float* ball;           /* [13][N] pos3 quat4 linvel3 angvel3 */
    uint32_t* flags;       /* [A][N] PPENV_FLAG_*, one word per side */
    uint32_t* episode;     /* [N] resets so far (RNG counter) */
    float* serve_override; /* [3][N]; used instead of the RNG while the override is on */
} ppenv_buffers;

/* ---- lifecycle ------------------------------------------------------------ */

/* Bytes of device memory one handle needs for `cfg->num_envs` envs. */
size_t ppenv_arena_bytes(const ppenv_config* cfg);

/* Create an environment.  If arena_dev is non-NULL it must point to at least
 * ppenv_arena_bytes(cfg) bytes of 256-byte-aligned device memory owned by the
 * caller (e.g. a torch tensor) that outlives the handle; otherwise the library
 * allocates.  State is initialised as after the reference's _create_envs
 * (TT:512-643): dofs at rest, ball at its start pose with a fresh serve.
 * Replaces VecTask.__init__ -> create_sim -> _create_envs (TT:118,325-344). */
int ppenv_create(const ppenv_config* cfg, void* arena_dev, size_t arena_bytes, void* stream, struct ppenv** out);
void ppenv_destroy(struct ppenv* env);
int ppenv_buffers_of(struct ppenv* env, ppenv_buffers* out);
int ppenv_config_of(struct ppenv* env, ppenv_config* out);

/* ---- the hot path ---------------------------------------------------------- */

/* One VecTask.step(): clamp actions, action->PD target (TT:1003-1014), snapshot
 * ball vx (TT:1020), `substeps` physics substeps (gym.simulate), progress += 1
 * (TT:1023), reward + reset decision (TT:739-758), masked reset (TT:847-906),
 * observations (TT:770-799) — one fused kernel launch.
 * actions_dev: [A*N, 7] f32 row-major (the policy's tensor, not copied; A = 2 for PPENV_VARIANT_T4). */
int ppenv_step(struct ppenv* env, const float* actions_dev, void* stream);
/* `count` consecutive steps in one call, step i on actions_dev[i] (device pointers, host array): the launches of VecTask.step called `count` times
 * (tasks/humanoid_pingpong_3_actor_tilt.py:1002-1052 per call) issued back to back from native code.  For open-loop bursts — scripted or logged actions, an
 * action repeated over several control steps: one host call per step is host-bound at these kernel times (0.8-0.9 G env-steps/s from Python against
 * 1.03 G for a 20-step burst at 16 384 envs); the burst runs at the rate of a captured graph of the same steps without a capture.  Each step's observations / rewards / resets are written as by
 * ppenv_step (the last step's remain in the handle's buffers). */
int ppenv_step_sequence(struct ppenv* env, const float* const* actions_dev, int32_t count, void* stream);
/* The same step with its outputs redirected: obs [A*N, 80] f32 (16-byte aligned), rew [A*N] f32, reset [A*N] int64 receive this step's
 * observations, rewards and reset flags instead of the handle's obs_buf / rew_buf / reset_buf (NULL: the handle's own).  A rollout
 * collector passes the slices of its horizon-major buffers ([horizon, num_actors, ...], rl_games' experience layout), so nothing is
 * copied after the step.  ppenv_reduce_stats and ppenv_get_state keep reading the handle's own buffers. */
int ppenv_step_into(struct ppenv* env, const float* actions_dev, float* obs_dev, float* rew_dev, int64_t* reset_dev, void* stream);

/* Reset every env to its initial state with a fresh serve and recompute
 * observations (VecTask.reset()). */
int ppenv_reset_all(struct ppenv* env, void* stream);

/* reset_idx(env_ids) -> _reset_idx (TT:809-812, 847-906): the listed envs (local ids, int64 like the reference's
 * `reset_buf.nonzero()` result, TT:1034; duplicates are harmless) go back to the initial state with the serve of their next
 * episode (TN keeps the dof state, TN:888-901), progress 0, flags initial; every other env is untouched.  The reference
 * leaves obs_buf alone there (compute_observations runs later in its post_physics_step); with refresh_obs != 0 the rows of
 * the listed envs are rewritten with the observation of the reset state, so that a caller outside a step sees it. */
int ppenv_reset_idx(struct ppenv* env, const int64_t* env_ids_dev, int32_t count, int refresh_obs, void* stream);

/* pre_physics_step alone (TT:1002-1014): the PD position targets the step hands to the simulator — what the reference
 * passes to gym.set_dof_position_target_tensor — for actions [A*N, 7] (clamped to +-clipActions first, as upstream
 * VecTask.step does).  pd_tar_dev [A*N, 7].  The same device function the fused step uses; exists so that the mapping can
 * be pinned to the reference's own pre_physics_step (tests/golden/pre_physics.npz). */
int ppenv_pd_targets(struct ppenv* env, const float* actions_dev, float* pd_tar_dev, void* stream);

/* generate_random_speed_for_ball of the handle's variant (TT:296-323, T3:289-305, TN:301-328, T4:299-326) on explicit draws:
 * draws_dev [M,3] = (speed, tilt degrees, tilt_z degrees) in the reference's draw order -> vel_dev [M,3].  The same device
 * function the reset path feeds from its counter RNG; pinned to the reference functions by tests/golden/serve_draws.npz. */
int ppenv_serve_from_draws(struct ppenv* env, const float* draws_dev, int32_t m, float* vel_dev, void* stream);

/* Domain randomisation (cfg/task/HumanoidPingpongTiltG1.yaml:100-169, TT:849-850: apply_randomizations at reset time) for the fused
 * step of the 3-actor variants.  The sampling — distributions, ranges, schedules, `frequency` — is host policy
 * (isaacgym_amd/vec_task.py apply_randomizations, after upstream VecTask.apply_randomizations); what the kernel consumes is the
 * RESULT: per-env tables in device memory, SoA [rows][num_envs] like the state, plus two noise amplitudes.  Any table pointer may be
 * NULL (= not randomised).  While a randomisation is set, ppenv_step runs the table-reading instantiation of the step kernel; with
 * none set (the default, and after ppenv_set_randomization(env, NULL)) it runs exactly the code it ran before — bit-identical results.
 *   dof_stiffness_scale, dof_damping_scale [7][N]   multiply the drive gains Kp, Kd of each dof       (dof_properties, yaml:144-156)
 *   link_mass_scale [7][N]                          multiplies mass AND inertia of each arm link      (rigid_body_properties.mass, yaml:124-131)
 *   restitution_scale, friction_scale [N]           multiply the combined ball-vs-humanoid-shape coefficients, paddle included (yaml:132-143;
 *                                                   the combined restitution stays clamped to restitution_max)
 *   action_noise_sigma / observation_noise_sigma    additive Gaussian white noise on the raw actions (before the clipActions clamp) and on
 *                                                   every observation value (yaml:106-113), drawn from the counter RNG keyed by
 *                                                   (seed, global env id, episode, progress, index) — identical in kernel and oracle
 * Gravity (sim_params.gravity, yaml:114-120) is one value for the whole simulation, as upstream: ppenv_set_gravity. */
typedef struct ppenv_randomization {
    const float* dof_stiffness_scale;   /* [7][N] or NULL */
    const float* dof_damping_scale;     /* [7][N] or NULL */
    const float* link_mass_scale;       /* [7][N] or NULL */
    const float* restitution_scale;     /* [N] or NULL */
    const float* friction_scale;        /* [N] or NULL */
    float action_noise_sigma;           /* 0 = off */
    float observation_noise_sigma;      /* 0 = off */
} ppenv_randomization;
/* The tables are NOT copied: they must stay valid (and may be rewritten in place between steps) until the randomisation is replaced or
 * cleared.  The 4-actor variant takes the same [7][N] tables: an env's entries apply to both of its humanoids (two instances of the yaml's one
 * "humanoid" actor); its 14 action draws and its two agents' observation rows have noise indices of their own. */
int ppenv_set_randomization(struct ppenv* env, const ppenv_randomization* dr /* NULL: off */);
/* sim_params.gravity of the whole simulation (yaml:83,114-120; the task's own override is -9.8, TT:329-331). */
int ppenv_set_gravity(struct ppenv* env, float gravity_z);

/* The handle's device status word (PPENV_STATUS_* bits), readable at any time without synchronising (it lives in pinned
 * host memory the kernels write through).  Every entry point that touches the state checks it first and fails with
 * PPENV_EDEVICE once it is non-zero. */
uint32_t ppenv_status(struct ppenv* env);

/* The kernel ppenv_step launches for this handle as it is configured NOW (schedule chosen at create, randomisation state), by the
 * demangled name rocprofv3's kernel trace prints, e.g. "step_kernel_split<pp::ModelG1, 1, 0, 1, false>".  Host-only, no GPU call; the
 * string is static.  (No reference counterpart: measurement plumbing — bench.py reports it as roofline.kernel.) */
const char* ppenv_step_kernel_name(struct ppenv* env);

/* What the reference prints every 40 steps (mean reward, mean progress: TT:763-766) plus the number of
 * finished episodes, as sums over this handle's envs: out_dev[4] (f64) = { sum rew_buf, sum progress_buf,
 * sum episode, num_envs }.  One small reduction launch; the caller all-reduces it across ranks. */
int ppenv_reduce_stats(struct ppenv* env, double* out_dev, void* stream);

/* ---- Isaac-Gym tensor-API mode --------------------------------------------- */

/* post_physics_step on caller-supplied simulator tensors in the reference's own
 * layouts: the drop-in for TT:1022-1039 when the physics comes from elsewhere
 * (and the entry the golden-vector parity tests drive).
 *   rigid_body_states_dev [N, 42, 13], root_states_dev [N, 3, 13] (read AND
 *   written: reset rows are restored in place like TT:853-862),
 *   dof_states_dev [N, 7, 2] (same), dof_force_dev [N, 7],
 *   pre_ball_vx_dev [N] (ball vx captured in pre_physics_step, TT:1020).
 * Uses and updates the handle's progress/flags/episode buffers and writes
 * obs_buf / rew_buf / reset_buf.  The handle's own SoA physics state is not
 * touched. */
int ppenv_post_physics_step(struct ppenv* env, const float* rigid_body_states_dev, float* root_states_dev,
                            float* dof_states_dev, const float* dof_force_dev, const float* pre_ball_vx_dev,
                            void* stream);

/* gym.refresh_* equivalents (TT:801-807): materialise the reference-layout
 * tensors from the SoA state on demand. */
int ppenv_refresh_root_states(struct ppenv* env, float* root_states_dev /* [N,3,13] */, void* stream);
int ppenv_refresh_dof_states(struct ppenv* env, float* dof_states_dev /* [N,7,2] */, void* stream);
int ppenv_refresh_dof_force(struct ppenv* env, float* dof_force_dev /* [N,7] */, void* stream);
/* Only the 10 observed humanoid bodies, the table and the ball rows are
 * meaningful; the other humanoid rows are written as the static root pose. */
int ppenv_refresh_rigid_body_states(struct ppenv* env, float* rb_states_dev /* [N,42,13] */, void* stream);

/* ---- 27-DoF variant (HumanoidPingpongTiltNESSparse27DOF), tensor-API mode only ------------------
 * tasks/humanoid_pingpong_3_actor_all_dof.py ("TA"): post_physics_step TA:1145-1192 on caller-supplied
 * simulator tensors — compute_pingpong_reward_nv TA:1440-1690 (+ compute_gradient_penalty TA:1245-1301,
 * compute_imitation_reward TA:1313-1418), _reset_idx TA:965-1028, compute_observations TA:867-904
 * (-> TA:1811-1927, 313 values).  Stateless: every buffer is the caller's (a real Isaac Gym's tensors, or the ones
 * ppenv_ta_simulate below steps). */
#define PPENV_TA_NUM_DOF 27
#define PPENV_TA_NUM_OBS 313          /* TA:106: 60 + 54 + 7 + 138 + 54 */
#define PPENV_TA_NUM_BALANCE_BODIES 23 /* bodyStatesIdBalance, HumanoidPingpongTiltNESSparse27DOFG1.yaml:57 */
/* sticky flags (TA:279-286) and diagnostic "count" flags (TA:289-293), one uint32 per env */
#define PPENV_TA_FLAG_PADDLE_COND 1u
#define PPENV_TA_FLAG_HIT_TABLE_CALC 2u
#define PPENV_TA_FLAG_DIE_PENALTY_CALC 4u
#define PPENV_TA_FLAG_HUMANOID_DIE_CALC 8u
#define PPENV_TA_COUNT_CLOSER 16u
#define PPENV_TA_COUNT_HIT_PADDLE 32u
#define PPENV_TA_COUNT_CROSS_NET 64u
#define PPENV_TA_COUNT_HIT_TABLE 128u
#define PPENV_TA_COUNT_FALL_DOWN 256u
#define PPENV_TA_COUNT_MASK 0x1F0u

typedef struct ppenv_ta_params {
    int32_t num_envs;
    int32_t max_episode_length;              /* 160 */
    int32_t is_train;                        /* termination_distance 0.32 when set (TA:1407-1413) */
    int32_t env_id_offset;
    uint64_t seed;
    float alpha_velocity_reward;             /* 3000 */
    float power_coefficient;                 /* 0.002 */
    float hit_paddle_reward;                 /* 200 */
    float miss_paddle_penalty_coefficient;   /* -100 */
    float cross_net_reward;                  /* 1000 */
    float hit_table_reward;                  /* 3000 */
    float not_hit_table_penalty;             /* -1000 */
    float die_penalty;                       /* -3000 */
    float init_root[PPENV_NUM_ACTORS][7];    /* initial pos3 + quat4 of humanoid, table, ball (TA:578-579,678-680) */
    float init_dof_pos[PPENV_TA_NUM_DOF];    /* zeros (TA:249-251) */
    float init_dof_vel[PPENV_TA_NUM_DOF];
    float serve_speed_lo, serve_speed_hi;            /* 5.0 .. 5.4   (TA:129) */
    float serve_tilt_lo_deg, serve_tilt_hi_deg;      /* -8 .. 3      (TA:130) */
    float serve_tilt_z_lo_deg, serve_tilt_z_hi_deg;  /* 14 .. 24     (TA:131) */
    float ball_y_lo, ball_y_hi, ball_z_lo, ball_z_hi; /* (-0.5, 0.1), (0.96, 1.05)  (TA:133-134) */
    /* 1: initial_rb_states_dev holds ONE env's [42,13] block that every env is compared with.  The task creates every humanoid in
     * the same pose (TA:578-579), so the 23 balance bodies' initial states (TA:200, 1152) are the same in every env; with this set
     * the step reads 2 KB once instead of gathering 52-byte rows at a 2184-byte stride from an [N,42,13] tensor (the gathers cost
     * more HBM traffic than everything else the 27-dof step moves).  0: a full [N,42,13] tensor, per env, as the reference keeps it. */
    int32_t initial_rb_shared;
    int32_t pad_;
} ppenv_ta_params;

/* One post_physics_step (TA:1145-1192).  Tensors in the reference's layouts, all device pointers:
 *   rb_states, initial_rb_states [N,42,13]; root_states [N,3,13] and dof_states [N,27,2] are updated in
 *   place for reset envs (TA:969-983); dof_force [N,27]; pre_ball_vx [N];
 *   reset_override [N,5] = ball y, z, vx, vy, vz to use at resets instead of the RNG, or NULL;
 *   flags, episode [N] u32 and progress [N] i64 are read-modify-write; obs [N,313], rew [N], reset [N] i64 out.
 * As in the reference, the diagnostic count flags of ALL envs are cleared whenever any env resets
 * (TA:1162-1166); this is the only cross-env effect and is done by a second small launch.
 * scratch_any_reset: one word the caller zeroes ONCE, before the first call; every call leaves it zero again (the
 * clearing launch resets it, which saves a memset per step). */
int ppenv_ta_post_physics_step(const ppenv_ta_params* params, const float* rb_states_dev, const float* initial_rb_states_dev,
                               float* root_states_dev, float* dof_states_dev, const float* dof_force_dev,
                               const float* pre_ball_vx_dev, const float* reset_override_dev, uint32_t* flags_dev,
                               uint32_t* episode_dev, int64_t* progress_dev, float* obs_dev, float* rew_dev, int64_t* reset_dev,
                               uint32_t* scratch_any_reset_dev /* 1 word */, void* stream);

/* ---- 27-DoF variant: the rigid-body step -------------------------------------------------------
 * Replaces, for tasks/humanoid_pingpong_3_actor_all_dof.py ("TA"), pre_physics_step TA:1124-1143 (action -> PD target,
 * snapshot of the ball's vx), gym.simulate (external, closed: a free-floating 27-DoF humanoid — fix_base_link = False,
 * TA:462 — standing on the ground plane TA:400-407, DOF_MODE_POS drives with the 27 gains TA:757-774, one ball) and the
 * gym.refresh_* calls TA:1150.  PARITY UNPINNED like every rigid-body step here: the specification is the build's own
 * (DESIGN.md "TA physics").  Stateless like ppenv_ta_post_physics_step: the simulation state IS the caller's Isaac-Gym
 * layout tensors (root_states [N,3,13], dof_states [N,27,2]), which _reset_idx (TA:965-1028, inside
 * ppenv_ta_post_physics_step) rewrites in place for reset envs exactly as the reference does. */
#define PPENV_TA_NUM_LINKS 28   /* pelvis (floating) + one link per dof; link k >= 1 is moved by dof k-1 (dof order TA:1303-1311) */
#define PPENV_TA_NUM_FIXED 12   /* rigid bodies welded to a link: imu, contour, d435, head, logo, ..., hands, paddle */
#define PPENV_TA_MAX_CONTACTS 24 /* points of the humanoid tested against the ground plane */

typedef struct ppenv_ta_link {
    int32_t parent;            /* link index, parents before children; -1 for the pelvis */
    int32_t axis;              /* 0/1/2: joint rotates about x/y/z of the child frame */
    int32_t body;              /* Isaac Gym rigid-body index of this link (pingpong_note.txt:33) */
    float origin_xyz[3];       /* child frame in the parent frame at q = 0 */
    float origin_rot[9];
    float mass;                /* link + everything welded to it */
    float com[3];
    float inertia[6];          /* about the com, link axes: xx yy zz xy xz yz */
    float lower, upper;        /* joint limits -> action offset / scale (TA:729-733) */
    float kp, kd;              /* TA:757-774 */
    float effort, vel_limit, armature;
} ppenv_ta_link;

typedef struct ppenv_ta_fixed {
    int32_t body;              /* Isaac Gym rigid-body index */
    int32_t link;              /* the link it is welded to */
    float xyz[3], rot[9];      /* pose in that link's frame */
} ppenv_ta_fixed;

typedef struct ppenv_ta_model {
    ppenv_ta_link link[PPENV_TA_NUM_LINKS];
    ppenv_ta_fixed fixed[PPENV_TA_NUM_FIXED];
    /* humanoid-vs-ground (plane TA:400-407): points fixed in links — the eight sole corners of the two ankle-roll links,
     * then knees, pelvis, torso, head, elbows, hands — each a penalty contact */
    int32_t num_contacts;
    int32_t contact_link[PPENV_TA_MAX_CONTACTS];
    float contact_point[PPENV_TA_MAX_CONTACTS][3];
    float ground_z;            /* plane height for the humanoid AND the ball (see scene.py: PlaneParams.distance = -0.21) */
    float foot_stiffness;      /* N/m per point, penalty contact integrated implicitly */
    float foot_damping;        /* N s/m per point, normal */
    float foot_tangent_damping;/* N s/m per point: regularised Coulomb friction */
    float foot_friction;       /* combined mu (plane 1.0, TA yaml:78-79; humanoid shapes 0.5, TA:588) */
    float contact_fade_depth;  /* m: the contact damper fades in over this much overlap ... */
    float contact_fade_force;  /* N: ... and the implicit terms with the normal force, so the law has no jump at touch-down / lift-off */
    float contact_max_penetration; /* the spring saturates here: deeper overlap is pushed out at <= k pen_max / c_n (what
                                      max_depenetration_velocity does in PhysX) instead of storing 1/2 k pen^2 */
    /* Joint limits as joint-space torques (a clamp of q or qd after the solve would stop a link without any reaction on
     * its parent: harmless on a bolted base, a momentum source on a floating one): beyond a position limit a spring-damper
     * limit_stiffness / limit_damping, above the velocity limit a damper vel_limit_damping, all integrated implicitly
     * (they only add to the joint's diagonal inertia term, like the PD drive). */
    float limit_stiffness, limit_damping, vel_limit_damping;
    /* ball-vs-humanoid collision: `scene.shape[s].link` and `scene.paddle_link` index ppenv_ta_model.link;
     * the broad-phase sphere (scene.humanoid_bound_radius) is centred on this point of this link */
    int32_t bound_link;
    float bound_center[3];
} ppenv_ta_model;

struct ppenv_ta_sim;   /* opaque: device copy of the constants */
typedef struct ppenv_ta_sim ppenv_ta_sim;

/* `scene` supplies what the 27-DoF scene shares with the 3-actor ones: dt, substeps, ball_substeps, gravity_z, clip_actions,
 * contact scalars, ball, table, net, paddle blade and the humanoid's ball-collision shapes (its 7-DoF arm tables are ignored). */
int ppenv_ta_sim_create(const ppenv_config* scene, const ppenv_ta_model* model, void* stream, ppenv_ta_sim** out);
void ppenv_ta_sim_destroy(ppenv_ta_sim* sim);
/* The GPU the handle lives on (scene->device_id at create time): every ppenv_ta_* entry selects it before launching, whatever
 * the caller's current device is.  The stateless entries (ppenv_ta_post_physics_step, ppenv_t4_rewards) launch on the device
 * that owns their output tensor. */
int ppenv_ta_sim_device(const ppenv_ta_sim* sim);
/* PPENV_STATUS_* bits reported by the handle's kernels (0 = healthy), readable without synchronising; ppenv_ta_step fails with
 * PPENV_EDEVICE once it is non-zero.  ppenv_ta_sim_kernel: which kernel ppenv_ta_step launches — 2 = chain-wave (one lane per
 * env, one wave per limb; needs the model compiled into the library, csrc/ppenv_model_g1_ta.h), 1 = four lanes per env,
 * 0 = one lane per env (any tree).  PPENV_TA_KERNEL=chain|quad|lane forces one. */
uint32_t ppenv_ta_sim_status(const ppenv_ta_sim* sim);
int ppenv_ta_sim_kernel(const ppenv_ta_sim* sim);
/* sim_params.gravity of the 27-DoF simulation (its yaml's randomization_params.sim_params.gravity, 27DOFG1.yaml:123-124; the task's own value
 * is -9.8, TA:384-386): takes effect for every launch enqueued on `stream` after the call.  gravity_z <= 0. */
int ppenv_ta_sim_set_gravity(ppenv_ta_sim* sim, float gravity_z, void* stream);
/* ... and by name, as rocprofv3 prints it ("ta_chain_kernel<false>", "ta_chain_kernel<true>" with a randomisation set, ...). */
const char* ppenv_ta_sim_kernel_name(const ppenv_ta_sim* sim);
/* Domain randomisation of the 27-DoF task (cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml carries the same task.randomization_params block as the
 * 7-dof yamls; TA's apply_randomizations call sits in its reset path as TT:849-850's does): per-env tables in device memory, SoA, NULL = not
 * randomised — drive stiffness / damping scales [27][N], link mass scales [28][N] (link 0 = pelvis; mass and inertia together), restitution and
 * friction scales [N] of the humanoid's shapes and the paddle; additive Gaussian noise on the raw actions (before the clipActions clamp) and on every
 * observation value, drawn from the counter RNG keyed by (params.seed, global env id, episode, progress at the step's start, index).  The tables are read
 * by every following ppenv_ta_step (the caller keeps them alive); dr NULL switches the randomisation off.  Only with ppenv_ta_sim_kernel() == 2 (the
 * chain-wave kernel: the scales multiply its compiled-in literals). */
typedef struct ppenv_ta_randomization {
    const float* dof_stiffness_scale;   /* [27][N] */
    const float* dof_damping_scale;     /* [27][N] */
    const float* link_mass_scale;       /* [28][N] */
    const float* restitution_scale;     /* [N] */
    const float* friction_scale;        /* [N] */
    float action_noise_sigma, observation_noise_sigma;
} ppenv_ta_randomization;
int ppenv_ta_sim_set_randomization(ppenv_ta_sim* sim, const ppenv_ta_randomization* dr);
/* The policy's first-layer input written by ppenv_ta_step itself (observation normalisation fused into the step
 * kernel): next to obs_buf the chain-wave kernel stores out[N, ld_out] fp16 = clamp((obs - mean) * inv_std, -clip, clip), columns
 * 313 .. ld_out-1 zero — bit for bit what ppenv_mlp_prepare_input (ppenv_policy.h) makes of obs_buf, without that launch.  mean /
 * inv_std: [313] fp32 device arrays read at every step (rl_games' RunningMeanStd in eval mode); ld_out even, >= 313 (320 for the
 * LDS-DMA layer kernels).  out NULL switches it off.  Only with ppenv_ta_sim_kernel() == 2. */
int ppenv_ta_sim_set_policy_input(ppenv_ta_sim* sim, const float* mean_dev, const float* inv_std_dev, float clip, void* out_f16_dev, int32_t ld_out);
/* Host-only (no GPU call): 1 when `model` equals, bit for bit, the tables compiled into the chain-wave kernel
 * (csrc/ppenv_model_g1_ta.h, generated by isaacgym_amd/modelgen_ta.py), 0 when it differs, < 0 on an invalid model. */
int ppenv_ta_model_is_compiled(const ppenv_config* scene, const ppenv_ta_model* model);
/* One pre_physics_step + gym.simulate + refresh.  actions [N,27]; root_states [N,3,13] (humanoid, table, ball) and
 * dof_states [N,27,2] are read and updated in place; rb_states [N,42,13], dof_force [N,27], pre_ball_vx [N] are written. */
int ppenv_ta_simulate(ppenv_ta_sim* sim, int32_t num_envs, const float* actions_dev, float* root_states_dev, float* dof_states_dev,
                      float* rb_states_dev, float* dof_force_dev, float* pre_ball_vx_dev, void* stream);
/* The whole VecTask step of the 27-DoF task in ONE launch: ppenv_ta_simulate followed by ppenv_ta_post_physics_step, the task
 * arithmetic running on the rigid-body kernel's registers / LDS tiles and the cross-env count-flag clear (TA:1162-1166) done by
 * the workgroup that finishes last.  Arguments as in those two entries; rb_states receives the pre-reset body states, root / dof
 * states the post-reset ones (TA:1150-1160).  With the chain-wave kernel rb_states_dev may be NULL: the [N,42,13] tensor is
 * then not materialised (nothing in the step reads it back; ppenv_ta_forward_kinematics produces it on demand). */
int ppenv_ta_step(ppenv_ta_sim* sim, const ppenv_ta_params* params, const float* actions_dev, const float* initial_rb_states_dev,
                  float* root_states_dev, float* dof_states_dev, float* rb_states_dev, float* dof_force_dev, float* pre_ball_vx_dev,
                  const float* reset_override_dev, uint32_t* flags_dev, uint32_t* episode_dev, int64_t* progress_dev, float* obs_dev,
                  float* rew_dev, int64_t* reset_dev, uint32_t* scratch_any_reset_dev /* 1 word */, void* stream);
/* pre_physics_step's PD targets alone for actions [N,27] (TA:1131 with offset / scale TA:729-733, after the clipActions
 * clamp) and TA's generate_random_speed_for_ball (TA:346-377) on explicit draws [M,3] = (speed, tilt deg, tilt_z deg):
 * the 27-DoF counterparts of ppenv_pd_targets / ppenv_serve_from_draws. */
int ppenv_ta_pd_targets(ppenv_ta_sim* sim, int32_t num_envs, const float* actions_dev, float* pd_tar_dev, void* stream);
int ppenv_ta_serve_from_draws(ppenv_ta_sim* sim, const float* draws_dev, int32_t m, float* vel_dev, void* stream);
/* rigid-body states of the current root / dof states without stepping (initial_rb_states of TA:1152; tests) */
int ppenv_ta_forward_kinematics(ppenv_ta_sim* sim, int32_t num_envs, const float* root_states_dev, const float* dof_states_dev,
                                float* rb_states_dev, void* stream);

/* ---- 4-actor variant (Humanoid12PingpongTiltG1), reward functions only ------------------------------
 * tasks/humanoid_pingpong_4_actor_tilt.py ("T4") defines compute_humanoid1_pingpong_reward T4:1113-1278
 * (token-identical to TT's compute_pingpong_reward_nv) and its mirror compute_humanoid2_pingpong_reward
 * T4:1280-1439 for the second humanoid at x = 3.5, but its class never calls them (T4:743 names a
 * function that does not exist) and leaves the two-agent wiring open (obs "TODO" T4:786).  This entry
 * evaluates both functions in one launch on the class's tensors; flags use the PPENV_FLAG_* bits, one
 * word per side.  Under @torch.jit.script the functions' `flag |= ...` statements compile to out-of-place
 * ops, so the reference never writes the caller's flag tensors: flags*_in are read-only here and the
 * function-local updated words come back in flags*_out for a caller that wants to carry them.
 * `progress` is taken as given (the functions read progress_buf, they do not advance it). */
#define PPENV_T4_NUM_DOF 14
#define PPENV_T4_NUM_BODIES 82   /* humanoid1 0-39, humanoid2 40-79, table 80, ball 81 (T4:127,169-172) */
#define PPENV_T4_NUM_ACTORS 4    /* humanoid1, humanoid2, table, ball (T4:181-185) */
typedef struct ppenv_t4_params {
    int32_t num_envs;
    int32_t max_episode_length;
    float alpha_velocity_reward, power_coefficient, penalty, hit_table_reward, not_hit_table_penalty;
} ppenv_t4_params;
int ppenv_t4_rewards(const ppenv_t4_params* params, const float* rb_states_dev /* [N,82,13] */,
                     const float* root_states_dev /* [N,4,13] */, const float* dof_states_dev /* [N,14,2] */,
                     const float* dof_force_dev /* [N,14] */, const float* pre_ball_vx_dev /* [N] */,
                     const int64_t* progress_dev /* [N] */, const uint32_t* flags1_in_dev, const uint32_t* flags2_in_dev /* [N] */,
                     uint32_t* flags1_out_dev, uint32_t* flags2_out_dev /* [N] */,
                     float* rew1_dev, float* rew2_dev, int64_t* reset1_dev, int64_t* reset2_dev /* [N] out */, void* stream);

/* ---- state I/O (parity tests, checkpoint) ----------------------------------- */

/* Serve velocities to use at the next resets instead of the RNG
 * (the reference draws them from Python `random`, TT:857-862). on = 0 disables. */
int ppenv_set_serve_override(struct ppenv* env, const float* serve_dev /* [N,3] row-major or NULL */, int on, void* stream);

/* Host-side blob of the full per-env state, SoA, in this order:
 *   dof_pos f32[7A][N], dof_vel f32[7A][N], dof_force f32[7A][N], ball f32[13][N],
 *   flags u32[A][N], episode u32[N], progress i64[A*N], reset i64[A*N]      (A = num_agents). */
size_t ppenv_state_bytes(struct ppenv* env);
int ppenv_get_state(struct ppenv* env, void* dst_host, size_t n);
int ppenv_set_state(struct ppenv* env, const void* src_host, size_t n);

const char* ppenv_last_error(void);
int ppenv_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PPENV_H */
