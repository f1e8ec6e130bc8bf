// ppenv_ta_task.h — post_physics_step of the 27-DoF task for ONE env (TA:1145-1192): compute_pingpong_reward_nv TA:1440-1690
// (+ compute_gradient_penalty TA:1245-1301, compute_imitation_reward TA:1313-1418), _reset_idx TA:965-1028,
// compute_observations TA:867-904 (-> TA:1811-1927).  fp32 in the reference's operation order.  Shared by the stand-alone
// tensor-API kernel (ppenv_ta.hip: every pointer into the caller's global tensors) and the fused 27-DoF step
// (ppenv_ta_sim.hip: rb / root / dofs / force / obs rows are the workgroup's LDS tiles).
#pragma once

#include "ppenv_device.h"

namespace pp {
namespace tatask {
constexpr int TA_ND = PPENV_TA_NUM_DOF;
constexpr int TA_NBAL = PPENV_TA_NUM_BALANCE_BODIES;
constexpr int kTaObsIdTable[NB] = {0, 31, 32, 33, 34, 35, 36, 37, 38, 39};   // bodyStatesIdPingpong, 27DOF yaml:56
constexpr int kTaBalIdTable[TA_NBAL] = {0, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 27};   // yaml:57
// The two id lists as arithmetic: with the bodies split over the lanes of a quad the index is a run-time value, and a table in
// memory made every body wait for a global load of its id before its state could be fetched.
__host__ __device__ constexpr int ta_obs_id(int j) { return j == 0 ? 0 : 30 + j; }
__host__ __device__ constexpr int ta_bal_id(int j) { return j + (j >= 1 ? 1 : 0) + (j >= 7 ? 1 : 0) + (j >= 16 ? 3 : 0); }
constexpr bool ta_id_formulas_match() {
    for (int j = 0; j < NB; j++) if (ta_obs_id(j) != kTaObsIdTable[j]) return false;
    for (int j = 0; j < TA_NBAL; j++) if (ta_bal_id(j) != kTaBalIdTable[j]) return false;
    return true;
}
static_assert(ta_id_formulas_match(), "body id lists");
// inverse maps: position of a rigid body in the balance / observation id lists, or -1
__host__ __device__ constexpr int ta_bal_index(int body) {
    for (int j = 0; j < TA_NBAL; j++) if (kTaBalIdTable[j] == body) return j;
    return -1;
}
__host__ __device__ constexpr int ta_obs_index(int body) {
    for (int j = 0; j < NB; j++) if (kTaObsIdTable[j] == body) return j;
    return -1;
}

// sum over the NR lanes that share an env (NR = 4: the lanes of a quad, lane & 3 = role; every lane of the quad must call it)
template <int NR>
__device__ __forceinline__ float lanes_sum(float x) {
    if (NR == 4) { x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); }
    return x;
}

// i: env index local to `p` (keys the reset draws together with p.env_id_offset).  rb [42][13], irb [42][13] (initial body
// states), root [3][13] and dofs [27][2] (updated in place on reset), force_row [27], ov_row: 5 reset values or null,
// o: the 313-wide observation row.
// NR lanes may share the env (NR = 1, or the 4 lanes of a quad with role = lane & 3): the 23-body sums and the observation
// blocks are split over them, the scalar reward / reset logic is computed by every lane, and role 0 alone writes what is
// per env.  `write` = false computes without storing (lanes past the last env still take part in the shuffles); when NR > 1
// root / dofs must be memory all NR lanes see in program order (an LDS tile).
template <int NR>
__device__ __forceinline__ void ta_task_env(const ppenv_ta_params& p, int i, const float* rb, const float* irb, float* root, float* dofs,
                                            const float* force_row, float pvx, const float* ov_row, uint32_t* flags_i, uint32_t* episode_i,
                                            long long* progress_i, float* o, float* rew_i, long long* reset_i, uint32_t* any_reset, int role,
                                            bool write) {
    const bool owner = write && role == 0;
    float* ball = root + 2 * 13;
    float bs[13];                            // the ball row as this lane sees it (post-reset further down)
#pragma unroll
    for (int k = 0; k < 13; k++) bs[k] = ball[k];
    const float root_x = root[0];
    const uint32_t ep_in = *episode_i;
    float q[TA_ND], qd[TA_ND];
#pragma unroll
    for (int d = 0; d < TA_ND; d++) { q[d] = dofs[2 * d]; qd[d] = dofs[2 * d + 1]; }
    long long prog = *progress_i + 1;                                           // TA:1146
    uint32_t f = *flags_i;
    const bool paddle_cond = f & PPENV_TA_FLAG_PADDLE_COND, hit_table_calc = f & PPENV_TA_FLAG_HIT_TABLE_CALC;
    const bool die_pen_calc = f & PPENV_TA_FLAG_DIE_PENALTY_CALC, hum_die = f & PPENV_TA_FLAG_HUMANOID_DIE_CALC;

    // ---- compute_imitation_reward TA:1313-1418 (is_g1)
    float pos_acc = 0.f, vel_acc = 0.f, norm_acc = 0.f;
    // With the bodies split over NR lanes each lane's share (initial minus current position and velocity of its <= 6 bodies) is fetched in one
    // batch with a constant trip count and kept for the imitation observations further down: as a run-time loop every body waited for its own
    // loads of the initial state from global memory, and did so twice.
    constexpr int kMine = NR > 1 ? (TA_NBAL + NR - 1) / NR : 1;
    float dpv[kMine][6];
    if (NR > 1) {
#pragma unroll
        for (int t = 0; t < kMine; t++) {
            const int j = role + t * NR;
            const int id = ta_bal_id(j < TA_NBAL ? j : 0);
            const float* b = rb + id * 13;
            const float* r = irb + id * 13;
            dpv[t][0] = r[0] - b[0]; dpv[t][1] = r[1] - b[1]; dpv[t][2] = r[2] - b[2];
            dpv[t][3] = r[7] - b[7]; dpv[t][4] = r[8] - b[8]; dpv[t][5] = r[9] - b[9];
        }
#pragma unroll
        for (int t = 0; t < kMine; t++) {
            if (role + t * NR >= TA_NBAL) continue;
            const float dp0 = dpv[t][0], dp1 = dpv[t][1], dp2 = dpv[t][2], dv0 = dpv[t][3], dv1 = dpv[t][4], dv2 = dpv[t][5];
            pos_acc += (dp0 * dp0 + dp1 * dp1 + dp2 * dp2) / 3.0f;
            vel_acc += (dv0 * dv0 + dv1 * dv1 + dv2 * dv2) / 3.0f;
            norm_acc += sqrtf(dp0 * dp0 + dp1 * dp1 + dp2 * dp2);   // |b - r| = |r - b|, the same products
        }
    } else {
        for (int j = role; j < TA_NBAL; j += NR) {
            const float* b = rb + ta_bal_id(j) * 13;
            const float* r = irb + ta_bal_id(j) * 13;
            float dp0 = r[0] - b[0], dp1 = r[1] - b[1], dp2 = r[2] - b[2];
            float dv0 = r[7] - b[7], dv1 = r[8] - b[8], dv2 = r[9] - b[9];
            pos_acc += (dp0 * dp0 + dp1 * dp1 + dp2 * dp2) / 3.0f;
            vel_acc += (dv0 * dv0 + dv1 * dv1 + dv2 * dv2) / 3.0f;
            float e0 = b[0] - r[0], e1 = b[1] - r[1], e2 = b[2] - r[2];
            norm_acc += sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        }
    }
    pos_acc = lanes_sum<NR>(pos_acc); vel_acc = lanes_sum<NR>(vel_acc); norm_acc = lanes_sum<NR>(norm_acc);
    float r_body_pos = expf(-50.f * (pos_acc / (float)TA_NBAL));                // TA:1349-1351
    float r_body_vel = expf(-4.0f * (vel_acc / (float)TA_NBAL));                // TA:1354-1356
    float s22 = 0.f, s5 = 0.f, sv = 0.f;
#pragma unroll
    for (int d = 0; d < 22; d++) { float e = p.init_dof_pos[d] - q[d]; s22 += e * e; }
#pragma unroll
    for (int d = 22; d < TA_ND; d++) { float e = p.init_dof_pos[d] - q[d]; s5 += e * e; }
#pragma unroll
    for (int d = 0; d < 22; d++) { float e = p.init_dof_vel[d] - qd[d]; sv += e * e; }
    float r22 = (0.2f * 50.0f) * expf(-(5.0f * 500.0f) * (s22 / 22.0f));       // TA:1372-1380
    float r5 = 0.2f * expf(-5.0f * (s5 / 5.0f));                                // TA:1383-1387
    float r_dof_vel = expf(-0.05f * (sv / 22.0f));                              // TA:1393,1401
    float ref_reward = r22 + r5 + 0.2f * r_dof_vel + 0.4f * r_body_pos + 0.2f * r_body_vel;   // TA:1403
    const bool has_fallen = (norm_acc / (float)TA_NBAL) > (p.is_train ? 0.32f : 1e6f);        // TA:1407-1415
    if (has_fallen) ref_reward = 1.0f * -50.0f;                                 // TA:1416-1417

    // ---- compute_pingpong_reward_nv TA:1440-1690
    const float* paddle = rb + 39 * 13;
    const float bx = bs[0], by = bs[1], bz = bs[2], vx = bs[7];
    const float pelvis_h = rb[2];
    if (has_fallen) f |= PPENV_TA_COUNT_FALL_DOWN;                              // TA:1525-1529
    const bool x_close = fabsf(bx - paddle[0]) < 0.2f;                          // TA:1544
    const bool first_close = x_close && !paddle_cond;
    float dy = by - paddle[1], dz = bz - paddle[2];
    float yz = sqrtf(dy * dy + dz * dz);                                        // TA:1548
    const bool in_circle = yz < 0.15f;
    float pos_reward = 0.f;
    if (first_close && !hum_die) pos_reward = in_circle ? p.hit_paddle_reward : p.miss_paddle_penalty_coefficient * yz;   // TA:1555-1563
    if (first_close && in_circle) f |= PPENV_TA_COUNT_CLOSER;                   // TA:1566-1570
    const bool hit_paddle = pvx < 0.f && vx > 1.5f;                             // TA:1577
    if (hit_paddle) f |= PPENV_TA_COUNT_HIT_PADDLE;
    float vel_reward = (hit_paddle && !paddle_cond && !hum_die) ? p.alpha_velocity_reward * fabsf(vx) : 0.f;   // TA:1586-1590
    if (x_close) f |= PPENV_TA_FLAG_PADDLE_COND;                                // TA:1595
    float time_penalty = (bx > root_x && vx < 0.f) ? -0.01f * (float)prog : 0.f;   // TA:1602-1607
    // compute_gradient_penalty TA:1245-1301
    const bool z_in = bz >= 0.82f && bz <= 0.83f && vx > 0.f;
    float ddx = bx - 2.5f, ddy = by - 0.0f;
    float dist = sqrtf(ddx * ddx + ddy * ddy);
    const bool in_range = bx >= 1.9f && bx <= 3.1f && by >= -0.6f && by <= 0.6f;
    if (z_in && in_range) f |= PPENV_TA_COUNT_HIT_TABLE;
    float hit_rp = 0.f;
    if (z_in && !hit_table_calc && !hum_die) hit_rp = in_range ? p.hit_table_reward : p.not_hit_table_penalty * dist;
    if (z_in) f |= PPENV_TA_FLAG_HIT_TABLE_CALC;
    // net TA:1619-1650
    const bool over_net = bx > 1.72f && bx < 1.78f && vx > 0.f;
    const bool suitable = bz > 0.96f && bz < 1.25f;
    float over_h = 0.f;
    if (!suitable) over_h = bz > 1.25f ? bz - 1.25f : 0.96f - bz;
    float net_rp = 0.f;
    if (over_net && !hum_die) net_rp = suitable ? p.cross_net_reward : -400.f * over_h;
    if (net_rp > 0.f) f |= PPENV_TA_COUNT_CROSS_NET;                            // TA:1652-1656
    float power = 0.f;
#pragma unroll
    for (int d = 0; d < TA_ND; d++) power += fabsf(force_row[d] * qd[d]);
    float power_reward = -p.power_coefficient * power;                          // TA:1664-1665
    float die_penalty = (bz < 0.78f && !die_pen_calc && !hum_die) ? p.die_penalty : 0.f;   // TA:1677-1679
    if (bz < 0.78f) f |= PPENV_TA_FLAG_DIE_PENALTY_CALC;                        // TA:1681
    if (pelvis_h < 0.97f) f |= PPENV_TA_FLAG_HUMANOID_DIE_CALC;                 // TA:1683
    float reward = 0.f + (((((((pos_reward + power_reward) + vel_reward) + hit_rp) + net_rp) + die_penalty) + time_penalty) + ref_reward);   // TA:1686
    const long long rst = (prog >= (long long)p.max_episode_length - 1) ? 1 : 0;   // TA:1688: time-out only

    // ---- _reset_idx TA:965-1028
    if (rst) {
        if (owner) atomicOr(any_reset, 1u);
        const uint32_t ep = ep_in + 1u;
        if (owner) *episode_i = ep;
        float ov[5];
        if (ov_row) {
#pragma unroll
            for (int k = 0; k < 5; k++) ov[k] = ov_row[k];
        } else {
            const uint32_t gid = (uint32_t)(p.env_id_offset + i);
            float u[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint64_t s = mix64(p.seed + 0x9E3779B97F4A7C15ull * ((uint64_t)gid + 1));
                uint64_t x = mix64(s + 0x9E3779B97F4A7C15ull * ((uint64_t)ep * 8 + k + 1));
                u[k] = (float)(x >> 40) * (1.0f / 16777216.0f);
            }
            ov[0] = p.ball_y_lo + (p.ball_y_hi - p.ball_y_lo) * u[0];           // draw order TA:976-979: y, z, speed, tilt, tilt_z
            ov[1] = p.ball_z_lo + (p.ball_z_hi - p.ball_z_lo) * u[1];
            const float speed = p.serve_speed_lo + (p.serve_speed_hi - p.serve_speed_lo) * u[2];
            const float a = p.serve_tilt_lo_deg + (p.serve_tilt_hi_deg - p.serve_tilt_lo_deg) * u[3];
            const float az = p.serve_tilt_z_lo_deg + (p.serve_tilt_z_hi_deg - p.serve_tilt_z_lo_deg) * u[4];
            const V3 sv = serve_from_draws(PPENV_VARIANT_TN, speed, a, az);      // TA:370-375 is TN's form
            ov[2] = sv.x; ov[3] = sv.y; ov[4] = sv.z;
        }
#pragma unroll
        for (int k = 0; k < 7; k++) bs[k] = p.init_root[2][k];
#pragma unroll
        for (int k = 7; k < 13; k++) bs[k] = 0.f;
        bs[1] = ov[0]; bs[2] = ov[1]; bs[7] = ov[2]; bs[8] = ov[3]; bs[9] = ov[4];
        if (owner) {
#pragma unroll
            for (int a = 0; a < 2; a++) {
#pragma unroll
                for (int k = 0; k < 7; k++) root[a * 13 + k] = p.init_root[a][k];
#pragma unroll
                for (int k = 7; k < 13; k++) root[a * 13 + k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 13; k++) ball[k] = bs[k];
        }
#pragma unroll
        for (int d = 0; d < TA_ND; d++) {
            q[d] = p.init_dof_pos[d]; qd[d] = p.init_dof_vel[d];
            if (owner) { dofs[2 * d] = q[d]; dofs[2 * d + 1] = qd[d]; }
        }
        prog = 0;
        f &= ~(PPENV_TA_FLAG_PADDLE_COND | PPENV_TA_FLAG_DIE_PENALTY_CALC | PPENV_TA_FLAG_HUMANOID_DIE_CALC | PPENV_TA_FLAG_HIT_TABLE_CALC);   // TA:1021-1024
    }
    if (owner) { *progress_i = prog; *flags_i = f; *rew_i = reward; *reset_i = rst; }

    // ---- compute_observations TA:867-904 (body states pre-reset, dof / ball post-reset)
    float rq[4] = {rb[3], rb[4], rb[5], rb[6]}, hinv[4];
    heading_quat_inv(rq, hinv);
    const V3 rootp = mk(rb[0], rb[1], rb[2]);
    if (!write) return;
    for (int j = role; j < NB; j += NR) {
        const float* b = rb + ta_obs_id(j) * 13;
        V3 lp = heading_rotate(hinv, mk(b[0], b[1], b[2]) - rootp);
        V3 lv = heading_rotate(hinv, mk(b[7], b[8], b[9]));
        o[3 * j] = lp.x; o[3 * j + 1] = lp.y; o[3 * j + 2] = lp.z;
        o[30 + 3 * j] = lv.x; o[30 + 3 * j + 1] = lv.y; o[30 + 3 * j + 2] = lv.z;
    }
#pragma unroll
    for (int d = 0; d < TA_ND; d++)
        if (NR == 1 || (d & (NR - 1)) == role) { o[60 + d] = q[d]; o[60 + TA_ND + d] = qd[d] * 0.1f; }
    if (role == 0) {
        V3 lb = heading_rotate(hinv, mk(bs[0], bs[1], bs[2]) - rootp);
        V3 lv = heading_rotate(hinv, mk(bs[7], bs[8], bs[9]));
        o[114] = lb.x; o[115] = lb.y; o[116] = lb.z; o[117] = lv.x; o[118] = lv.y; o[119] = lv.z;
        o[120] = lb.y + (lv.y / (-lv.x + 1e-6f)) * lb.x;                        // TA:1839
    }
    if (NR > 1) {                                                               // TA:1891-1927, from the differences kept above
#pragma unroll
        for (int tt = 0; tt < kMine; tt++) {
            const int j = role + tt * NR;
            if (j >= TA_NBAL) continue;
            V3 t = heading_rotate(hinv, mk(dpv[tt][0], dpv[tt][1], dpv[tt][2]));
            V3 tv = heading_rotate(hinv, mk(dpv[tt][3], dpv[tt][4], dpv[tt][5]));
            o[121 + 3 * j] = t.x * 10.f; o[122 + 3 * j] = t.y * 10.f; o[123 + 3 * j] = t.z * 10.f;
            o[121 + 3 * TA_NBAL + 3 * j] = tv.x; o[122 + 3 * TA_NBAL + 3 * j] = tv.y; o[123 + 3 * TA_NBAL + 3 * j] = tv.z;
        }
    } else {
        for (int j = role; j < TA_NBAL; j += NR) {                              // TA:1891-1927
            const float* b = rb + ta_bal_id(j) * 13;
            const float* r = irb + ta_bal_id(j) * 13;
            V3 t = heading_rotate(hinv, mk(r[0] - b[0], r[1] - b[1], r[2] - b[2]));
            V3 tv = heading_rotate(hinv, mk(r[7] - b[7], r[8] - b[8], r[9] - b[9]));
            o[121 + 3 * j] = t.x * 10.f; o[122 + 3 * j] = t.y * 10.f; o[123 + 3 * j] = t.z * 10.f;
            o[121 + 3 * TA_NBAL + 3 * j] = tv.x; o[122 + 3 * TA_NBAL + 3 * j] = tv.y; o[123 + 3 * TA_NBAL + 3 * j] = tv.z;
        }
    }
#pragma unroll
    for (int d = 0; d < TA_ND; d++)
        if (NR == 1 || (d & (NR - 1)) == role) { o[121 + 6 * TA_NBAL + d] = p.init_dof_pos[d]; o[121 + 6 * TA_NBAL + TA_ND + d] = p.init_dof_vel[d]; }
}
}  // namespace tatask
}  // namespace pp
