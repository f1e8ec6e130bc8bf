// ppenv_device.h — per-env arithmetic of the fused HumanoidPingpong step (fp32).
//
// One lane owns one env.  Everything here is straight-line inline code on
// registers: the articulated-body algorithm for the 7-DoF arm in link
// coordinates (Featherstone, RBDA table 7.1) with the PD drive folded in
// implicitly, the ball's micro-stepped contact model, and the reference's
// reward / reset / observation arithmetic (tasks/humanoid_pingpong_3_actor_tilt.py
// "TT":1105-1270,1640-1708 and the T3 / TN counterparts, cited inline).
//
// Functions are PP_HD (host + device) so that tests/csrc/host_shim.cpp can run
// the identical arithmetic on the CPU next to the oracle; the product only ever
// calls them from the HIP kernels in ppenv_kernels.hip.
//
// The chain topology is a compile-time parameter (joint axes and the links the
// collision shapes hang on) so that every per-joint loop unrolls into
// register-resident code; numeric model data comes from ppenv_config, which the
// kernels receive by value in the kernarg segment (wave-uniform -> SGPRs).
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ppenv.h"

#if defined(__HIPCC__)
#define PP_HD __host__ __device__ __forceinline__
#else
#define PP_HD inline __attribute__((always_inline))
#endif

// Diagnostic builds only (-DPP_STAMP, tools/gpu_stamps.sh): PP_STAMP_AT(k) records the shader clock of
// phase boundary k for lane 0 of each wave.  Product builds compile this to nothing.
#if defined(PP_STAMP) && defined(__HIP_DEVICE_COMPILE__)
#define PP_STAMP_AT(k)                                                                      \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        unsigned long long t_;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if ((threadIdx.x & 63) == 0) pp_stamp_buf[blockIdx.x * 32 + (k)] = t_;                      \
    } while (0)
extern __device__ unsigned long long pp_stamp_buf[];
#else
#define PP_STAMP_AT(k) do { } while (0)
#endif

namespace pp {

constexpr int ND = PPENV_NUM_DOF;
constexpr int NB = PPENV_NUM_OBS_BODIES;

// ---- joint-angle trig.  Device: the hardware v_sin/v_cos (abs err ~4e-7 on
// [-pi, pi], where the limits keep every joint); host shim: libm.
PP_HD void sincos_joint(float q, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    s = __sinf(q);
    c = __cosf(q);
#else
    s = sinf(q);
    c = cosf(q);
#endif
}

// 1-ulp hardware reciprocal / reciprocal square root for the physics (reward and observations keep
// IEEE division and sqrt so that thresholds and roundings follow the reference's torch arithmetic)
#if defined(__HIP_DEVICE_COMPILE__)
PP_HD float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
PP_HD float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
#else
PP_HD float rcp_fast(float x) { return 1.0f / x; }
PP_HD float rsq_fast(float x) { return 1.0f / sqrtf(x); }
#endif

// Everything the step needs at run time that is not compiled into the model: ~170 dwords derived from
// ppenv_config once, on the host, at create time (make_step_consts).  The kernels take it BY VALUE, so
// it sits in the kernarg segment and the whole block is fetched with a handful of wide scalar loads at
// wave start — one wait — instead of ~80 separate scalar loads scattered through the step, each of
// which a lone wave per SIMD cannot hide (measured: ~16k of a wave's 52k cycles).
struct RewardConsts {
    int32_t variant, max_episode_length;
    float alpha, power_coefficient, penalty, hit_table_reward, not_hit_table_penalty;
};
// Where one humanoid stands: everything that differs between humanoid 1 and humanoid 2 of the 4-actor variant
// (the arm model itself is the same compiled chain).
struct ArmSite {
    float base_pos[3], base_rot[9], base_grav[3];                         // chain base; base_grav = base_rot^T (0, 0, -gravity_z)
    float static_a[PPENV_MAX_SHAPES][3], static_b[PPENV_MAX_SHAPES][3];   // world end points of the static shapes
    float bound_center[3];
    float root_pos[3], root_rot[9], root_quat[4], hinv[4];   // pelvis (obs_body[0]); hinv = calc_heading_quat_inv(root_quat)
};
constexpr int kMaxArms = 2;
struct StepConsts {
    int32_t num_envs, env_id_offset, substeps, ball_substeps;
    int32_t num_arms, pad_;                      // humanoids per env (1; 2 for the 4-actor variant)
    uint64_t seed;
    RewardConsts rc;
    float h, clip_actions;                       // substep length dt / substeps
    ArmSite site[kMaxArms];                      // site[1] is used by PPENV_VARIANT_T4 only
    // ball / contacts
    float hb, inv_m, inv_h;                      // micro-step length, 1 / ball_substeps, 1 / h
    float contact_offset, bounce_threshold, depen_cap, ball_r, inv_kr, stick_factor, gdv, damp;
    float ground_z, ground_e, ground_mu;
    ppenv_box table, net;
    float bound_r2;
    float paddle_e, paddle_mu, e_cap;
    float shape_e[PPENV_MAX_SHAPES], shape_mu[PPENV_MAX_SHAPES];
    // task
    float table_pos[3], table_quat[4];
    float serve_speed_lo, serve_speed_hi, serve_tilt_lo_deg, serve_tilt_hi_deg, serve_tilt_z_lo_deg, serve_tilt_z_hi_deg;
    float ball_init_pos[3], ball_init_quat[4];
    float init_dof_pos[PPENV_NUM_DOF], init_dof_vel[PPENV_NUM_DOF];
};

// ------------------------------------------------------------------ small math
struct V3 { float x, y, z; };
struct M3 { float m[9]; };                 // row-major
struct S3 { float xx, yy, zz, xy, xz, yz; }; // symmetric 3x3

PP_HD V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
PP_HD V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
PP_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PP_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PP_HD V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PP_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PP_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PP_HD V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PP_HD V3 madd(V3 a, V3 b, float s) { return mk(a.x + b.x * s, a.y + b.y * s, a.z + b.z * s); }  // a + b*s
PP_HD float comp(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
PP_HD V3 unit(int k) { return mk(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f); }

PP_HD V3 row(const M3& a, int i) { return mk(a.m[3 * i], a.m[3 * i + 1], a.m[3 * i + 2]); }
PP_HD V3 col(const M3& a, int j) { return mk(a.m[j], a.m[3 + j], a.m[6 + j]); }
PP_HD V3 mul(const M3& a, V3 v) { return mk(dot(row(a, 0), v), dot(row(a, 1), v), dot(row(a, 2), v)); }
PP_HD V3 tmul(const M3& a, V3 v) { return mk(dot(col(a, 0), v), dot(col(a, 1), v), dot(col(a, 2), v)); }  // a^T v
PP_HD M3 mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
PP_HD M3 mul_t(const M3& a, const M3& b) {   // a * b^T
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = dot(row(a, i), row(b, j));
    return r;
}
PP_HD M3 ldm(const float* p) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = p[i];
    return r;
}
PP_HD M3 from_sym(const S3& s) {
    M3 r = {{s.xx, s.xy, s.xz, s.xy, s.yy, s.yz, s.xz, s.yz, s.zz}};
    return r;
}
PP_HD V3 mul(const S3& s, V3 v) {
    return mk(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
PP_HD V3 symcol(const S3& s, int k) { return k == 0 ? mk(s.xx, s.xy, s.xz) : (k == 1 ? mk(s.xy, s.yy, s.yz) : mk(s.xz, s.yz, s.zz)); }
PP_HD float symdiag(const S3& s, int k) { return k == 0 ? s.xx : (k == 1 ? s.yy : s.zz); }
// s -= u u^T * k
PP_HD void sym_rank1_sub(S3& s, V3 u, float k) {
    V3 uk = u * k;
    s.xx -= u.x * uk.x; s.yy -= u.y * uk.y; s.zz -= u.z * uk.z;
    s.xy -= u.x * uk.y; s.xz -= u.x * uk.z; s.yz -= u.y * uk.z;
}
// E * S * E^T for symmetric S
PP_HD S3 rot_sym(const M3& e, const S3& s) {
    M3 t = mul(e, from_sym(s));
    S3 r;
    r.xx = dot(row(t, 0), row(e, 0)); r.yy = dot(row(t, 1), row(e, 1)); r.zz = dot(row(t, 2), row(e, 2));
    r.xy = dot(row(t, 0), row(e, 1)); r.xz = dot(row(t, 0), row(e, 2)); r.yz = dot(row(t, 1), row(e, 2));
    return r;
}
PP_HD void add_sym(S3& a, const S3& b) { a.xx += b.xx; a.yy += b.yy; a.zz += b.zz; a.xy += b.xy; a.xz += b.xz; a.yz += b.yz; }

// ---- two at a time.  A lone wave issues one VALU instruction per ~4-5 cycles whatever its width (tools/micro/valu_issue.hip,
// ifetch.hip), and v_pk_{mul,add,fma}_f32 carries two fp32 operations in that slot, with either half of an operand broadcast for
// free.  Wherever the step applies one operator to two independent operands (the same rotation to the angular and the linear
// half of a spatial vector, to the two symmetric blocks of an articulated inertia, to the two ends of a capsule) the pair is
// written as a 2-vector so that the compiler emits the packed instruction.  The arithmetic per element is the scalar code's.
typedef float f2 __attribute__((vector_size(8)));
struct V3p { f2 x, y, z; };                   // two 3-vectors
struct S3p { f2 xx, yy, zz, xy, xz, yz; };    // two symmetric 3x3
PP_HD f2 pk(float a, float b) { f2 r; r[0] = a; r[1] = b; return r; }   // (not `{a, b}`: with a struct field for `a` clang loads a 2-vector from its address)
PP_HD V3p pk(V3 a, V3 b) { V3p r = {pk(a.x, b.x), pk(a.y, b.y), pk(a.z, b.z)}; return r; }
PP_HD S3p pk(S3 a, S3 b) { S3p r = {pk(a.xx, b.xx), pk(a.yy, b.yy), pk(a.zz, b.zz), pk(a.xy, b.xy), pk(a.xz, b.xz), pk(a.yz, b.yz)}; return r; }
PP_HD V3 lo(V3p a) { return mk(a.x[0], a.y[0], a.z[0]); }
PP_HD V3 hi(V3p a) { return mk(a.x[1], a.y[1], a.z[1]); }
PP_HD S3 lo(S3p a) { S3 r = {a.xx[0], a.yy[0], a.zz[0], a.xy[0], a.xz[0], a.yz[0]}; return r; }
PP_HD S3 hi(S3p a) { S3 r = {a.xx[1], a.yy[1], a.zz[1], a.xy[1], a.xz[1], a.yz[1]}; return r; }
PP_HD V3p operator+(V3p a, V3p b) { V3p r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
PP_HD V3p operator*(V3p a, float s) { V3p r = {a.x * s, a.y * s, a.z * s}; return r; }
PP_HD f2 dot(V3 a, V3p b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PP_HD f2 dot(V3p a, V3p b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PP_HD V3p cross(V3 a, V3p b) { V3p r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
PP_HD V3p cross(V3p a, V3 b) { V3p r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
PP_HD V3p mul(const M3& a, V3p v) { V3p r = {dot(row(a, 0), v), dot(row(a, 1), v), dot(row(a, 2), v)}; return r; }
PP_HD V3p tmul(const M3& a, V3p v) { V3p r = {dot(col(a, 0), v), dot(col(a, 1), v), dot(col(a, 2), v)}; return r; }
PP_HD V3p mul(const S3p& s, V3p v) {   // element by element: (s.lo v.lo, s.hi v.hi)
    V3p r = {s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z};
    return r;
}
PP_HD void sym_rank1_sub(S3p& s, V3p u, float k) {
    V3p uk = u * k;
    s.xx -= u.x * uk.x; s.yy -= u.y * uk.y; s.zz -= u.z * uk.z;
    s.xy -= u.x * uk.y; s.xz -= u.x * uk.z; s.yz -= u.y * uk.z;
}
PP_HD S3p rot_sym(const M3& e, const S3p& s) {   // E S E^T of both
    const f2 S[9] = {s.xx, s.xy, s.xz, s.xy, s.yy, s.yz, s.xz, s.yz, s.zz};
    f2 t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = e.m[3 * i] * S[j] + e.m[3 * i + 1] * S[3 + j] + e.m[3 * i + 2] * S[6 + j];
    S3p r;
    r.xx = t[0] * e.m[0] + t[1] * e.m[1] + t[2] * e.m[2]; r.yy = t[3] * e.m[3] + t[4] * e.m[4] + t[5] * e.m[5]; r.zz = t[6] * e.m[6] + t[7] * e.m[7] + t[8] * e.m[8];
    r.xy = t[0] * e.m[3] + t[1] * e.m[4] + t[2] * e.m[5]; r.xz = t[0] * e.m[6] + t[1] * e.m[7] + t[2] * e.m[8]; r.yz = t[3] * e.m[6] + t[4] * e.m[7] + t[5] * e.m[8];
    return r;
}

// E(q) = origin_rot * Rot(axis, q): child coordinates -> parent coordinates
template <int AX>
PP_HD M3 joint_rot(const float* r0, float c, float s) {
    V3 c0 = mk(r0[0], r0[3], r0[6]), c1 = mk(r0[1], r0[4], r0[7]), c2 = mk(r0[2], r0[5], r0[8]);
    V3 e0, e1, e2;
    if (AX == 0) { e0 = c0; e1 = c1 * c + c2 * s; e2 = c2 * c - c1 * s; }
    else if (AX == 1) { e0 = c0 * c - c2 * s; e1 = c1; e2 = c0 * s + c2 * c; }
    else { e0 = c0 * c + c1 * s; e1 = c1 * c - c0 * s; e2 = c2; }
    M3 e = {{e0.x, e1.x, e2.x, e0.y, e1.y, e2.y, e0.z, e1.z, e2.z}};
    return e;
}

// xyzw unit quaternion -> row-major rotation matrix
PP_HD void quat_to_rot(const float q[4], float r[9]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    r[0] = 1.f - 2.f * (y * y + z * z); r[1] = 2.f * (x * y - z * w); r[2] = 2.f * (x * z + y * w);
    r[3] = 2.f * (x * y + z * w); r[4] = 1.f - 2.f * (x * x + z * z); r[5] = 2.f * (y * z - x * w);
    r[6] = 2.f * (x * z - y * w); r[7] = 2.f * (y * z + x * w); r[8] = 1.f - 2.f * (x * x + y * y);
}
// rotation matrix -> xyzw quaternion with w >= 0
PP_HD void rot_to_quat(const M3& r, float q[4]) {
    // Shepperd's four cases (pivot on the trace or on the largest diagonal entry) written with selects: lanes of a wave pick different
    // cases, and as branches each lane paid for all four (a square root and three divisions apiece).
    const float tr = r.m[0] + r.m[4] + r.m[8];
    const bool c0 = tr > 0.f;
    const bool c1 = !c0 && r.m[0] > r.m[4] && r.m[0] > r.m[8];
    const bool c2 = !c0 && !c1 && r.m[4] > r.m[8];
    const bool c3 = !c0 && !c1 && !c2;
    const float piv = c1 ? r.m[0] : (c2 ? r.m[4] : r.m[8]);
    const float s = sqrtf(1.0f + (c0 ? tr : 2.f * piv - tr)) * 2.f;      // 1 + m_pp - (the other two) = 1 + 2 m_pp - tr
    const float inv = rcp_fast(s), big = 0.25f * s;   // v_rcp_f32: one ulp, against ten instructions for the IEEE quotient
    const float d0 = r.m[7] - r.m[5], d1 = r.m[2] - r.m[6], d2 = r.m[3] - r.m[1];
    const float a0 = r.m[1] + r.m[3], a1 = r.m[2] + r.m[6], a2 = r.m[5] + r.m[7];
    float w = c0 ? big : inv * (c1 ? d0 : (c2 ? d1 : d2));
    float x = c1 ? big : inv * (c0 ? d0 : (c2 ? a0 : a1));
    float y = c2 ? big : inv * (c0 ? d1 : (c1 ? a0 : a2));
    float z = c3 ? big : inv * (c0 ? d2 : (c1 ? a1 : a2));
    if (w < 0.f) { x = -x; y = -y; z = -z; w = -w; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// ------------------------------------------------------------------------ RNG
// Counter-based: (seed, global env id, episode, draw) -> U[0,1).  Same function
// as the oracle's (oracle/ppenv_oracle.c rng_uniform); the reference's host-side
// Python `random` stream (TT:307-312) cannot be reproduced by a vectorised env.
// (mix64 is the SplitMix64 finaliser, used by the 27-DoF tensor-API kernel's draws.)
PP_HD uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
// 32-bit avalanche hash ("lowbias32"): two multiplies per call.  64-bit multiplies cost four quarter-rate 32-bit
// multiplies each on CDNA; the serve draw sits on every step's path (drawn speculatively), so the keyed counter is 32-bit.
PP_HD uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
PP_HD float rng_uniform(uint64_t seed, uint32_t gid, uint32_t episode, uint32_t k) {
    uint32_t h = hash32(gid ^ (uint32_t)seed);
    h = hash32(h + episode * 0x9E3779B9u + (uint32_t)(seed >> 32));
    h = hash32(h + (k + 1u) * 0x85EBCA6Bu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// sin / cos of a serve angle (|x| <= ~0.5 rad): Taylor to x^9 / x^8, exact to fp32 there
PP_HD void sincos_small(float x, float& s, float& c) {
    float x2 = x * x;
    s = x * (1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f + x2 * (-1.f / 5040.f + x2 * (1.f / 362880.f)))));
    c = 1.f + x2 * (-0.5f + x2 * (1.f / 24.f + x2 * (-1.f / 720.f + x2 * (1.f / 40320.f))));
}
// generate_random_speed_for_ball from its three draws (speed, tilt and tilt_z in degrees, in the reference's draw order):
// T3:289-305, TT:296-323 (sic: the y component multiplies two sines and z has no tilt_z), TN:301-328, T4:299-326 = TT's,
// TA:346-377 = TN's form.  `form` is a PPENV_VARIANT_* value (TA passes PPENV_VARIANT_TN).
// Pinned to the reference functions by tests/golden/serve_draws.npz.
PP_HD V3 serve_from_draws(int form, float speed, float tilt_deg, float tilt_z_deg) {
    const float deg = 0.017453292519943295f;
    float sa, ca, sz, cz;
    sincos_small(tilt_deg * deg, sa, ca);
    sincos_small(tilt_z_deg * deg, sz, cz);
    if (form == PPENV_VARIANT_T3) return mk(-speed * ca, -speed * sa, 0.f);                                             // T3:296-300
    if (form == PPENV_VARIANT_TT || form == PPENV_VARIANT_T4) return mk(-speed * ca * cz, -speed * sa * sz, -speed * sa); // TT:307-318 (sic)
    return mk(-speed * ca * cz, speed * sa * cz, speed * sz);                                                           // TN:312-323
}
PP_HD V3 serve_velocity(const StepConsts& K, uint32_t gid, uint32_t episode) {
    float u0 = rng_uniform(K.seed, gid, episode, 0);
    float u1 = rng_uniform(K.seed, gid, episode, 1);
    float u2 = rng_uniform(K.seed, gid, episode, 2);
    float speed = K.serve_speed_lo + (K.serve_speed_hi - K.serve_speed_lo) * u0;
    float a = K.serve_tilt_lo_deg + (K.serve_tilt_hi_deg - K.serve_tilt_lo_deg) * u1;
    float az = K.serve_tilt_z_lo_deg + (K.serve_tilt_z_hi_deg - K.serve_tilt_z_lo_deg) * u2;
    return serve_from_draws(K.rc.variant, speed, a, az);
}
// pre_physics_step's action -> PD target (TT:1008 with offset / scale = (hi +- lo) / 2, TT:664-665; TA:1131, 729-733), after
// upstream VecTask.step's clamp to +-clipActions.  Pinned to the reference's pre_physics_step by tests/golden/pre_physics.npz.
PP_HD float pd_target(float action, float lo, float hi, float clip) {
    const float a = fminf(fmaxf(action, -clip), clip);
    return 0.5f * (hi + lo) + 0.5f * (hi - lo) * a;
}

// Domain randomisation of one env (include/ppenv.h ppenv_randomization): the table entries of this env, fetched at the start of the step by
// the table-reading instantiation of the step kernel (DR = true).  Every DR = false instantiation ignores it and compiles to the code it
// was before the randomisation existed.
struct EnvDR {
    float kp[PPENV_NUM_DOF], kd[PPENV_NUM_DOF], ms[PPENV_NUM_DOF];   // scales of the drive gains and of the link masses (inertias with them)
    float es, fs;                                                     // scales of the combined restitution / friction of the humanoid's shapes
    float act_sigma, obs_sigma;
    uint32_t key_progress;                                            // progress at the START of the step: keys the noise draws
};
// Additive Gaussian white noise: Box-Muller on two counter-RNG draws.  Round 4: the draws of a step come in PAIRS — indices 2 j and 2 j + 1
// share one (u1, u2) and take the cosine and the sine branch (index 2 j is the value round 3 drew) — and on the device the three
// transcendentals are the hardware's (v_log_f32, v_sqrt_f32, v_cos_f32 / v_sin_f32, which take their angle in revolutions: exactly
// Box-Muller's 2 pi u2): a draw costs ~15 instructions instead of ~70 (libm's logf / cosf with their range reductions), which was more
// than the step's own physics at 87 draws per env-step.  The oracle restates the same pairing with libm; the two agree to
// ~1e-6 of a unit normal, i.e. sigma x 1e-6 on a noisy value — far inside the parity tests' tolerances, no longer "the same float
// arithmetic".  The reference's own noise comes from torch's generator (upstream VecTask.apply_randomizations): the stream was never
// pinnable, the distribution is what matters (tests/test_policy_mlp.py checks the sampler's moments).
PP_HD void dr_gauss_pair(uint64_t seed, uint32_t gid, uint32_t episode, uint32_t progress, uint32_t pair, float& g_cos, float& g_sin) {
    const uint32_t k = progress * 256u + 2u * pair;
    const float u1 = rng_uniform(seed ^ 0x5DEECE66Dull, gid, episode, 2u * k), u2 = rng_uniform(seed ^ 0x5DEECE66Dull, gid, episode, 2u * k + 1u);
#if defined(__HIP_DEVICE_COMPILE__)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862944f * __builtin_amdgcn_logf(fmaxf(u1, 5.9604645e-8f)));     // sqrt(-2 ln u1) = sqrt(-2 ln 2 log2 u1)
    g_cos = rad * __builtin_amdgcn_cosf(u2);
    g_sin = rad * __builtin_amdgcn_sinf(u2);
#else
    const float rad = sqrtf(-2.0f * logf(fmaxf(u1, 5.9604645e-8f)));
    g_cos = rad * cosf(6.2831853f * u2);
    g_sin = rad * sinf(6.2831853f * u2);
#endif
}
PP_HD float dr_gauss(uint64_t seed, uint32_t gid, uint32_t episode, uint32_t progress, uint32_t index) {
    float c, s;
    dr_gauss_pair(seed, gid, episode, progress, index >> 1, c, s);      // (unrolled callers with static indices: the pair's common part is computed once)
    return (index & 1u) ? s : c;
}

// Per-joint constant groups of the compiled model: kinematics, PD drive, inertial.
struct JointKin { float origin_xyz[3]; float origin_rot[9]; };
struct JointDrive { float lower, upper, kp, kd, effort, vel_limit, armature; };
struct JointInertial { float mass; float com[3]; float inertia[6]; };
struct ModelShape { float a[3], b[3]; float radius; };                       // link-attached capsule / sphere
struct ModelPaddle { float center[3], normal[3]; float radius, half_thickness; };

// --------------------------------------------------------------- compiled arm model
// Every template parameter `T` below is a model struct generated by isaacgym_amd/modelgen.py
// (ppenv_model_g1.h): T::axis(i), T::kin(i), T::drive(i), T::inertial(i), T::kShapes,
// T::shape_link(s), T::shape(s), T::paddle(), T::tip_frame(j) — all constexpr, so with the per-joint
// loops unrolled they fold into instruction literals and exact zeros / identities drop out.

// world-frame collision geometry of the arm at the start of a substep: points and their velocities
template <int NSHAPES>
struct ArmGeom {
    V3 pc, pn, vpc, pnd;          // paddle centre, blade normal, centre velocity, normal rate (omega x n)
    V3 a[NSHAPES], b[NSHAPES];    // capsule end points (static shapes: constants)
    V3 va[NSHAPES], vb[NSHAPES];  // their velocities (static shapes: zero)
};

// per-joint quantities pass 1 leaves for passes 2 and 3
struct JointSave {
    float c, s;   // cos q, sin q
    V3 w, v;      // link velocity, link coordinates (angular, linear at the link origin)
    V3 ua, ub;    // U = I^A S, angular / linear halves
    float dinv, u;
};

// world pose + velocity of one observed body: pos3 quat4 lin3 ang3 is the
// reference's rigid-body row layout (TT:166-168)
struct BodyState { V3 pos; M3 rot; V3 lin, ang; };

// Forward kinematics sweep base -> tip.  Fills JointSave::{c,s,w,v}; hands every
// link's world transform to `vis(i, Rw, pw, w_link, v_link)`.
template <class T, class Visitor>
PP_HD void fk_sweep(const ArmSite& S, const float* q, const float* qd, JointSave* js, Visitor& vis) {
    M3 Rp = ldm(S.base_rot);
    V3 pp = ld3(S.base_pos);
    V3 wp = mk(0, 0, 0), vp = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < ND; i++) {
        const JointKin J = T::kin(i);
        float s, c;
        sincos_joint(q[i], s, c);
        M3 E = T::axis(i) == 0 ? joint_rot<0>(J.origin_rot, c, s) : (T::axis(i) == 1 ? joint_rot<1>(J.origin_rot, c, s) : joint_rot<2>(J.origin_rot, c, s));
        V3 r = ld3(J.origin_xyz);
        V3 pw = pp + mul(Rp, r);
        M3 Rw = mul(Rp, E);
        V3 w = tmul(E, wp);
        V3 v = tmul(E, vp + cross(wp, r));
        if (T::axis(i) == 0) w.x += qd[i]; else if (T::axis(i) == 1) w.y += qd[i]; else w.z += qd[i];
        js[i].c = c; js[i].s = s; js[i].w = w; js[i].v = v;
        vis(i, Rw, pw, w, v);
        Rp = Rw; pp = pw; wp = w; vp = v;
    }
}

// collects the moving collision geometry during an FK sweep
template <class T>
struct GeomVisitor {
    ArmGeom<T::kShapes>& g;
    PP_HD explicit GeomVisitor(ArmGeom<T::kShapes>& gg) : g(gg) {}
    PP_HD void operator()(int i, const M3& Rw, V3 pw, V3 w, V3 v) {
        const V3 ww = mul(Rw, w), vw = mul(Rw, v);   // link angular / origin velocity in world axes
        if (i == ND - 1) {   // paddle_link is validated to be the last link at create time
            const ModelPaddle P = T::paddle();
            V3 off = mul(Rw, ld3(P.center));
            g.pc = pw + off;
            g.vpc = vw + cross(ww, off);
            g.pn = mul(Rw, ld3(P.normal));
            g.pnd = cross(ww, g.pn);
        }
#pragma unroll
        for (int s = 0; s < T::kShapes; s++)
            if (T::shape_link(s) == i) {
                const ModelShape sh = T::shape(s);
                V3 oa = mul(Rw, ld3(sh.a)), ob = mul(Rw, ld3(sh.b));
                g.a[s] = pw + oa; g.va[s] = vw + cross(ww, oa);
                g.b[s] = pw + ob; g.vb[s] = vw + cross(ww, ob);
            }
    }
};
template <class T>
PP_HD void static_geometry(const ArmSite& S, ArmGeom<T::kShapes>& g) {
#pragma unroll
    for (int s = 0; s < T::kShapes; s++)
        if (T::shape_link(s) < 0) { g.a[s] = ld3(S.static_a[s]); g.b[s] = ld3(S.static_b[s]); g.va[s] = mk(0, 0, 0); g.vb[s] = mk(0, 0, 0); }
}

// geometry + the observed bodies (obs_body[1..7] are the chain links, [8],[9] ride on the last link).
// FULL = false keeps only what the fused step consumes (position + linear velocity: TT:1671-1673 reads
// body_pos / body_vel, the reward reads the paddle position); FULL = true also keeps orientation and
// angular velocity for ppenv_refresh_rigid_body_states.
template <class T, bool FULL>
struct BodyVisitor {
    ArmGeom<T::kShapes>& g;
    BodyState* bodies;   // [NB]
    PP_HD BodyVisitor(ArmGeom<T::kShapes>& gg, BodyState* b) : g(gg), bodies(b) {}
    PP_HD void operator()(int i, const M3& Rw, V3 pw, V3 w, V3 v) {
        GeomVisitor<T> gv(g);
        gv(i, Rw, pw, w, v);
        V3 ww = mul(Rw, w), vw = mul(Rw, v);
        bodies[1 + i].pos = pw; bodies[1 + i].lin = vw;
        if (FULL) { bodies[1 + i].rot = Rw; bodies[1 + i].ang = ww; }
        if (i == ND - 1) {
#pragma unroll
            for (int j = 8; j < NB; j++) {
                const JointKin f = T::tip_frame(j);
                V3 off = mul(Rw, ld3(f.origin_xyz));
                bodies[j].pos = pw + off;
                bodies[j].lin = vw + cross(ww, off);
                if (FULL) { bodies[j].rot = mul(Rw, ldm(f.origin_rot)); bodies[j].ang = ww; }
            }
        }
    }
};
template <bool FULL>
PP_HD void static_body(const ArmSite& S, BodyState& b) {   // obs_body[0]: the pelvis, fixed at the root pose
    b.pos = ld3(S.root_pos); b.lin = mk(0, 0, 0);
    if (FULL) { b.rot = ldm(S.root_rot); b.ang = mk(0, 0, 0); }
}

// --------------------------------------------------- ABA passes 2 and 3 (RBDA 7.1)
// tau / arm_eff: drive torque and joint-space inertia added on the diagonal
// (armature + the implicit PD terms).  Returns qdd.
template <class T, bool DR = false>
PP_HD void aba_solve(const ArmSite& S, JointSave* js, const float* qd, const float* tau, const float* arm_eff, float* qdd, const EnvDR* dr = nullptr) {
    // articulated inertia / bias force handed down by the child, in this link's coordinates
    S3 cA = {0, 0, 0, 0, 0, 0}, cD = {0, 0, 0, 0, 0, 0};
    M3 cB = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    V3 cn = mk(0, 0, 0), cf = mk(0, 0, 0);
#pragma unroll
    for (int i = ND - 1; i >= 0; i--) {
        const JointKin JK = T::kin(i);
        const JointInertial J = T::inertial(i);
        const int ax = T::axis(i);
        V3 w = js[i].w, v = js[i].v;
        // rigid-body inertia about the link origin: [[Io, m c x],[m c x^T, m 1]]
        const float msc = DR ? dr->ms[i] : 1.f;              // rigid_body_properties.mass scaling: mass and inertia together
        float m = DR ? J.mass * msc : J.mass;
        V3 cm = ld3(J.com);
        V3 mc = cm * m;
        float cc = dot(cm, cm);
        float Jin[6];
#pragma unroll
        for (int t = 0; t < 6; t++) Jin[t] = DR ? J.inertia[t] * msc : J.inertia[t];
        S3 A = {Jin[0] + m * (cc - cm.x * cm.x), Jin[1] + m * (cc - cm.y * cm.y), Jin[2] + m * (cc - cm.z * cm.z),
                Jin[3] - m * cm.x * cm.y, Jin[4] - m * cm.x * cm.z, Jin[5] - m * cm.y * cm.z};
        V3 h_ang = mul(A, w) + cross(mc, v);
        V3 h_lin = v * m - cross(mc, w);
        V3 pn = cross(w, h_ang) + cross(v, h_lin) + cn;
        V3 pf = cross(w, h_lin) + cf;
        add_sym(A, cA);
        M3 B = {{cB.m[0], cB.m[1] - mc.z, cB.m[2] + mc.y, cB.m[3] + mc.z, cB.m[4], cB.m[5] - mc.x, cB.m[6] - mc.y, cB.m[7] + mc.x, cB.m[8]}};
        S3 D = {cD.xx + m, cD.yy + m, cD.zz + m, cD.xy, cD.xz, cD.yz};

        V3 ua = symcol(A, ax), ub = row(B, ax);
        float dinv = rcp_fast(symdiag(A, ax) + arm_eff[i]);
        float u = tau[i] - comp(pn, ax);
        js[i].ua = ua; js[i].ub = ub; js[i].dinv = dinv; js[i].u = u;
        if (i > 0) {
            sym_rank1_sub(A, ua, dinv);
            sym_rank1_sub(D, ub, dinv);
            V3 uad = ua * dinv;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float k = comp(uad, r);
                B.m[3 * r] -= k * ub.x; B.m[3 * r + 1] -= k * ub.y; B.m[3 * r + 2] -= k * ub.z;
            }
            V3 e = unit(ax);
            V3 cw = cross(w, e) * qd[i], cv = cross(v, e) * qd[i];   // c = v x S qd
            float ud = u * dinv;
            V3 pan = pn + mul(A, cw) + mul(B, cv) + ua * ud;
            V3 paf = pf + tmul(B, cw) + mul(D, cv) + ub * ud;
            // to the parent's coordinates: rotate by E, shift the origin by r
            M3 E = ax == 0 ? joint_rot<0>(JK.origin_rot, js[i].c, js[i].s) : (ax == 1 ? joint_rot<1>(JK.origin_rot, js[i].c, js[i].s) : joint_rot<2>(JK.origin_rot, js[i].c, js[i].s));
            V3 r = ld3(JK.origin_xyz);
            S3 Ar = rot_sym(E, A), Dr = rot_sym(E, D);
            M3 Br = mul_t(mul(E, B), E);
            V3 nr = mul(E, pan), fr = mul(E, paf);
            // Bp = Br + r x Dr  (column-wise cross)
            M3 Dm = from_sym(Dr);
            M3 Bp;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                V3 x = cross(r, col(Dm, j));
                Bp.m[j] = Br.m[j] + x.x; Bp.m[3 + j] = Br.m[3 + j] + x.y; Bp.m[6 + j] = Br.m[6 + j] + x.z;
            }
            // Ap[i][j] = Ar[i][j] + (r x row_j(Bp))[i] + (r x row_i(Br))[j]
            V3 wp0 = cross(r, row(Bp, 0)), wp1 = cross(r, row(Bp, 1)), wp2 = cross(r, row(Bp, 2));
            V3 wb0 = cross(r, row(Br, 0)), wb1 = cross(r, row(Br, 1)), wb2 = cross(r, row(Br, 2));
            cA.xx = Ar.xx + wp0.x + wb0.x; cA.yy = Ar.yy + wp1.y + wb1.y; cA.zz = Ar.zz + wp2.z + wb2.z;
            cA.xy = Ar.xy + wp1.x + wb0.y; cA.xz = Ar.xz + wp2.x + wb0.z; cA.yz = Ar.yz + wp2.y + wb1.z;
            cB = Bp; cD = Dr;
            cn = nr + cross(r, fr); cf = fr;
        }
    }
    // pass 3: accelerations base -> tip; the base "accelerates" upward by |g|
    V3 aw = mk(0, 0, 0);
    V3 av = ld3(S.base_grav);
#pragma unroll
    for (int i = 0; i < ND; i++) {
        const JointKin J = T::kin(i);
        const int ax = T::axis(i);
        M3 E = ax == 0 ? joint_rot<0>(J.origin_rot, js[i].c, js[i].s) : (ax == 1 ? joint_rot<1>(J.origin_rot, js[i].c, js[i].s) : joint_rot<2>(J.origin_rot, js[i].c, js[i].s));
        V3 r = ld3(J.origin_xyz);
        V3 e = unit(ax);
        V3 aw2 = tmul(E, aw) + cross(js[i].w, e) * qd[i];
        V3 av2 = tmul(E, av + cross(aw, r)) + cross(js[i].v, e) * qd[i];
        float a = (js[i].u - dot(js[i].ua, aw2) - dot(js[i].ub, av2)) * js[i].dinv;
        qdd[i] = a;
        if (ax == 0) aw2.x += a; else if (ax == 1) aw2.y += a; else aw2.z += a;
        aw = aw2; av = av2;
    }
}

// PD drive (DOF_MODE_POS, TT:414,463) + one semi-implicit Euler substep of the arm.
// The PD terms are integrated implicitly (h Kd + h^2 Kp on the joint-space inertia diagonal) and the explicit part of the
// torque is clamped to the effort limit: continuous at saturation, so a joint within rounding of the limit does not
// change the step (an earlier draft switched to a constant torque without the implicit terms there).
template <class T, bool DR = false>
PP_HD void arm_substep(const ArmSite& S, JointSave* js, float* q, float* qd, const float* target, float h, float* tau_drive, const EnvDR* dr = nullptr) {
    float tau[ND], arm[ND], qdd[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const JointDrive J = T::drive(d);
        const float kp = DR ? J.kp * dr->kp[d] : J.kp, kd = DR ? J.kd * dr->kd[d] : J.kd;   // dof_properties.stiffness / damping scaling
        float err = target[d] - q[d];
        tau[d] = fminf(fmaxf(kp * (err - h * qd[d]) - kd * qd[d], -J.effort), J.effort);
        arm[d] = J.armature + h * kd + h * h * kp;
    }
    aba_solve<T, DR>(S, js, qd, tau, arm, qdd, dr);
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const JointDrive J = T::drive(d);
        const float kp = DR ? J.kp * dr->kp[d] : J.kp, kd = DR ? J.kd * dr->kd[d] : J.kd;
        float err = target[d] - q[d];
        float vn = qd[d] + h * qdd[d];
        // dof_force reports the drive torque at the end-of-substep velocity, within the actuator's limit
        tau_drive[d] = fminf(fmaxf(kp * (err - h * vn) - kd * vn, -J.effort), J.effort);
        vn = fminf(fmaxf(vn, -J.vel_limit), J.vel_limit);
        float qn = q[d] + h * vn;
        if (qn > J.upper) { qn = J.upper; vn = fminf(vn, 0.f); }
        if (qn < J.lower) { qn = J.lower; vn = fmaxf(vn, 0.f); }
        q[d] = qn; qd[d] = vn;
    }
}

// ---------------------------------------------------------------- ball contacts
struct Ball { V3 p, v, w; float quat[4]; };

// hot scalars of the contact model, fetched / derived once per substep
struct BallConsts {
    float contact_offset, bounce_threshold, depen_cap, r;
    float inv_kr;        // 1 / (k r):  spin change per unit tangential impulse, I = k m r^2
    float stick_factor;  // 1 / (1 + 1/k): tangential impulse that stops slipping, per unit slip speed
};

// n: unit normal surface -> ball; s: separation; u: surface velocity at the contact
PP_HD void contact_resolve(const BallConsts& k, Ball& b, V3 n, float s, V3 u, float e, float mu) {
    if (!(s < k.contact_offset)) return;
    V3 vrel = b.v + cross(b.w, n * (-k.r)) - u;
    float vn = dot(vrel, n);
    if (vn < 0.f) {
        float e_eff = (-vn > k.bounce_threshold) ? e : 0.f;
        float jn = -(1.f + e_eff) * vn;
        V3 vt = vrel - n * vn;
        float vt2 = dot(vt, vt);
        if (vt2 > 1e-18f) {
            float inv = rsq_fast(vt2);
            V3 dir = vt * inv;
            float jt = fminf(mu * jn, vt2 * inv * k.stick_factor);
            b.v = b.v - dir * jt;
            b.w = b.w + cross(n, dir) * (jt * k.inv_kr);
        }
        b.v = b.v + n * jn;
    }
    if (s < 0.f) b.p = madd(b.p, n, fminf(-s, k.depen_cap));
}
PP_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

PP_HD void contact_box(const BallConsts& k, Ball& b, const ppenv_box& box) {
    V3 d = b.p - ld3(box.center);
    V3 h = ld3(box.half);
    // cheap reject: farther than radius + contact offset from the box on some axis
    float reach = k.r + k.contact_offset;
    if (fabsf(d.x) > h.x + reach || fabsf(d.y) > h.y + reach || fabsf(d.z) > h.z + reach) return;
    V3 q = mk(clampf(d.x, -h.x, h.x), clampf(d.y, -h.y, h.y), clampf(d.z, -h.z, h.z));
    V3 diff = d - q;
    float d2 = dot(diff, diff);
    V3 n;
    float s;
    if (d2 > 1e-24f) { float inv = rsq_fast(d2); n = diff * inv; s = d2 * inv - k.r; }
    else {   // centre inside the box: leave through the nearest face
        float px = h.x - fabsf(d.x), py = h.y - fabsf(d.y), pz = h.z - fabsf(d.z);
        if (px <= py && px <= pz) { n = mk(d.x >= 0.f ? 1.f : -1.f, 0, 0); s = -px - k.r; }
        else if (py <= pz) { n = mk(0, d.y >= 0.f ? 1.f : -1.f, 0); s = -py - k.r; }
        else { n = mk(0, 0, d.z >= 0.f ? 1.f : -1.f); s = -pz - k.r; }
    }
    contact_resolve(k, b, n, s, mk(0, 0, 0), box.restitution, box.friction);
}
PP_HD void contact_capsule(const BallConsts& k, Ball& b, V3 a, V3 bb, V3 ua, V3 ub, float radius, float e, float mu) {
    V3 ab = bb - a;
    float l2 = dot(ab, ab), t = 0.f;
    if (l2 > 1e-12f) t = clampf(dot(b.p - a, ab) * rcp_fast(l2), 0.f, 1.f);
    V3 cp = madd(a, ab, t);
    V3 diff = b.p - cp;
    float d2 = dot(diff, diff);
    if (d2 > (radius + k.r + k.contact_offset) * (radius + k.r + k.contact_offset)) return;   // not touching
    float inv = d2 > 1e-24f ? rsq_fast(d2) : 0.f;
    V3 n = d2 > 1e-24f ? diff * inv : mk(0, 0, 1);
    V3 u = madd(ua, ub - ua, t);
    contact_resolve(k, b, n, d2 * inv - radius - k.r, u, e, mu);
}
// the paddle blade: solid disc, centre cc, unit axis nn, centre velocity uc, axis rate nd
PP_HD void contact_disc(const BallConsts& k, Ball& b, V3 cc, V3 nn, V3 uc, V3 nd, float R, float tp, float e, float mu) {
    V3 d = b.p - cc;
    float hgt = dot(d, nn);
    V3 radial = d - nn * hgt;
    float rr2 = dot(radial, radial);
    float reach = k.r + k.contact_offset;
    if (fabsf(hgt) > tp + reach || rr2 > (R + reach) * (R + reach)) return;   // not touching
    float rr = sqrtf(rr2);
    V3 n, closest;
    float s;
    if (fabsf(hgt) < tp && rr < R) {   // centre inside the blade: leave through the nearer face
        float sg = hgt >= 0.f ? 1.f : -1.f;
        n = nn * sg;
        s = -(tp - fabsf(hgt)) - k.r;
        closest = cc + nn * (sg * tp) + radial;
    } else {
        // ball centre minus closest point, split into its axial and radial parts: formed as `b.p - closest` the
        // radial part of a centre over the face (rr < R) is rounding noise instead of exactly zero, and a ball pressed
        // 2 cm into the blade (centre 0.1 mm above the face) then gets a normal that is off by 1e-3
        float dh = hgt - clampf(hgt, -tp, tp);
        float er = fmaxf(rr - R, 0.f);
        V3 diff = nn * dh + (er > 0.f ? radial * (er * rcp_fast(rr)) : mk(0, 0, 0));
        closest = b.p - diff;
        float d2 = dh * dh + er * er;
        float inv = d2 > 1e-24f ? rsq_fast(d2) : 0.f;
        n = d2 > 1e-24f ? diff * inv : nn;
        s = d2 * inv - k.r;
    }
    V3 u = uc + cross(cross(nn, nd), closest - cc);
    contact_resolve(k, b, n, s, u, e, mu);
}

PP_HD V3 lerp(V3 a, V3 b, float f) { return madd(a, b - a, f); }

// One physics substep of the ball: ball_substeps micro-steps against the static scene and the arm's shapes,
// which move linearly from their pose at the START of the substep with the velocities they have there
// (contacts are generated from start-of-step poses, as PhysX does).  Within a substep the ball and the arm
// therefore do not depend on each other — which is what lets the two-wave kernel run them concurrently.
// A = number of humanoids (2 for the 4-actor variant: humanoid 1's shapes are visited first, then humanoid 2's).
// bound[arm]: centre of the humanoid's broad-phase sphere (fixed with the base here; it follows the torso of the
// free-floating 27-DoF humanoid).
template <class T, int A, bool DR = false>
PP_HD void ball_substep(const StepConsts& K, Ball& b, const ArmGeom<T::kShapes> (&g)[A], const V3 (&bound)[A], const EnvDR* dr = nullptr) {
    const int M = K.ball_substeps;
    const float hb = K.hb, h = K.h;
    BallConsts k;
    k.contact_offset = K.contact_offset;
    k.bounce_threshold = K.bounce_threshold;
    k.depen_cap = K.depen_cap;
    k.r = K.ball_r;
    k.inv_kr = K.inv_kr;
    k.stick_factor = K.stick_factor;
    const float gdv = K.gdv, damp = K.damp;
    const float ground_z = K.ground_z, ground_e = K.ground_e, ground_mu = K.ground_mu;
    const ppenv_box& table = K.table;
    const ppenv_box& net = K.net;
    // broad-phase spheres of the humanoid shapes for this substep
    const ModelPaddle P = T::paddle();
    const float e_cap = K.e_cap;   // restitution_max: a scaled coefficient stays within it
    const float pad_e = DR ? fminf(K.paddle_e * dr->es, e_cap) : K.paddle_e, pad_mu = DR ? K.paddle_mu * dr->fs : K.paddle_mu;
    const float reach = k.r + k.contact_offset + 1e-4f;
    const float pad_rr = sqrtf(P.radius * P.radius + P.half_thickness * P.half_thickness) + reach;
    const float pad_r2 = pad_rr * pad_rr;
    V3 sc0[A][T::kShapes], svc[A][T::kShapes];
    float sr2[A][T::kShapes];
#pragma unroll
    for (int arm = 0; arm < A; arm++)
#pragma unroll
        for (int s = 0; s < T::kShapes; s++) {
            sc0[arm][s] = (g[arm].a[s] + g[arm].b[s]) * 0.5f;
            svc[arm][s] = (g[arm].va[s] + g[arm].vb[s]) * 0.5f;
            V3 hl = (g[arm].b[s] - g[arm].a[s]) * 0.5f;
            float rr = sqrtf(dot(hl, hl)) + T::shape(s).radius + reach;
            sr2[arm][s] = rr * rr;
        }
    PP_STAMP_AT(26);
    for (int m = 0; m < M; m++) {
        const float t = (float)m * hb;   // time since the start of the substep
        PP_STAMP_AT(27 + (m & 3));
        b.v.z += gdv;
        b.w = b.w * damp;
#ifndef PP_BALL_SKIP
#define PP_BALL_SKIP 0      // profiling builds only (tools/gpu_stamps.py): bit 0 paddle, 1 link shapes, 2 table + net, 3 ground
#endif
        if (!(PP_BALL_SKIP & 8)) contact_resolve(k, b, mk(0, 0, 1), b.p.z - ground_z - k.r, mk(0, 0, 0), ground_e, ground_mu);
        if (!(PP_BALL_SKIP & 4)) {
            contact_box(k, b, table);
            contact_box(k, b, net);
        }
#pragma unroll
        for (int arm = 0; arm < A; arm++) {
            const ArmGeom<T::kShapes>& ga = g[arm];
            V3 db = b.p - bound[arm];
#if defined(PP_ABLATE) && PP_ABLATE == 1   // profiling build: no humanoid shapes
            if (false) {
#else
            if (dot(db, db) < K.bound_r2) {
#endif
                // Broad phase: one bounding sphere per shape (centre moves linearly over the substep).  The
                // narrow phase behind it is unchanged, and a ball outside the sphere cannot touch the shape,
                // so results are identical; but some lane of a wave is nearly always near the arm, and the
                // wave then pays ~10 instructions per shape instead of the full closest-point code.
                V3 cc = madd(ga.pc, ga.vpc, t);
                V3 dpc = b.p - cc;
                if (!(PP_BALL_SKIP & 1) && dot(dpc, dpc) < pad_r2) {
                    V3 nn = madd(ga.pn, ga.pnd, t);
                    nn = nn * rsq_fast(dot(nn, nn));
                    contact_disc(k, b, cc, nn, ga.vpc, ga.pnd, P.radius, P.half_thickness, pad_e, pad_mu);
                }
#pragma unroll
                for (int s = 0; s < T::kShapes; s++) {
                    V3 dsc = b.p - madd(sc0[arm][s], svc[arm][s], t);
                    if (!(PP_BALL_SKIP & 2) && dot(dsc, dsc) < sr2[arm][s]) {
                        const float radius = T::shape(s).radius;
                        const float e = DR ? fminf(K.shape_e[s] * dr->es, e_cap) : K.shape_e[s], mu = DR ? K.shape_mu[s] * dr->fs : K.shape_mu[s];
                        if (T::shape_link(s) < 0)
                            contact_capsule(k, b, ga.a[s], ga.b[s], mk(0, 0, 0), mk(0, 0, 0), radius, e, mu);
                        else
                            contact_capsule(k, b, madd(ga.a[s], ga.va[s], t), madd(ga.b[s], ga.vb[s], t), ga.va[s], ga.vb[s], radius, e, mu);
                    }
                }
            }
        }
        b.p = madd(b.p, b.v, hb);
    }
    PP_STAMP_AT(31);
    // orientation: q <- normalize(q + h/2 (w,0) (x) q), xyzw
    float wx = b.w.x, wy = b.w.y, wz = b.w.z, x = b.quat[0], y = b.quat[1], z = b.quat[2], w = b.quat[3];
    float kq = 0.5f * h;
    float nx = x + kq * (wx * w + wy * z - wz * y);
    float ny = y + kq * (wy * w + wz * x - wx * z);
    float nz = z + kq * (wz * w + wx * y - wy * x);
    float nw = w + kq * (-wx * x - wy * y - wz * z);
    float inv = rsq_fast(nx * nx + ny * ny + nz * nz + nw * nw);
    b.quat[0] = nx * inv; b.quat[1] = ny * inv; b.quat[2] = nz * inv; b.quat[3] = nw * inv;
}

// ------------------------------------------------------ reward / reset decision
// my_quat_rotate (isaacgymenvs.utils.torch_jit_utils, xyzw): a = v(2w^2-1), b = 2w(qv x v), c = 2 qv (qv.v)
PP_HD V3 quat_rotate(const float q[4], V3 v) {
    float qw = q[3];
    V3 qv = mk(q[0], q[1], q[2]);
    float s = 2.0f * (qw * qw) - 1.0f;
    V3 cr = cross(qv, v);
    float d = dot(qv, v);
    return mk(v.x * s + cr.x * qw * 2.0f + qv.x * d * 2.0f, v.y * s + cr.y * qw * 2.0f + qv.y * d * 2.0f,
              v.z * s + cr.z * qw * 2.0f + qv.z * d * 2.0f);
}
// calc_heading_quat_inv (same module): rotation by -heading about z
PP_HD void heading_quat_inv(const float q[4], float out[4]) {
    V3 rd = quat_rotate(q, mk(1, 0, 0));
    float heading = atan2f(rd.y, rd.x);
    float theta = (-heading) / 2.0f;
    float sz = sinf(theta), w = cosf(theta);
    float nrm = fmaxf(sqrtf(sz * sz + w * w), 1e-9f);
    out[0] = 0.f; out[1] = 0.f; out[2] = sz / nrm; out[3] = w / nrm;
}

// my_quat_rotate by a heading quaternion.  calc_heading_quat_inv always returns a rotation about z, q = (0, 0, sz, w), for
// which my_quat_rotate's three terms a = v(2w^2-1), b = 2w(q x v), c = 2q(q.v) collapse to
//   x' = x(2w^2-1) - 2 w sz y,   y' = y(2w^2-1) + 2 w sz x,   z' = z((2w^2-1) + 2 sz^2) = z
// — the same fp32 products as quat_rotate's x and y (its cross-product terms with q.x = q.y = 0 are exact zeros), 6 flops
// instead of ~30 per vector; 24 vectors per observation row (TT:1657-1660,1696-1697).
PP_HD V3 heading_rotate(const float hinv[4], V3 v) {
    float qw = hinv[3], sz = hinv[2];
    float s = 2.0f * (qw * qw) - 1.0f;
    float k = sz * qw * 2.0f;
    return mk(v.x * s - v.y * k, v.y * s + v.x * k, v.z);
}

struct RewardIn {
    float humanoid_x;   // humanoid1_root_states[..., 0]
    V3 paddle;          // humanoid1_paddle_rb_states[..., 0:3]
    float pre_vx;       // pre_ball2_root_states[..., 7]
    V3 bp;              // ball position
    float vx;           // ball vx
    float power;        // sum_j |dof_force_j * dof_vel_j|
    long long progress; // already incremented (TT:1023)
};

// compute_pingpong_reward_nv TT:1105-1270 / compute_pingpong_reward T3:1080-1173 /
// compute_pingpong_reward_only_paddle TN:1115-1322.  flags is read-modify-write.
PP_HD float compute_reward(const RewardConsts& c, const RewardIn& in, uint32_t& flags, long long& reset) {
    const float Bx = in.bp.x, By = in.bp.y, Bz = in.bp.z, vx = in.vx, pre_vx = in.pre_vx;
    const float alpha = c.alpha, penalty = c.penalty, threshold = 0.1f;
    const float power_reward = -c.power_coefficient * in.power;
    const int variant = c.variant;
    uint32_t f = flags;
    float reward;
    long long die = 0;
    if (variant == PPENV_VARIANT_T3) {
        V3 dp = in.paddle - in.bp;
        float dist = sqrtf(dp.x * dp.x + dp.y * dp.y + dp.z * dp.z);
        float pos_reward = 1.0f / (1.0f + 1.5f * dist * dist);                                 // T3:1117
        float vel_reward = (pre_vx < 0.f && vx > 0.f) ? alpha * fabsf(vx) : 0.f;               // T3:1126-1128
        reward = pos_reward + power_reward + vel_reward;                                        // T3:1141
        bool missed = Bx < in.paddle.x - 1e-3f;                                                 // T3:1146
        if (missed) { reward = reward + penalty; die = 1; }                                     // T3:1149,1158
        if (Bz < threshold) die = 1;                                                            // T3:1161
    } else if (variant == PPENV_VARIANT_TT || variant == PPENV_VARIANT_T4) {   // T4:1113-1278 == TT:1105-1270
        V3 dp = in.paddle - in.bp;
        float dist = sqrtf(dp.x * dp.x + dp.y * dp.y + dp.z * dp.z);                            // TT:1144-1146
        float pos_reward = 1.0f / (1.0f + 1.5f * dist * dist);                                  // TT:1147
        bool cond = pre_vx < 0.f && vx > 0.f;                                                   // TT:1153
        float vel_reward = (cond && !(f & PPENV_FLAG_COND_CALC)) ? alpha * fabsf(vx) : 0.f;     // TT:1156-1160
        if (cond) f |= PPENV_FLAG_COND_CALC;                                                    // TT:1163
        bool missed = Bx < in.humanoid_x - 0.05f;                                               // TT:1169
        reward = missed ? 0.f + penalty : 0.f;                                                  // TT:1172-1173
        bool bounce = Bz < 0.83f && vx > 0.f && By < 0.6f && By > -0.6f;                        // TT:1184
        float hit = 0.f;
        bool early = Bx < 2.44f && bounce;
        if (early && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.not_hit_table_penalty;              // TT:1187-1191
        if (early) { f |= PPENV_FLAG_REWARD_CALC; f &= ~PPENV_FLAG_NO_BOUNCE; }                 // TT:1192,1196
        bool inx = Bx > 2.44f && Bx < 3.1f;                                                     // TT:1199
        bool good = inx && bounce && (f & PPENV_FLAG_NO_BOUNCE);
        if (good && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.hit_table_reward;                    // TT:1201-1205
        if (good) f |= PPENV_FLAG_REWARD_CALC;                                                  // TT:1206
        if (Bx >= 3.1f && vx > 0.f && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.not_hit_table_penalty;   // TT:1209-1213
        if (Bx >= 3.1f) f |= PPENV_FLAG_REWARD_CALC;                                            // TT:1214 (no vx guard)
        float net = (Bx > 1.7f && Bx < 1.8f && vx > 0.f && By < 0.4f && By > -0.4f && Bz > 0.98f && Bz < 1.14f) ? 400.f : 0.f;  // TT:1226-1244
        reward += (((pos_reward + power_reward) + vel_reward) + hit) + net;                     // TT:1251
        if (Bz < threshold) die = 1;                                                            // TT:1263
    } else {
        bool hit_paddle = pre_vx < 0.f && vx > 1.0f;                                            // TN:1160
        bool missed = (Bx < in.humanoid_x - 0.05f) || (Bx < in.paddle.x - 0.1f);                // TN:1165
        reward = (!(f & PPENV_FLAG_MISSED_CALC) && missed) ? 0.f + penalty : 0.f;               // TN:1172-1176
        if (missed) f |= PPENV_FLAG_MISSED_CALC;                                                // TN:1178
        float dy = in.paddle.y - By, dz = in.paddle.z - Bz;
        float dist = sqrtf(dy * dy + dz * dz);                                                  // TN:1185-1186
        float pos_reward = 0.f;
        if (!(f & PPENV_FLAG_COND_CALC) || (Bx < in.humanoid_x - 0.05f)) pos_reward = 1.0f * expf(-20.0f * dist * dist);  // TN:1188-1192
        float vel_reward = (hit_paddle && !(f & PPENV_FLAG_COND_CALC)) ? alpha * fabsf(vx) : 0.f;   // TN:1198-1202
        if (hit_paddle) f |= PPENV_FLAG_COND_CALC;                                              // TN:1204
        reward += (pos_reward + power_reward) + vel_reward;                                     // TN:1299
        if (Bz < threshold) reward = -800.f + reward;                                           // TN:1313-1315
    }
    flags = f;
    reset = (in.progress >= (long long)c.max_episode_length - 1) ? 1 : die;                     // TT:1265
    return reward;
}

// compute_humanoid2_pingpong_reward T4:1280-1439: TT's reward mirrored for the humanoid at the far end of the table
PP_HD float compute_reward_side2(const RewardConsts& c, const RewardIn& in, uint32_t& flags, long long& reset) {
    const float Bx = in.bp.x, By = in.bp.y, Bz = in.bp.z, vx = in.vx, pre_vx = in.pre_vx;
    uint32_t f = flags;
    V3 dp = in.paddle - in.bp;
    float dist = sqrtf(dp.x * dp.x + dp.y * dp.y + dp.z * dp.z);
    float pos_reward = 1.0f / (1.0f + 1.5f * dist * dist);                                     // T4:1305-1308
    bool cond = pre_vx > 0.f && vx < 0.f;                                                      // T4:1328
    float vel_reward = (cond && !(f & PPENV_FLAG_COND_CALC)) ? c.alpha * fabsf(vx) : 0.f;
    if (cond) f |= PPENV_FLAG_COND_CALC;
    bool missed = Bx > in.humanoid_x + 0.05f;                                                  // T4:1344
    float reward = missed ? 0.f + c.penalty : 0.f;
    bool bounce = Bz < 0.83f && vx < 0.f && By < 0.6f && By > -0.6f;                           // T4:1359
    float hit = 0.f;
    bool early = Bx > 1.06f && bounce;                                                         // T4:1363
    if (early && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.not_hit_table_penalty;
    if (early) { f |= PPENV_FLAG_REWARD_CALC; f &= ~PPENV_FLAG_NO_BOUNCE; }
    bool inx = Bx < 1.06f && Bx > 0.4f;                                                        // T4:1374
    bool good = inx && bounce && (f & PPENV_FLAG_NO_BOUNCE);
    if (good && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.hit_table_reward;
    if (good) f |= PPENV_FLAG_REWARD_CALC;
    if (Bx <= 0.4f && vx < 0.f && !(f & PPENV_FLAG_REWARD_CALC)) hit = c.not_hit_table_penalty;   // T4:1384
    if (Bx <= 0.4f) f |= PPENV_FLAG_REWARD_CALC;                                               // T4:1389
    float net = (Bx > 1.7f && Bx < 1.8f && vx < 0.f && By < 0.4f && By > -0.4f && Bz > 0.98f && Bz < 1.14f) ? 400.f : 0.f;   // T4:1401-1409
    float power_reward = -c.power_coefficient * in.power;
    reward += (((pos_reward + power_reward) + vel_reward) + hit) + net;                        // T4:1426
    flags = f;
    reset = (in.progress >= (long long)c.max_episode_length - 1) ? 1 : (Bz < 0.1f ? 1 : 0);   // T4:1431-1437
    return reward;
}

// ------------------------------------------------------------- the fused step
// A = humanoids per env (1; 2 for the 4-actor variant): arm a's joints are q[a * ND ...], its sticky flags flags[a]
template <int A>
struct EnvStateT {
    float q[A * ND], qd[A * ND], dof_force[A * ND];
    Ball ball;
    long long progress;
    uint32_t flags[A], episode;
};
using EnvState = EnvStateT<1>;

// Physics part of one VecTask.step for one env (pre_physics_step + gym.simulate):
// updates st in place, returns the pre-reset observed-body states (bodies[a * NB + j]) and pre_vx.
// actions: arm a's seven values at actions[a * ND ...] (rows A*e + a of the [A*N, 7] tensor are adjacent).
template <class T, int A, bool DR = false>
PP_HD void simulate_env(const StepConsts& K, const float* actions, EnvStateT<A>& st, BodyState* bodies, float& pre_vx, const EnvDR* dr = nullptr, uint32_t gid = 0) {
    float target[A * ND];
#pragma unroll
    for (int d = 0; d < A * ND; d++) {   // VecTask.step clamp + TT:1008 (offset/scale TT:664-665)
        float act = actions[d];
        if (DR) { if (dr->act_sigma > 0.f) act += dr->act_sigma * dr_gauss(K.seed, gid, st.episode, dr->key_progress, (uint32_t)d); }   // yaml:110-113, before the clamp
        target[d] = pd_target(act, T::drive(d % ND).lower, T::drive(d % ND).upper, K.clip_actions);
    }
    pre_vx = st.ball.v.x;   // TT:1020
    const int substeps = K.substeps;
    const float h = K.h;
    JointSave js[A][ND];
    ArmGeom<T::kShapes> g[A];
    V3 bound[A];
#pragma unroll
    for (int a = 0; a < A; a++) {
        bound[a] = ld3(K.site[a].bound_center);
        static_geometry<T>(K.site[a], g[a]);
        GeomVisitor<T> gv(g[a]);
        fk_sweep<T>(K.site[a], &st.q[a * ND], &st.qd[a * ND], js[a], gv);
    }
    PP_STAMP_AT(2);
    for (int s = 0; s < substeps; s++) {
        ball_substep<T, A, DR>(K, st.ball, g, bound, dr);                 // against the arms as they are at the substep's start
        PP_STAMP_AT(3 + 3 * s);
#pragma unroll
        for (int a = 0; a < A; a++)
            arm_substep<T, DR>(K.site[a], js[a], &st.q[a * ND], &st.qd[a * ND], &target[a * ND], h, &st.dof_force[a * ND], dr);
        PP_STAMP_AT(4 + 3 * s);
#pragma unroll
        for (int a = 0; a < A; a++) {
            if (s + 1 < substeps) {
                GeomVisitor<T> gv(g[a]);
                fk_sweep<T>(K.site[a], &st.q[a * ND], &st.qd[a * ND], js[a], gv);
            } else {
                BodyVisitor<T, false> bv(g[a], &bodies[a * NB]);
                fk_sweep<T>(K.site[a], &st.q[a * ND], &st.qd[a * ND], js[a], bv);
            }
        }
        PP_STAMP_AT(5 + 3 * s);
    }
#pragma unroll
    for (int a = 0; a < A; a++) static_body<false>(K.site[a], bodies[a * NB]);
}

// FK only (create / reset_all / refresh): observed-body states of the current dof state
template <class T>
PP_HD void bodies_of_state(const ArmSite& S, const float* q, const float* qd, BodyState* bodies) {
    JointSave js[ND];
    ArmGeom<T::kShapes> g;
    BodyVisitor<T, true> bv(g, bodies);
    fk_sweep<T>(S, q, qd, js, bv);
    static_body<true>(S, bodies[0]);
}

// initial simulation state of an env with the serve of `episode` (TT:853-867)
template <int A>
PP_HD void reset_state(const StepConsts& K, EnvStateT<A>& st, V3 serve, bool reset_dofs) {
    st.ball.p = ld3(K.ball_init_pos);
#pragma unroll
    for (int k = 0; k < 4; k++) st.ball.quat[k] = K.ball_init_quat[k];
    st.ball.v = serve;
    st.ball.w = mk(0, 0, 0);
    if (reset_dofs) {
#pragma unroll
        for (int d = 0; d < A * ND; d++) { st.q[d] = K.init_dof_pos[d % ND]; st.qd[d] = K.init_dof_vel[d % ND]; }   // T4:873: both humanoids
    }
}

// Observation row: TT:770-799 -> compute_humanoid_observations TT:1669-1708 +
// compute_pingpong_observations TT:1640-1666.  `obs` is indexed obs[k * stride].
// `hinv` = calc_heading_quat_inv(root_rot) (TT:1684).  The fused path passes the value computed once
// at create time (the pelvis is fixed); tensor-API mode computes it per env from the caller's tensor.
// compute_humanoid_observations, body part (TT:1696-1697): obs[0:30] local positions, obs[30:60] local velocities
template <int J0 = 0, int J1 = NB, class Store>
PP_HD void write_obs_bodies(const V3* body_pos, const V3* body_vel, const float hinv[4], Store& store) {
    V3 root = body_pos[0];
#pragma unroll
    for (int j = J0; j < J1; j++) {
        V3 lp = heading_rotate(hinv, body_pos[j] - root);   // TT:1696
        V3 lv = heading_rotate(hinv, body_vel[j]);          // TT:1697
        store(3 * j, lp.x); store(3 * j + 1, lp.y); store(3 * j + 2, lp.z);
        store(3 * NB + 3 * j, lv.x); store(3 * NB + 3 * j + 1, lv.y); store(3 * NB + 3 * j + 2, lv.z);
    }
}
// dof part (TT:1702-1703) and compute_pingpong_observations (TT:1640-1666): obs[60:80]
template <class Store>
PP_HD void write_obs_tail(V3 root, const float hinv[4], const float* q, const float* qd, V3 ball_p, V3 ball_v, Store& store) {
#pragma unroll
    for (int d = 0; d < ND; d++) {                       // TT:1702-1703
        store(6 * NB + d, q[d]);
        store(6 * NB + ND + d, qd[d] * 0.1f);
    }
    V3 lb = heading_rotate(hinv, ball_p - root);            // TT:1657-1659
    V3 lbv = heading_rotate(hinv, ball_v);                  // TT:1660
    store(6 * NB + 2 * ND, lb.x); store(6 * NB + 2 * ND + 1, lb.y); store(6 * NB + 2 * ND + 2, lb.z);
    store(6 * NB + 2 * ND + 3, lbv.x); store(6 * NB + 2 * ND + 4, lbv.y); store(6 * NB + 2 * ND + 5, lbv.z);
}
// Observation row: TT:770-799 -> compute_humanoid_observations TT:1669-1708 + compute_pingpong_observations TT:1640-1666
template <class Store>
PP_HD void write_obs(const V3* body_pos, const V3* body_vel, const float hinv[4], const float* q, const float* qd,
                     V3 ball_p, V3 ball_v, Store& store) {
    write_obs_bodies(body_pos, body_vel, hinv, store);
    write_obs_tail(body_pos[0], hinv, q, qd, ball_p, ball_v, store);
}

// post_physics_step for one env of the fused path (TT:1022-1039): progress, reward,
// masked reset, observations.  serve_override: used instead of the RNG when non-null.
// BODY_OBS = false: the body block obs[0:60] has already been written (by the arm wave of step_kernel_split).
// A = 2 (4-actor variant, the build's completion of T4's two-agent wiring): agent a gets rew[a] and the obs row
// stores[a]; both sides see the same ball, progress and reset rule (T4:1431-1437 on either side), so the env resets
// as one.  Like the class's only call site (T4:746-747) both reward functions receive the WHOLE dof tensors: the
// power term sums over all A*7 dofs.  Each agent observes its own seven dofs (80 = 30+30+7+7+3+3, T4:98).
template <int A, bool BODY_OBS = true, class Store>
PP_HD void post_physics_env(const StepConsts& K, uint32_t gid, EnvStateT<A>& st, const BodyState* bodies, float pre_vx,
                            const V3* serve_override, float* rew, long long& reset, Store* stores) {
    st.progress += 1;                                                            // TT:1023
    RewardIn in;
    in.humanoid_x = K.site[0].root_pos[0];
    in.paddle = bodies[NB - 1].pos;
    in.pre_vx = pre_vx;
    in.bp = st.ball.p;
    in.vx = st.ball.v.x;
    float power = 0.f;
#pragma unroll
    for (int d = 0; d < A * ND; d++) power += fabsf(st.dof_force[d] * st.qd[d]);   // TT:1246
    in.power = power;
    in.progress = st.progress;
    rew[0] = compute_reward(K.rc, in, st.flags[0], reset);
    if (A > 1) {
        RewardIn in2 = in;
        in2.humanoid_x = K.site[A - 1].root_pos[0];
        in2.paddle = bodies[(A - 1) * NB + NB - 1].pos;
        long long reset2;
        rew[A - 1] = compute_reward_side2(K.rc, in2, st.flags[A - 1], reset2);
        reset = (reset | reset2) ? 1 : 0;
    }
    if (reset) {                                                                 // TT:1034-1036 -> 847-906
        st.episode += 1;
        V3 serve = serve_override ? *serve_override : serve_velocity(K, gid, st.episode);
        reset_state(K, st, serve, K.rc.variant != PPENV_VARIANT_TN);            // TN:888-901 keeps the dof state
        st.progress = 0;                                                         // TT:902
#pragma unroll
        for (int a = 0; a < A; a++) st.flags[a] = PPENV_FLAG_NO_BOUNCE;          // TT:903-905
    }
    // TT:1039: dof / ball already show the reset state, body states are the pre-reset ones
#pragma unroll
    for (int a = 0; a < A; a++) {
        if (BODY_OBS) {
            V3 bpos[NB], bvel[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) { bpos[j] = bodies[a * NB + j].pos; bvel[j] = bodies[a * NB + j].lin; }
            write_obs_bodies(bpos, bvel, K.site[a].hinv, stores[a]);
        }
        write_obs_tail(bodies[a * NB].pos, K.site[a].hinv, &st.q[a * ND], &st.qd[a * ND], st.ball.p, st.ball.v, stores[a]);
    }
}

}  // namespace pp

// ----------------------------------------------------------------- StepConsts from the config (host)
namespace pp {
inline RewardConsts make_reward_consts(const ppenv_config& c) {
    RewardConsts r;
    r.variant = c.variant; r.max_episode_length = c.max_episode_length;
    r.alpha = c.alpha_velocity_reward; r.power_coefficient = c.power_coefficient; r.penalty = c.penalty;
    r.hit_table_reward = c.hit_table_reward; r.not_hit_table_penalty = c.not_hit_table_penalty;
    return r;
}
inline StepConsts make_step_consts(const ppenv_config& c) {
    StepConsts K;
    memset(&K, 0, sizeof K);
    K.num_envs = c.num_envs; K.env_id_offset = c.env_id_offset; K.substeps = c.substeps; K.ball_substeps = c.ball_substeps;
    K.num_arms = c.num_humanoids == 2 ? 2 : 1;
    K.seed = c.seed;
    K.rc = make_reward_consts(c);
    K.h = c.dt / (float)c.substeps;
    K.clip_actions = c.clip_actions;
    for (int a = 0; a < kMaxArms; a++) {
        const bool second = a == 1 && c.num_humanoids == 2;   // a 3-actor config mirrors site[0] into site[1] (never read)
        ArmSite& S = K.site[a];
        const float* bp = second ? c.base2_pos : c.base_pos;
        const float* br = second ? c.base2_rot : c.base_rot;
        const float* rp = second ? c.humanoid2_root_pos : c.obs_body[0].xyz;
        const float* rq = second ? c.humanoid2_root_quat : c.humanoid_root_quat;
        const float* bc = second ? c.humanoid2_bound_center : c.humanoid_bound_center;
        const ppenv_shape* sh = second ? c.shape2 : c.shape;
        for (int k = 0; k < 3; k++) { S.base_pos[k] = bp[k]; S.root_pos[k] = rp[k]; S.bound_center[k] = bc[k]; }
        for (int k = 0; k < 9; k++) S.base_rot[k] = br[k];
        for (int k = 0; k < 3; k++) S.base_grav[k] = br[6 + k] * (-c.gravity_z);   // base_rot^T (0, 0, -g): third row of base_rot
        for (int k = 0; k < 4; k++) S.root_quat[k] = rq[k];
        if (second) quat_to_rot(rq, S.root_rot);
        else for (int k = 0; k < 9; k++) S.root_rot[k] = c.obs_body[0].rot[k];
        heading_quat_inv(rq, S.hinv);   // calc_heading_quat_inv of the fixed pelvis, once (TT:1684)
        for (int s = 0; s < PPENV_MAX_SHAPES; s++)
            for (int k = 0; k < 3; k++) { S.static_a[s][k] = sh[s].a[k]; S.static_b[s][k] = sh[s].b[k]; }
    }
    K.inv_m = 1.0f / (float)c.ball_substeps;
    K.hb = K.h * K.inv_m;
    K.inv_h = 1.0f / K.h;
    K.contact_offset = c.contact_offset; K.bounce_threshold = c.bounce_threshold;
    K.depen_cap = c.max_depenetration_velocity * K.hb;
    K.ball_r = c.ball_radius;
    K.inv_kr = 1.0f / (c.ball_inertia_factor * c.ball_radius);
    K.stick_factor = c.ball_inertia_factor / (c.ball_inertia_factor + 1.0f);
    K.gdv = c.gravity_z * K.hb;
    K.damp = fmaxf(1.0f - c.ball_angular_damping * K.hb, 0.0f);
    K.ground_z = c.ground_z; K.ground_e = c.ground_restitution; K.ground_mu = c.ground_friction;
    K.table = c.table; K.net = c.net;
    K.bound_r2 = c.humanoid_bound_radius * c.humanoid_bound_radius;
    K.paddle_e = c.paddle_restitution; K.paddle_mu = c.paddle_friction; K.e_cap = c.restitution_max;
    for (int s = 0; s < PPENV_MAX_SHAPES; s++) {
        K.shape_e[s] = c.shape[s].restitution; K.shape_mu[s] = c.shape[s].friction;
    }
    for (int k = 0; k < 3; k++) { K.table_pos[k] = c.table_root_pos[k]; K.ball_init_pos[k] = c.ball_init_pos[k]; }
    for (int k = 0; k < 4; k++) { K.table_quat[k] = c.table_root_quat[k]; K.ball_init_quat[k] = c.ball_init_quat[k]; }
    K.serve_speed_lo = c.serve_speed_lo; K.serve_speed_hi = c.serve_speed_hi;
    K.serve_tilt_lo_deg = c.serve_tilt_lo_deg; K.serve_tilt_hi_deg = c.serve_tilt_hi_deg;
    K.serve_tilt_z_lo_deg = c.serve_tilt_z_lo_deg; K.serve_tilt_z_hi_deg = c.serve_tilt_z_hi_deg;
    for (int d = 0; d < PPENV_NUM_DOF; d++) { K.init_dof_pos[d] = c.init_dof_pos[d]; K.init_dof_vel[d] = c.init_dof_vel[d]; }
    return K;
}
}  // namespace pp

// ------------------------------------------------------- the compiled model and its runtime check
// PPENV_MODEL_HEADER: a build for another asset names the header isaacgym_amd/modelgen.py generated from ITS tables
// (urdf.arm_specs -> scene.use_arm_tables -> modelgen.generate); the default is the committed placeholder model.
#ifndef PPENV_MODEL_HEADER
#define PPENV_MODEL_HEADER "ppenv_model_g1.h"
#endif
#include PPENV_MODEL_HEADER

namespace pp {
// Does the runtime config describe exactly the arm model `T` was compiled from?  (bitwise on the floats)
template <class T>
inline bool model_matches(const ppenv_config& c) {
    auto same = [](const float* a, const float* b, int n) {
        for (int i = 0; i < n; i++)
            if (!(a[i] == b[i])) return false;
        return true;
    };
    if (c.num_shapes != T::kShapes || c.paddle_link != ND - 1 || c.paddle_obs_index != NB - 1) return false;
    for (int i = 0; i < ND; i++) {
        const ppenv_joint& j = c.joint[i];
        const JointKin k = T::kin(i);
        const JointDrive d = T::drive(i);
        const JointInertial in = T::inertial(i);
        if (j.axis != T::axis(i) || !same(j.origin_xyz, k.origin_xyz, 3) || !same(j.origin_rot, k.origin_rot, 9)) return false;
        if (!same(&j.lower, &d.lower, 7)) return false;
        if (j.mass != in.mass || !same(j.com, in.com, 3) || !same(j.inertia, in.inertia, 6)) return false;
    }
    for (int s = 0; s < T::kShapes; s++) {
        const ModelShape m = T::shape(s);
        if (c.shape[s].link != T::shape_link(s) || c.shape[s].radius != m.radius) return false;
        if (T::shape_link(s) >= 0 && (!same(c.shape[s].a, m.a, 3) || !same(c.shape[s].b, m.b, 3))) return false;
    }
    const ModelPaddle p = T::paddle();
    if (!same(c.paddle_center, p.center, 3) || !same(c.paddle_normal, p.normal, 3) || c.paddle_radius != p.radius ||
        c.paddle_half_thickness != p.half_thickness)
        return false;
    if (c.obs_body[0].link != -1) return false;
    for (int j = 1; j <= ND; j++) {
        const ppenv_frame& f = c.obs_body[j];
        if (f.link != j - 1 || f.xyz[0] != 0.f || f.xyz[1] != 0.f || f.xyz[2] != 0.f) return false;
        for (int k = 0; k < 9; k++)
            if (f.rot[k] != ((k % 4 == 0) ? 1.f : 0.f)) return false;
    }
    {   // step_kernel_split takes the paddle body position from the blade centre of its geometry sweep
        const JointKin t = T::tip_frame(NB - 1);
        if (!same(t.origin_xyz, p.center, 3)) return false;
    }
    for (int j = ND + 1; j < NB; j++) {
        const JointKin t = T::tip_frame(j);
        if (c.obs_body[j].link != ND - 1 || !same(c.obs_body[j].xyz, t.origin_xyz, 3) || !same(c.obs_body[j].rot, t.origin_rot, 9)) return false;
    }
    return true;
}
}  // namespace pp
