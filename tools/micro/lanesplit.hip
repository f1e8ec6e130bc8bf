// Micro-benchmark (round 3): ONE link of the articulated-body algorithm's inward pass (RBDA table 7.1, the body of
// pp::aba_solve's pass-2 loop — link inertia + bias force, U / D / u, rank-1 update, articulated bias, transform to the parent) in two forms:
//
//   scalar       one lane per env, 64 envs per wave: the product's form.  The model is compile-time literals, so exact zeros, identity
//                frames and the symmetry of the 3x3 blocks fold away.
//   distributed  "split one env's spatial algebra over lanes while every lane visits the same link": the lanes of a quad hold the
//                x / y / z ROW of every 3x3 block and the x / y / z component of every 3-vector (lane 3 idles), 16 envs per wave;
//                rows meet through DPP quad permutes (v_mov_b32_dpp / v_mul_f32_dpp / v_add_f32_dpp — hipcc folds a DPP operand into
//                VOP2 multiplies and adds but not into v_fma / v_fmac, which is what a broadcast-accumulate needs).
//
// Each form runs REPS dependent link steps (a link's output is the next one's child input) on ONE wave per workgroup and reports shader
// cycles per link step; the first step's results are compared between the forms.  Build / run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -fno-signed-zeros -ffinite-math-only \
//         -I isaacgym_amd/csrc tools/micro/lanesplit.hip -o /tmp/lanesplit && /tmp/lanesplit
// Static VALU counts of the two forms: tools/micro/lanesplit_counts.sh (disassembly of the same build).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ppenv_device.h"
#include "ppenv_model_g1.h"

using namespace pp;
typedef ModelG1 T;

// ---------------------------------------------------------------------------------------------------- scalar form (the product's arithmetic)
struct ChildS { S3 cA, cD; M3 cB; V3 cn, cf; };
struct OutS { V3 ua, ub; float dinv, u; };

template <int I>
__device__ __forceinline__ void scalar_link(V3 w, V3 v, float qd, float c, float s, float tau, float arm_eff, ChildS& ch, OutS& o) {
    const JointKin JK = T::kin(I);
    const JointInertial J = T::inertial(I);
    constexpr int ax = T::axis(I);
    float m = J.mass;
    V3 cm = ld3(J.com);
    V3 mc = cm * m;
    float cc = dot(cm, cm);
    S3 A = {J.inertia[0] + m * (cc - cm.x * cm.x), J.inertia[1] + m * (cc - cm.y * cm.y), J.inertia[2] + m * (cc - cm.z * cm.z),
            J.inertia[3] - m * cm.x * cm.y, J.inertia[4] - m * cm.x * cm.z, J.inertia[5] - m * cm.y * cm.z};
    V3 h_ang = mul(A, w) + cross(mc, v);
    V3 h_lin = v * m - cross(mc, w);
    V3 pn = cross(w, h_ang) + cross(v, h_lin) + ch.cn;
    V3 pf = cross(w, h_lin) + ch.cf;
    add_sym(A, ch.cA);
    M3 B = {{ch.cB.m[0], ch.cB.m[1] - mc.z, ch.cB.m[2] + mc.y, ch.cB.m[3] + mc.z, ch.cB.m[4], ch.cB.m[5] - mc.x, ch.cB.m[6] - mc.y, ch.cB.m[7] + mc.x, ch.cB.m[8]}};
    S3 D = {ch.cD.xx + m, ch.cD.yy + m, ch.cD.zz + m, ch.cD.xy, ch.cD.xz, ch.cD.yz};
    V3 ua = symcol(A, ax), ub = row(B, ax);
    float dinv = rcp_fast(symdiag(A, ax) + arm_eff);
    float u = tau - comp(pn, ax);
    o.ua = ua; o.ub = ub; o.dinv = dinv; o.u = u;
    sym_rank1_sub(A, ua, dinv);
    sym_rank1_sub(D, ub, dinv);
    V3 uad = ua * dinv;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float k = comp(uad, r);
        B.m[3 * r] -= k * ub.x; B.m[3 * r + 1] -= k * ub.y; B.m[3 * r + 2] -= k * ub.z;
    }
    V3 e = unit(ax);
    V3 cw = cross(w, e) * qd, cv = cross(v, e) * qd;
    float ud = u * dinv;
    V3 pan = pn + mul(A, cw) + mul(B, cv) + ua * ud;
    V3 paf = pf + tmul(B, cw) + mul(D, cv) + ub * ud;
    M3 E = ax == 0 ? joint_rot<0>(JK.origin_rot, c, s) : (ax == 1 ? joint_rot<1>(JK.origin_rot, c, s) : joint_rot<2>(JK.origin_rot, c, s));
    V3 r = ld3(JK.origin_xyz);
    S3 Ar = rot_sym(E, A), Dr = rot_sym(E, D);
    M3 Br = mul_t(mul(E, B), E);
    V3 nr = mul(E, pan), fr = mul(E, paf);
    M3 Dm = from_sym(Dr);
    M3 Bp;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        V3 x = cross(r, col(Dm, j));
        Bp.m[j] = Br.m[j] + x.x; Bp.m[3 + j] = Br.m[3 + j] + x.y; Bp.m[6 + j] = Br.m[6 + j] + x.z;
    }
    V3 wp0 = cross(r, row(Bp, 0)), wp1 = cross(r, row(Bp, 1)), wp2 = cross(r, row(Bp, 2));
    V3 wb0 = cross(r, row(Br, 0)), wb1 = cross(r, row(Br, 1)), wb2 = cross(r, row(Br, 2));
    ch.cA.xx = Ar.xx + wp0.x + wb0.x; ch.cA.yy = Ar.yy + wp1.y + wb1.y; ch.cA.zz = Ar.zz + wp2.z + wb2.z;
    ch.cA.xy = Ar.xy + wp1.x + wb0.y; ch.cA.xz = Ar.xz + wp2.x + wb0.z; ch.cA.yz = Ar.yz + wp2.y + wb1.z;
    ch.cB = Bp; ch.cD = Dr;
    ch.cn = nr + cross(r, fr); ch.cf = fr;
}

// ------------------------------------------------------------------------------------------------------------------- distributed form
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int J> __device__ __forceinline__ float bc(float x) { return dpp<QP(J, J, J, J)>(x); }          // lane J of the quad
__device__ __forceinline__ float r1(float x) { return dpp<QP(1, 2, 0, 3)>(x); }                            // x_{k+1}
__device__ __forceinline__ float r2(float x) { return dpp<QP(2, 0, 1, 3)>(x); }                            // x_{k+2}
template <int AX> __device__ __forceinline__ float swp(float x) {                                          // the two lanes a rotation about AX mixes
    return AX == 0 ? dpp<QP(0, 2, 1, 3)>(x) : (AX == 1 ? dpp<QP(2, 1, 0, 3)>(x) : dpp<QP(1, 0, 2, 3)>(x));
}
struct Row { float c[3]; };                      // lane k: row k of a 3x3 block
struct ChildD { Row A, B, D; float cn, cf; };
struct OutD { float ua, ub, dinv, u; };

__device__ __forceinline__ float LC(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }   // a per-lane constant
__device__ __forceinline__ float sel3(int k, float x0, float x1, float x2) { return k == 0 ? x0 : (k == 1 ? x1 : x2); }

// y = Rot(AX, angle) x with alpha / beta the per-lane coefficients (alpha_k = 1 on the axis, c off it; beta = -s, +s on the two mixed lanes)
template <int AX> __device__ __forceinline__ float rot_vec(float al, float be, float x) { return al * x + be * swp<AX>(x); }
// X' = Rot X Rot^T for a row-distributed block
template <int AX>
__device__ __forceinline__ Row rot_block(float al, float be, float c, float s, Row X) {
    constexpr int a = (AX + 1) % 3, b = (AX + 2) % 3;
    Row L;
#pragma unroll
    for (int j = 0; j < 3; j++) L.c[j] = al * X.c[j] + be * swp<AX>(X.c[j]);
    Row R;
    R.c[AX] = L.c[AX];
    R.c[a] = c * L.c[a] - s * L.c[b];
    R.c[b] = s * L.c[a] + c * L.c[b];
    return R;
}

// per-lane constants of joint I (in the kernel these live in VGPRs for the whole launch)
template <int I>
struct LaneConsts {
    float Io[3], Smc[3], mD[3];     // rows of the link's rotational inertia about its origin, of skew(m c), of m * 1
    float mc1, mc2;                 // (m c)_{k+1}, (m c)_{k+2}
    float r1c, r2c, Rk[3];          // r_{k+1}, r_{k+2}, row k of skew(r)
    float sg;                       // sign pattern of (x cross e_axis): +1 on lane axis+1, -1 on lane axis+2, 0 on the axis
    float onax;                     // 1 on the axis lane
    float al0, be0;                 // the constant frame rotation R0 (about x), as alpha / beta
    __device__ __forceinline__ void init(int k) {
        const JointKin JK = T::kin(I);
        const JointInertial J = T::inertial(I);
        constexpr int ax = T::axis(I);
        const float m = J.mass, cx = J.com[0], cy = J.com[1], cz = J.com[2], cc = cx * cx + cy * cy + cz * cz;
        const float A[3][3] = {{J.inertia[0] + m * (cc - cx * cx), J.inertia[3] - m * cx * cy, J.inertia[4] - m * cx * cz},
                               {J.inertia[3] - m * cx * cy, J.inertia[1] + m * (cc - cy * cy), J.inertia[5] - m * cy * cz},
                               {J.inertia[4] - m * cx * cz, J.inertia[5] - m * cy * cz, J.inertia[2] + m * (cc - cz * cz)}};
        const float mx = m * cx, my = m * cy, mz = m * cz;
        const float S[3][3] = {{0, -mz, my}, {mz, 0, -mx}, {-my, mx, 0}};
#pragma unroll
        for (int j = 0; j < 3; j++) { Io[j] = LC(k, A[0][j], A[1][j], A[2][j]); Smc[j] = LC(k, S[0][j], S[1][j], S[2][j]); mD[j] = LC(k, j == 0 ? m : 0.f, j == 1 ? m : 0.f, j == 2 ? m : 0.f); }
        mc1 = LC(k, my, mz, mx); mc2 = LC(k, mz, mx, my);
        const float rx = JK.origin_xyz[0], ry = JK.origin_xyz[1], rz = JK.origin_xyz[2];
        r1c = LC(k, ry, rz, rx); r2c = LC(k, rz, rx, ry);
        Rk[0] = LC(k, 0.f, rz, -ry); Rk[1] = LC(k, -rz, 0.f, rx); Rk[2] = LC(k, ry, -rx, 0.f);
        constexpr int a = (ax + 1) % 3, b = (ax + 2) % 3;
        sg = LC(k, a == 0 ? 1.f : (b == 0 ? -1.f : 0.f), a == 1 ? 1.f : (b == 1 ? -1.f : 0.f), a == 2 ? 1.f : (b == 2 ? -1.f : 0.f));
        onax = LC(k, ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f);
        // R0 = Rot_x(theta0): rows (1,0,0), (0,c0,-s0), (0,s0,c0)
        const float c0 = JK.origin_rot[4], s0 = JK.origin_rot[7];
        al0 = LC(k, 1.f, c0, c0); be0 = LC(k, 0.f, -s0, s0);
    }
};
template <int I> constexpr bool frame_is_identity() { return T::kin(I).origin_rot[4] == 1.0f && T::kin(I).origin_rot[7] == 0.0f; }

template <int I>
__device__ __forceinline__ void dist_link(int k, const LaneConsts<I>& L, float w, float v, float qd, float c, float s, float tau, float arm_eff, ChildD& ch, OutD& o) {
    constexpr int ax = T::axis(I);
    const JointKin JK = T::kin(I);
    const float m = T::inertial(I).mass;
    // link momentum and bias force
    const float w1 = r1(w), w2 = r2(w), v1 = r1(v), v2 = r2(v);
    const float h_ang = L.Io[0] * bc<0>(w) + L.Io[1] * bc<1>(w) + L.Io[2] * bc<2>(w) + (L.mc1 * v2 - L.mc2 * v1);
    const float h_lin = m * v - (L.mc1 * w2 - L.mc2 * w1);
    float pn = (w1 * r2(h_ang) - w2 * r1(h_ang)) + (v1 * r2(h_lin) - v2 * r1(h_lin)) + ch.cn;
    float pf = (w1 * r2(h_lin) - w2 * r1(h_lin)) + ch.cf;
    Row A, B, D;
#pragma unroll
    for (int j = 0; j < 3; j++) { A.c[j] = ch.A.c[j] + L.Io[j]; B.c[j] = ch.B.c[j] + L.Smc[j]; D.c[j] = ch.D.c[j] + L.mD[j]; }
    // U = I^A S: column `ax` of [A; B^T]
    const float ua = A.c[ax];
    const float ub = sel3(k, bc<ax>(B.c[0]), bc<ax>(B.c[1]), bc<ax>(B.c[2]));
    const float dinv = rcp_fast(bc<ax>(A.c[ax]) + arm_eff);
    const float u = tau - bc<ax>(pn);
    o.ua = ua; o.ub = ub; o.dinv = dinv; o.u = u;
    // rank-1 update
    const float uad = ua * dinv, ubd = ub * dinv;
    {
        const float ua0 = bc<0>(ua), ua1 = bc<1>(ua), ua2 = bc<2>(ua), ub0 = bc<0>(ub), ub1 = bc<1>(ub), ub2 = bc<2>(ub);
        A.c[0] -= uad * ua0; A.c[1] -= uad * ua1; A.c[2] -= uad * ua2;
        B.c[0] -= uad * ub0; B.c[1] -= uad * ub1; B.c[2] -= uad * ub2;
        D.c[0] -= ubd * ub0; D.c[1] -= ubd * ub1; D.c[2] -= ubd * ub2;
    }
    // c = v x S qd (zero on the axis lane), articulated bias
    const float sq = L.sg * qd;
    const float cw = sq * swp<ax>(w), cv = sq * swp<ax>(v);
    const float ud = u * dinv;
    constexpr int a = (ax + 1) % 3, b = (ax + 2) % 3;
    float pan = pn + A.c[a] * bc<a>(cw) + A.c[b] * bc<b>(cw) + B.c[a] * bc<a>(cv) + B.c[b] * bc<b>(cv) + ua * ud;
    // B^T cw: lane j scales its row by its own cw, the column sums go round the quad, lane k keeps column k
    const float t0 = B.c[0] * cw, t1 = B.c[1] * cw, t2 = B.c[2] * cw;
    const float s0 = t0 + r1(t0) + r2(t0), s1 = t1 + r1(t1) + r2(t1), s2 = t2 + r1(t2) + r2(t2);
    float paf = pf + sel3(k, s0, s1, s2) + D.c[a] * bc<a>(cv) + D.c[b] * bc<b>(cv) + ub * ud;
    // to the parent's coordinates: E = R0 Rot(ax, q)
    const float al = L.onax + (1.f - L.onax) * c, be = L.sg * -s;   // beta: -s on lane axis+1, +s on lane axis+2
    Row Ar = rot_block<ax>(al, be, c, s, A), Br = rot_block<ax>(al, be, c, s, B), Dr = rot_block<ax>(al, be, c, s, D);
    float nr = rot_vec<ax>(al, be, pan), fr = rot_vec<ax>(al, be, paf);
    if (!frame_is_identity<I>()) {
        const float c0 = JK.origin_rot[4], s0r = JK.origin_rot[7];
        Ar = rot_block<0>(L.al0, L.be0, c0, s0r, Ar); Br = rot_block<0>(L.al0, L.be0, c0, s0r, Br); Dr = rot_block<0>(L.al0, L.be0, c0, s0r, Dr);
        nr = rot_vec<0>(L.al0, L.be0, nr); fr = rot_vec<0>(L.al0, L.be0, fr);
    }
    // shift the origin by r
    Row Bp;
#pragma unroll
    for (int j = 0; j < 3; j++) Bp.c[j] = Br.c[j] + (L.r1c * r2(Dr.c[j]) - L.r2c * r1(Dr.c[j]));       // + (r x column j of Dr)_k
    const float rx = JK.origin_xyz[0], ry = JK.origin_xyz[1], rz = JK.origin_xyz[2];
    const float t3[3] = {ry * Br.c[2] - rz * Br.c[1], rz * Br.c[0] - rx * Br.c[2], rx * Br.c[1] - ry * Br.c[0]};   // (r x row_k(Br))_c: local, literal r
    Row Ap;
    Ap.c[0] = Ar.c[0] + (L.Rk[0] * bc<0>(Bp.c[0]) + L.Rk[1] * bc<0>(Bp.c[1]) + L.Rk[2] * bc<0>(Bp.c[2])) + t3[0];   // (r x row_c(Bp))_k
    Ap.c[1] = Ar.c[1] + (L.Rk[0] * bc<1>(Bp.c[0]) + L.Rk[1] * bc<1>(Bp.c[1]) + L.Rk[2] * bc<1>(Bp.c[2])) + t3[1];
    Ap.c[2] = Ar.c[2] + (L.Rk[0] * bc<2>(Bp.c[0]) + L.Rk[1] * bc<2>(Bp.c[1]) + L.Rk[2] * bc<2>(Bp.c[2])) + t3[2];
    ch.A = Ap; ch.B = Bp; ch.D = Dr;
    ch.cn = nr + (L.r1c * r2(fr) - L.r2c * r1(fr));
    ch.cf = fr;
}

// ------------------------------------------------------------------------------------------------------------------------------ harness
constexpr int kOut = 32;   // floats per env: cA 9 (full), cB 9, cD 9, cn 3 -> 30 (cf = last 3 of a second block)

template <int I>
__global__ __launch_bounds__(64) void scalar_kernel(const float* in, float* out, unsigned long long* cyc, int reps) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const float* p = in + (size_t)e * 48;
    V3 w = ld3(p), v = ld3(p + 3);
    float qd = p[6], q = p[7], tau = p[8], arm = 0.05f;
    ChildS ch;
    ch.cA = S3{p[9], p[10], p[11], p[12], p[13], p[14]};
    ch.cD = S3{p[15], p[16], p[17], p[18], p[19], p[20]};
#pragma unroll
    for (int j = 0; j < 9; j++) ch.cB.m[j] = p[21 + j];
    ch.cn = ld3(p + 30); ch.cf = ld3(p + 33);
    float s, c;
    sincos_joint(q, s, c);
    OutS o;
    scalar_link<I>(w, v, qd, c, s, tau, arm, ch, o);     // step 1: compared with the distributed form
    float* r = out + (size_t)e * 40;
    const float res[40] = {ch.cA.xx, ch.cA.xy, ch.cA.xz, ch.cA.xy, ch.cA.yy, ch.cA.yz, ch.cA.xz, ch.cA.yz, ch.cA.zz, ch.cB.m[0], ch.cB.m[1], ch.cB.m[2], ch.cB.m[3],
                           ch.cB.m[4], ch.cB.m[5], ch.cB.m[6], ch.cB.m[7], ch.cB.m[8], ch.cD.xx, ch.cD.xy, ch.cD.xz, ch.cD.xy, ch.cD.yy, ch.cD.yz, ch.cD.xz, ch.cD.yz,
                           ch.cD.zz, ch.cn.x, ch.cn.y, ch.cn.z, ch.cf.x, ch.cf.y, ch.cf.z, o.ua.x, o.ua.y, o.ua.z, o.ub.x, o.ub.y, o.ub.z, o.dinv};
#pragma unroll
    for (int j = 0; j < 40; j++) r[j] = res[j];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; it++) {
        // keep the chain bounded and dependent: the child's inertia is what the previous link handed up, damped
        ch.cA.xx *= 0.5f; ch.cA.yy *= 0.5f; ch.cA.zz *= 0.5f; ch.cn = ch.cn * 0.5f;
        w.x += 1e-3f * ch.cf.x;
        scalar_link<I>(w, v, qd, c, s, tau, arm, ch, o);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[(size_t)gridDim.x * 64 * 40 + e] = ch.cA.xx + ch.cn.x + o.u;     // keeps the loop alive
}

template <int I>
__global__ __launch_bounds__(64) void dist_kernel(const float* in, float* out, unsigned long long* cyc, int reps) {
    const int k = threadIdx.x & 3;
    const int e = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int kk = k < 3 ? k : 2;                      // lane 3 mirrors lane 2's data (its results are never read)
    const float* p = in + (size_t)e * 48;
    float w = p[kk], v = p[3 + kk];
    float qd = p[6], q = p[7], tau = p[8], arm = 0.05f;
    // symmetric blocks are stored as six numbers (xx yy zz xy xz yz): row kk of the full matrix
    auto symrow = [&](const float* s6, int j) {
        const int idx[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};
        return s6[idx[kk][j]];
    };
    ChildD ch;
#pragma unroll
    for (int j = 0; j < 3; j++) { ch.A.c[j] = symrow(p + 9, j); ch.D.c[j] = symrow(p + 15, j); ch.B.c[j] = p[21 + 3 * kk + j]; }
    ch.cn = p[30 + kk]; ch.cf = p[33 + kk];
    float s, c;
    sincos_joint(q, s, c);
    LaneConsts<I> L;
    L.init(kk);
    OutD o;
    dist_link<I>(kk, L, w, v, qd, c, s, tau, arm, ch, o);
    if (k < 3) {
        float* r = out + (size_t)e * 40;
#pragma unroll
        for (int j = 0; j < 3; j++) { r[3 * k + j] = ch.A.c[j]; r[9 + 3 * k + j] = ch.B.c[j]; r[18 + 3 * k + j] = ch.D.c[j]; }
        r[27 + k] = ch.cn; r[30 + k] = ch.cf; r[33 + k] = o.ua; r[36 + k] = o.ub;
        if (k == 0) r[39] = o.dinv;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; it++) {
        ch.A.c[0] *= LC(kk, 0.5f, 1.f, 1.f); ch.A.c[1] *= LC(kk, 1.f, 0.5f, 1.f); ch.A.c[2] *= LC(kk, 1.f, 1.f, 0.5f); ch.cn *= 0.5f;
        w += LC(kk, 1e-3f, 0.f, 0.f) * ch.cf;
        dist_link<I>(kk, L, w, v, qd, c, s, tau, arm, ch, o);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[(size_t)gridDim.x * 16 * 40 + blockIdx.x * 64 + threadIdx.x] = ch.A.c[0] + ch.cn + o.u;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int I>
void run(int nenv, int reps) {
    std::vector<float> h((size_t)nenv * 48);
    srand(1234 + I);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int e = 0; e < nenv; e++) {
        float* p = &h[(size_t)e * 48];
        for (int j = 0; j < 6; j++) p[j] = 3.f * rnd();              // w, v
        p[6] = 5.f * rnd(); p[7] = 1.5f * rnd(); p[8] = 10.f * rnd();   // qd, q, tau
        // child inertia: a plausible spd rotational block, small coupling, mass block
        float a[6] = {0.02f + 0.01f * rnd(), 0.03f + 0.01f * rnd(), 0.025f + 0.01f * rnd(), 0.003f * rnd(), 0.003f * rnd(), 0.003f * rnd()};
        float d[6] = {1.2f + 0.2f * rnd(), 1.1f + 0.2f * rnd(), 1.3f + 0.2f * rnd(), 0.05f * rnd(), 0.05f * rnd(), 0.05f * rnd()};
        for (int j = 0; j < 6; j++) { p[9 + j] = a[j]; p[15 + j] = d[j]; }
        for (int j = 0; j < 9; j++) p[21 + j] = 0.05f * rnd();
        for (int j = 0; j < 6; j++) p[30 + j] = 2.f * rnd();
    }
    float *din, *dout_s, *dout_d;
    unsigned long long *cyc_s, *cyc_d;
    const int gs = nenv / 64, gd = nenv / 16;
    CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout_s, (size_t)nenv * 41 * 4 + 64 * 4)); CK(hipMalloc(&dout_d, (size_t)nenv * 40 * 4 + (size_t)gd * 64 * 4));
    CK(hipMalloc(&cyc_s, gs * 8)); CK(hipMalloc(&cyc_d, gd * 8));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_s = 0, ms_d = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(scalar_kernel<I>, dim3(gs), dim3(64), 0, 0, din, dout_s, cyc_s, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_s, e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(dist_kernel<I>, dim3(gd), dim3(64), 0, 0, din, dout_d, cyc_d, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_d, e0, e1));
    }
    std::vector<float> os((size_t)nenv * 40), od((size_t)nenv * 40);
    std::vector<unsigned long long> cs(gs), cd(gd);
    CK(hipMemcpy(os.data(), dout_s, os.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(od.data(), dout_d, od.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cs.data(), cyc_s, gs * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cd.data(), cyc_d, gd * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t j = 0; j < os.size(); j++) worst = fmax(worst, fabs((double)os[j] - od[j]) / (1e-3 + fabs((double)os[j])));
    auto med = [](std::vector<unsigned long long>& x) { std::sort(x.begin(), x.end()); return (double)x[x.size() / 2]; };
    const double ms = med(cs) / reps, md = med(cd) / reps;
    printf("joint %d (axis %d, frame %s): scalar %7.1f cycles / link step (64 envs per wave, %d waves), distributed %7.1f (16 envs per wave, %d waves): "
           "chain ratio %.2f;  kernel %.3f vs %.3f ms at %d envs;  first-step agreement: worst rel. diff %.1e\n",
           I, T::axis(I), frame_is_identity<I>() ? "identity" : "rotated ", ms, gs, md, gd, md / ms, ms_s, ms_d, nenv, worst);
    CK(hipFree(din)); CK(hipFree(dout_s)); CK(hipFree(dout_d)); CK(hipFree(cyc_s)); CK(hipFree(cyc_d));
}

int main(int argc, char** argv) {
    const int nenv = argc > 1 ? atoi(argv[1]) : 16384, reps = argc > 2 ? atoi(argv[2]) : 64;
    printf("one link of the ABA inward pass, %d dependent repetitions, one wave per workgroup\n", reps);
    run<1>(nenv, reps);
    run<3>(nenv, reps);
    run<4>(nenv, reps);
    run<6>(nenv, reps);
    return 0;
}
