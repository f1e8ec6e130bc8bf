// ppenv_ta_device.h — per-env arithmetic of the 27-DoF variant's rigid-body step (fp32).
//
// The free-floating 28-link humanoid of tasks/humanoid_pingpong_3_actor_all_dof.py ("TA": fix_base_link = False TA:462,
// 27 position-driven dofs TA:757-774, ground plane TA:400-407) stepped with the floating-base articulated-body algorithm
// (Featherstone, RBDA table 9.4) in link coordinates.  Specification: DESIGN.md "TA physics"; the oracle restates it with
// Newton-Euler sweeps + a dense 33x33 solve in fp64 (oracle/ppenv_oracle.c, ta_simulate_env).
//
// One lane owns one env.  Unlike the 7-DoF arm (compiled-in chain, everything in registers) the tree does not fit a
// lane's registers: per-link quantities live in a lane-private column of an LDS array, addressed through the `Store`
// accessor (st(slot) -> float&), and the link loops are real loops over constant tables (all lanes of a wave visit the
// same link, so every table read is a scalar load).  tests/csrc/host_shim.cpp runs the same code over a plain array.
#pragma once

#include "ppenv_device.h"

namespace pp {
namespace ta {

constexpr int NL = PPENV_TA_NUM_LINKS;
constexpr int NDOF = PPENV_TA_NUM_DOF;

// ---- constants of the tree, derived once on the host (make_ta_consts) and read from global memory by the kernels
struct alignas(16) LinkC {                                 // 40 dwords: the quad kernel reads it from LDS in 16-byte pieces
    int32_t parent, axis, body, cfirst, ccount;            // contacts [cfirst, cfirst + ccount) sit on this link
    int32_t geo_mask, ffirst, fcount;                       // bit s: ball-collision shape s rides on this link; bit 6 paddle; bit 7 bound centre.  Welded bodies [ffirst, ffirst + fcount) of `fixed` ride on it
    float r[3], R0[9];                                      // child frame in the parent frame at q = 0
    float mass, mc[3], Io[6];                               // m, m * com, inertia about the link ORIGIN (xx yy zz xy xz yz)
    float lo, hi, kp, kd, effort, vlim, armature;
    float pad_[3];
};
static_assert(sizeof(LinkC) == 160, "LinkC layout");
struct FixedC { int32_t body, link, pad_[2]; float xyz[3], rot[9]; };
// the scalars and small tables: ~70 dwords, small enough to travel by value in a kernel argument
struct TAScal {
    int32_t num_contacts, paddle_link, bound_link, num_shapes;
    int32_t shape_link[PPENV_MAX_SHAPES];
    float shape_a[PPENV_MAX_SHAPES][3], shape_b[PPENV_MAX_SHAPES][3];
    float paddle_center[3], paddle_normal[3], bound_center[3];
    float ground_z, k_n, c_n, c_t, mu, pen_max, fade_depth, fade_force;
    float k_lim, c_lim, c_vlim;
    float gravity_z, h, clip_actions;
    int32_t substeps;
};
struct TAConsts {
    TAScal sc;
    LinkC link[NL];
    FixedC fixed[PPENV_TA_NUM_FIXED];
    float cpoint[PPENV_TA_MAX_CONTACTS][3];                 // sorted by link
};

// ---- per-lane storage layout: NL link records, then NDOF dof records
constexpr int F_E = 0, F_W = 9, F_V = 12, F_RW = 15, F_PW = 24;     // E = joint transform child -> parent (kept: three passes use it)
constexpr int F_A = 27, F_B = 33, F_D = 42, F_PN = 48, F_PF = 51;    // articulated inertia blocks A (sym) B D (sym), bias force
constexpr int F_UA = 27, F_UB = 30, F_DINV = 33, F_U = 34, F_AW = 35, F_AV = 38;   // overlay A..: written once the link's inertia is consumed
constexpr int LINK_STRIDE = 54;
constexpr int DOF_BASE = NL * LINK_STRIDE;
constexpr int G_Q = 0, G_QD = 1, G_TARGET = 2, G_FORCE = 3;
constexpr int DOF_STRIDE = 4;
constexpr int NUM_SLOTS = DOF_BASE + NDOF * DOF_STRIDE;

template <class Store> PP_HD V3 ldv(Store& st, int slot) { return mk(st(slot), st(slot + 1), st(slot + 2)); }
template <class Store> PP_HD void stv(Store& st, int slot, V3 v) { st(slot) = v.x; st(slot + 1) = v.y; st(slot + 2) = v.z; }
template <class Store> PP_HD M3 ldm3(Store& st, int slot) { M3 r; for (int k = 0; k < 9; k++) r.m[k] = st(slot + k); return r; }
template <class Store> PP_HD void stm3(Store& st, int slot, const M3& a) { for (int k = 0; k < 9; k++) st(slot + k) = a.m[k]; }
template <class Store> PP_HD S3 lds3(Store& st, int slot) { S3 r = {st(slot), st(slot + 1), st(slot + 2), st(slot + 3), st(slot + 4), st(slot + 5)}; return r; }
template <class Store> PP_HD void sts3(Store& st, int slot, const S3& a) { st(slot) = a.xx; st(slot + 1) = a.yy; st(slot + 2) = a.zz; st(slot + 3) = a.xy; st(slot + 4) = a.xz; st(slot + 5) = a.yz; }

PP_HD M3 joint_rot_rt(const float* r0, int ax, float c, float s) {   // joint_rot with a run-time axis
    return ax == 0 ? joint_rot<0>(r0, c, s) : (ax == 1 ? joint_rot<1>(r0, c, s) : joint_rot<2>(r0, c, s));
}
PP_HD M3 quat_to_m3(const float q[4]) { M3 r; quat_to_rot(q, r.m); return r; }
PP_HD S3 sym_from(const float* p) { S3 r = {p[0], p[1], p[2], p[3], p[4], p[5]}; return r; }
PP_HD M3 cross_mat_scaled(V3 a) { M3 r = {{0.f, -a.z, a.y, a.z, 0.f, -a.x, -a.y, a.x, 0.f}}; return r; }   // a^
// r^ S r^T for symmetric S (the angular block a point inertia S at offset r adds)
PP_HD S3 rsr(V3 r, const S3& s) {
    M3 sm = from_sym(s);
    // T = r^ S : rows are r x cols... (r^ S)_ij = (r x col_j(S))_i
    V3 t0 = cross(r, col(sm, 0)), t1 = cross(r, col(sm, 1)), t2 = cross(r, col(sm, 2));
    M3 t = {{t0.x, t1.x, t2.x, t0.y, t1.y, t2.y, t0.z, t1.z, t2.z}};
    // (T r^T)_ij = -(T r^)_ij = (r x row_i(T))_j  [row_i(T) r^T = r x row_i(T)]
    V3 u0 = cross(r, row(t, 0)), u1 = cross(r, row(t, 1)), u2 = cross(r, row(t, 2));
    S3 o = {u0.x, u1.y, u2.z, u0.y, u0.z, u1.z};
    return o;
}

// world pose + link-frame twist of the pelvis from its root_states row (pos3 quat4 linvel3 angvel3, world)
struct BaseState { V3 p; float quat[4]; V3 vw, ww; };

// ---- register-level primitives (shared by the one-lane-per-env passes below and the four-lanes-per-env kernel)
struct ArtI { S3 A; M3 B; S3 D; V3 pn, pf; };        // articulated inertia [A B; B^T D] and bias force (angular, linear)
struct JointOut { V3 ua, ub; float dinv, u; };       // what the outward pass needs from the inward one

// Rigid-body inertia, velocity-product and gravity bias, ground contacts of one link (link coordinates, about its origin).
// cpoint: the contact-point table (C.cpoint, or a copy of it nearer to the lanes).
PP_HD ArtI link_dynamics(const TAScal& C, const LinkC& L, const float (*cpoint)[3], const M3& Rw, V3 pw, V3 w, V3 v) {
    S3 A = sym_from(L.Io);
    V3 mc = ld3(L.mc);
    const float m = L.mass;
    V3 h_ang = mul(A, w) + cross(mc, v);
    V3 h_lin = v * m - cross(mc, w);
    const V3p wxh = cross(w, pk(h_ang, h_lin));             // pairs that take the same operator are packed (ppenv_device.h, V3p / S3p)
    V3 pn = lo(wxh) + cross(v, h_lin);
    V3 pf = hi(wxh);
    V3 gb = mk(Rw.m[6], Rw.m[7], Rw.m[8]) * C.gravity_z;   // R^T (0, 0, g)
    pn = pn - cross(mc, gb);
    pf = pf - gb * m;
    M3 B = cross_mat_scaled(mc);
    S3 D = {m, m, m, 0.f, 0.f, 0.f};
    const V3 zb = mk(Rw.m[6], Rw.m[7], Rw.m[8]);           // world z axis in link coordinates
    for (int k = L.cfirst; k < L.cfirst + L.ccount; k++) {
        V3 r = ld3(cpoint[k]);
        V3 pc = pw + mul(Rw, r);
        float pen = C.ground_z - pc.z;
        if (!(pen > 0.f)) continue;
        V3 vloc = v + cross(w, r);
        V3 vw = mul(Rw, vloc);
        // every switch of the contact law is a ramp (see the oracle): damper fades in with depth, implicit terms with force
        float wfade = pen < C.fade_depth ? pen / C.fade_depth : 1.f;
        float fn0 = C.k_n * fminf(pen, C.pen_max) - wfade * C.c_n * vw.z;
        if (!(fn0 > 0.f)) continue;
        float gfade = fn0 < C.fade_force ? fn0 / C.fade_force : 1.f;
        float vt = sqrtf(vw.x * vw.x + vw.y * vw.y);
        float dt = gfade * C.c_t;
        if (dt * vt > C.mu * fn0) dt = C.mu * fn0 / vt;
        V3 fb = tmul(Rw, mk(-dt * vw.x, -dt * vw.y, fn0));
        pn = pn - cross(r, fb);
        pf = pf - fb;
        // h D_b = h (dt 1 + (dn - dt) z z^T), J = [-r^, 1]
        float dn = gfade * (wfade * C.c_n + C.h * C.k_n);
        float hd = C.h * dt, hz = C.h * (dn - dt);
        S3 Dp = {hd + hz * zb.x * zb.x, hd + hz * zb.y * zb.y, hd + hz * zb.z * zb.z, hz * zb.x * zb.y, hz * zb.x * zb.z, hz * zb.y * zb.z};
        add_sym(D, Dp);
        add_sym(A, rsr(r, Dp));
        M3 dm = from_sym(Dp);   // B += r^ Dp
        V3 b0 = cross(r, col(dm, 0)), b1 = cross(r, col(dm, 1)), b2 = cross(r, col(dm, 2));
        B.m[0] += b0.x; B.m[1] += b1.x; B.m[2] += b2.x;
        B.m[3] += b0.y; B.m[4] += b1.y; B.m[5] += b2.y;
        B.m[6] += b0.z; B.m[7] += b1.z; B.m[8] += b2.z;
    }
    ArtI o = {A, B, D, pn, pf};
    return o;
}
PP_HD void add_art(ArtI& a, const ArtI& b) {
    add_sym(a.A, b.A); add_sym(a.D, b.D);
    for (int k = 0; k < 9; k++) a.B.m[k] += b.B.m[k];
    a.pn = a.pn + b.pn; a.pf = a.pf + b.pf;
}

// pose and link-frame twist of a link from its parent's (in: parent's Rw pw w v; out: the link's)
PP_HD void link_kinematics(const LinkC& L, float q, float qd, M3& Rw, V3& pw, V3& w, V3& v, M3& E) {
    float s, c;
    sincos_joint(q, s, c);
    const int ax = L.axis;
    E = joint_rot_rt(L.R0, ax, c, s);
    V3 r = ld3(L.r);
    const V3p wv = tmul(E, pk(w, v + cross(w, r)));
    V3 wn = lo(wv), vn = hi(wv);
    if (ax == 0) wn.x += qd; else if (ax == 1) wn.y += qd; else wn.z += qd;
    pw = pw + mul(Rw, r);
    Rw = mul(Rw, E);
    w = wn; v = vn;
}

// drive + limit torques of dof d at the start of a substep (implicit PD with its explicit part clamped to the effort
// limit; limit spring-damper and velocity-cap damper implicit): torque and the joint-space inertia they add
PP_HD void joint_torque(const TAScal& C, const LinkC& L, float q, float qd, float target, float& tau, float& arm) {
    const float h = C.h;
    float err = target - q;
    tau = fminf(fmaxf(L.kp * (err - h * qd) - L.kd * qd, -L.effort), L.effort);   // implicit PD, explicit part within the effort limit
    arm = L.armature + h * L.kd + h * h * L.kp;
    float over = q > L.hi ? q - L.hi : (q < L.lo ? q - L.lo : 0.f);
    if (over != 0.f) {
        // quadratic toe over the first 0.01 rad, tangent stiffness in the implicit term (see the oracle)
        float x = fabsf(over), ramp = fminf(x * 100.f, 1.f);
        float phi = x < 0.01f ? x * x * 50.f : x - 0.005f;
        float kt = C.k_lim * ramp;
        tau += -copysignf(C.k_lim * phi, over) - kt * h * qd - ramp * C.c_lim * qd;
        arm += h * (ramp * C.c_lim + h * kt);
    }
    if (fabsf(qd) > L.vlim) {
        float ex = fabsf(qd) - L.vlim;
        tau += -C.c_vlim * copysignf(ex, qd);
        arm += h * C.c_vlim * fminf(ex, 1.f);                  // fades in over 1 rad/s
    }
}

// One joint of the inward pass (RBDA 9.4 with the drive's implicit terms on the joint diagonal).
// I: in = the link's articulated inertia and bias (own + children, link coordinates); out = its contribution to the
// parent (parent coordinates, about the parent's origin).
PP_HD void inward_step(const TAScal& C, const LinkC& L, ArtI& I, V3 w, V3 v, const M3& E, float q, float qd, float target, JointOut& jo) {
    const int ax = L.axis;
    float tau, arm;
    joint_torque(C, L, q, qd, target, tau, arm);
    S3& A = I.A; S3& D = I.D; M3& B = I.B;
    V3 ua = symcol(A, ax), ub = row(B, ax);
    float dinv = rcp_fast(symdiag(A, ax) + arm);
    float u = tau - comp(I.pn, ax);
    jo.ua = ua; jo.ub = ub; jo.dinv = dinv; jo.u = u;
    S3p AD = pk(A, D);
    const V3p uab = pk(ua, ub);
    sym_rank1_sub(AD, uab, dinv);
    V3 uad = ua * dinv;
    for (int r = 0; r < 3; r++) {
        float k = comp(uad, r);
        B.m[3 * r] -= k * ub.x; B.m[3 * r + 1] -= k * ub.y; B.m[3 * r + 2] -= k * ub.z;
    }
    V3 e = unit(ax);
    const V3p c = cross(pk(w, v), e) * qd;                  // c = v x S qd, angular and linear half
    V3 cw = lo(c), cv = hi(c);
    float ud = u * dinv;
    const V3p pa = pk(I.pn, I.pf) + mul(AD, c) + pk(mul(B, cv), tmul(B, cw)) + uab * ud;   // p + I^a c + U u / D
    V3 r = ld3(L.r);
    const S3p ADr = rot_sym(E, AD);
    const S3 Ar = lo(ADr), Dr = hi(ADr);
    M3 Br = mul_t(mul(E, B), E);
    const V3p nfr = mul(E, pa);
    V3 nr = lo(nfr), fr = hi(nfr);
    M3 Dm = from_sym(Dr);
    M3 Bp;
    for (int j = 0; j < 3; j++) {
        V3 x = cross(r, col(Dm, j));
        Bp.m[j] = Br.m[j] + x.x; Bp.m[3 + j] = Br.m[3 + j] + x.y; Bp.m[6 + j] = Br.m[6 + j] + x.z;
    }
    const V3p w0 = cross(r, pk(row(Bp, 0), row(Br, 0))), w1 = cross(r, pk(row(Bp, 1), row(Br, 1))), w2 = cross(r, pk(row(Bp, 2), row(Br, 2)));
    S3 Ap = {Ar.xx + w0.x[0] + w0.x[1], Ar.yy + w1.y[0] + w1.y[1], Ar.zz + w2.z[0] + w2.z[1],
             Ar.xy + w1.x[0] + w0.y[1], Ar.xz + w2.x[0] + w0.z[1], Ar.yz + w2.y[0] + w1.z[1]};
    I.A = Ap; I.B = Bp; I.D = Dr;
    I.pn = nr + cross(r, fr); I.pf = fr;
}

// One joint of the outward pass: the link's acceleration from its parent's (aw, av: in = parent's, out = the link's), and
// the joint's semi-implicit Euler update with the reported drive torque.
PP_HD void outward_step(const TAScal& C, const LinkC& L, const M3& E, V3 w, V3 v, const JointOut& jo, V3& aw, V3& av,
                        float target, float& q, float& qd, float& force) {
    const int ax = L.axis;
    V3 r = ld3(L.r), e = unit(ax);
    const V3p a2 = tmul(E, pk(aw, av + cross(aw, r))) + cross(pk(w, v), e) * qd;   // (angular, linear)
    V3 aw2 = lo(a2), av2 = hi(a2);
    const f2 ud = dot(pk(jo.ua, jo.ub), a2);
    float qdd = (jo.u - ud[0] - ud[1]) * jo.dinv;
    if (ax == 0) aw2.x += qdd; else if (ax == 1) aw2.y += qdd; else aw2.z += qdd;
    aw = aw2; av = av2;
    float err = target - q;
    float vn = qd + C.h * qdd;
    force = fminf(fmaxf(L.kp * (err - C.h * vn) - L.kd * vn, -L.effort), L.effort);   // reported within the actuator's limit
    q = q + C.h * vn;
    qd = vn;
}

// ---- one lane per env: the passes over the lane's store
// Pass 1: kinematics base -> tips.  Fills c s w v Rw pw of every link; with DYN also the link's inertia / bias record.
template <class Store>
PP_HD void store_art(Store& st, int o, const ArtI& I) {
    sts3(st, o + F_A, I.A); stm3(st, o + F_B, I.B); sts3(st, o + F_D, I.D);
    stv(st, o + F_PN, I.pn); stv(st, o + F_PF, I.pf);
}
template <class Store>
PP_HD ArtI load_art(Store& st, int o) {
    ArtI I = {lds3(st, o + F_A), ldm3(st, o + F_B), lds3(st, o + F_D), ldv(st, o + F_PN), ldv(st, o + F_PF)};
    return I;
}
template <bool DYN, class Store>
PP_HD void pass_kinematics(const TAConsts& C, Store& st, const BaseState& b) {
    {
        M3 Rw = quat_to_m3(b.quat);
        V3 w = tmul(Rw, b.ww), v = tmul(Rw, b.vw);
        stv(st, F_W, w); stv(st, F_V, v); stm3(st, F_RW, Rw); stv(st, F_PW, b.p);
        if (DYN) store_art(st, 0, link_dynamics(C.sc, C.link[0], C.cpoint, Rw, b.p, w, v));
    }
    for (int i = 1; i < NL; i++) {
        const LinkC& L = C.link[i];
        const int o = i * LINK_STRIDE, po = L.parent * LINK_STRIDE;
        float q = st(DOF_BASE + (i - 1) * DOF_STRIDE + G_Q), qd = st(DOF_BASE + (i - 1) * DOF_STRIDE + G_QD);
        M3 Rw = ldm3(st, po + F_RW);
        V3 pw = ldv(st, po + F_PW), w = ldv(st, po + F_W), v = ldv(st, po + F_V);
        M3 E;
        link_kinematics(L, q, qd, Rw, pw, w, v, E);
        stm3(st, o + F_E, E);
        stv(st, o + F_W, w); stv(st, o + F_V, v); stm3(st, o + F_RW, Rw); stv(st, o + F_PW, pw);
        if (DYN) store_art(st, o, link_dynamics(C.sc, L, C.cpoint, Rw, pw, w, v));
    }
}

// Pass 2: articulated inertias tips -> base
template <class Store>
PP_HD void pass_inertia(const TAConsts& C, Store& st) {
    for (int i = NL - 1; i >= 1; i--) {
        const LinkC& L = C.link[i];
        const int o = i * LINK_STRIDE, po = L.parent * LINK_STRIDE, dofo = DOF_BASE + (i - 1) * DOF_STRIDE;
        ArtI I = load_art(st, o);
        JointOut jo;
        inward_step(C.sc, L, I, ldv(st, o + F_W), ldv(st, o + F_V), ldm3(st, o + F_E), st(dofo + G_Q), st(dofo + G_QD), st(dofo + G_TARGET), jo);
        // the link's own record is consumed: keep what pass 3 needs in its place
        stv(st, o + F_UA, jo.ua); stv(st, o + F_UB, jo.ub); st(o + F_DINV) = jo.dinv; st(o + F_U) = jo.u;
        ArtI P = load_art(st, po);
        add_art(P, I);
        store_art(st, po, P);
    }
}

PP_HD M3 inv_sym(const S3& s) {   // inverse of a symmetric positive-definite 3x3
    float c00 = s.yy * s.zz - s.yz * s.yz, c01 = s.xz * s.yz - s.xy * s.zz, c02 = s.xy * s.yz - s.xz * s.yy;
    float c11 = s.xx * s.zz - s.xz * s.xz, c12 = s.xy * s.xz - s.xx * s.yz, c22 = s.xx * s.yy - s.xy * s.xy;
    float idet = 1.0f / (s.xx * c00 + s.xy * c01 + s.xz * c02);
    M3 r = {{c00 * idet, c01 * idet, c02 * idet, c01 * idet, c11 * idet, c12 * idet, c02 * idet, c12 * idet, c22 * idet}};
    return r;
}

// base acceleration from [A B; B^T D] [alpha; a] = -[pn; pf]
PP_HD void solve_base_art(const ArtI& I, V3& alpha, V3& a) {
    M3 Di = inv_sym(I.D);
    M3 BDi = mul(I.B, Di);                     // B D^-1
    M3 S = mul_t(BDi, I.B);                    // B D^-1 B^T
    S3 Sc = {I.A.xx - S.m[0], I.A.yy - S.m[4], I.A.zz - S.m[8], I.A.xy - 0.5f * (S.m[1] + S.m[3]), I.A.xz - 0.5f * (S.m[2] + S.m[6]),
             I.A.yz - 0.5f * (S.m[5] + S.m[7])};
    V3 rhs = mul(BDi, I.pf) - I.pn;
    alpha = mul(inv_sym(Sc), rhs);
    a = -mul(Di, I.pf + tmul(I.B, alpha));
}
template <class Store>
PP_HD void solve_base(Store& st, V3& alpha, V3& a) { solve_base_art(load_art(st, 0), alpha, a); }

// Pass 3: accelerations base -> tips, and the joints' semi-implicit Euler update (q, qd, reported drive torque)
template <class Store>
PP_HD void pass_accelerate(const TAConsts& C, Store& st, V3 alpha0, V3 a0) {
    stv(st, F_AW, alpha0); stv(st, F_AV, a0);
    for (int i = 1; i < NL; i++) {
        const LinkC& L = C.link[i];
        const int o = i * LINK_STRIDE, po = L.parent * LINK_STRIDE, dofo = DOF_BASE + (i - 1) * DOF_STRIDE;
        V3 aw = ldv(st, po + F_AW), av = ldv(st, po + F_AV);
        JointOut jo = {ldv(st, o + F_UA), ldv(st, o + F_UB), st(o + F_DINV), st(o + F_U)};
        float q = st(dofo + G_Q), qd = st(dofo + G_QD), force;
        outward_step(C.sc, L, ldm3(st, o + F_E), ldv(st, o + F_W), ldv(st, o + F_V), jo, aw, av, st(dofo + G_TARGET), q, qd, force);
        stv(st, o + F_AW, aw); stv(st, o + F_AV, av);
        st(dofo + G_FORCE) = force; st(dofo + G_Q) = q; st(dofo + G_QD) = qd;
    }
}

// base: semi-implicit Euler in world coordinates (classical acceleration of the origin = R (a + w x v))
PP_HD void integrate_base_regs(const TAScal& C, const M3& Rw, V3 wb, V3 vb, BaseState& b, V3 alpha, V3 a) {
    V3 aw = mul(Rw, a + cross(wb, vb)), alw = mul(Rw, alpha);
    const float h = C.h;
    b.vw = madd(b.vw, aw, h);
    b.ww = madd(b.ww, alw, h);
    b.p = madd(b.p, b.vw, h);
    float x = b.quat[0], y = b.quat[1], z = b.quat[2], w = b.quat[3], kq = 0.5f * h;
    float nx = x + kq * (b.ww.x * w + b.ww.y * z - b.ww.z * y);
    float ny = y + kq * (b.ww.y * w + b.ww.z * x - b.ww.x * z);
    float nz = z + kq * (b.ww.z * w + b.ww.x * y - b.ww.y * x);
    float nw = w + kq * (-b.ww.x * x - b.ww.y * y - b.ww.z * z);
    float inv = rsq_fast(nx * nx + ny * ny + nz * nz + nw * nw);
    b.quat[0] = nx * inv; b.quat[1] = ny * inv; b.quat[2] = nz * inv; b.quat[3] = nw * inv;
}
template <class Store>
PP_HD void integrate_base(const TAConsts& C, Store& st, BaseState& b, V3 alpha, V3 a) {
    integrate_base_regs(C.sc, ldm3(st, F_RW), ldv(st, F_W), ldv(st, F_V), b, alpha, a);
}

// world position / velocity of a point fixed in link `link` (after pass_kinematics)
template <class Store>
PP_HD void point_state(Store& st, int link, V3 r, V3& p, V3& v) {
    const int o = link * LINK_STRIDE;
    M3 Rw = ldm3(st, o + F_RW);
    p = ldv(st, o + F_PW) + mul(Rw, r);
    v = mul(Rw, ldv(st, o + F_V) + cross(ldv(st, o + F_W), r));
}

// ball-collision geometry of the humanoid at the start of a substep
template <int NSHAPES, class Store>
PP_HD void collision_geometry(const TAConsts& C, Store& st, ArmGeom<NSHAPES>& g, V3& bound) {
    const int po = C.sc.paddle_link * LINK_STRIDE;
    M3 Rp = ldm3(st, po + F_RW);
    point_state(st, C.sc.paddle_link, ld3(C.sc.paddle_center), g.pc, g.vpc);
    g.pn = mul(Rp, ld3(C.sc.paddle_normal));
    g.pnd = cross(mul(Rp, ldv(st, po + F_W)), g.pn);
    for (int s = 0; s < NSHAPES; s++) {
        point_state(st, C.sc.shape_link[s], ld3(C.sc.shape_a[s]), g.a[s], g.va[s]);
        point_state(st, C.sc.shape_link[s], ld3(C.sc.shape_b[s]), g.b[s], g.vb[s]);
    }
    V3 dummy;
    point_state(st, C.sc.bound_link, ld3(C.sc.bound_center), bound, dummy);
}

// the compiled shape radii / paddle blade are the 3-actor model's; here every shape rides on a moving link
struct ModelG1TA : ModelG1 {
    PP_HD static constexpr int shape_link(int) { return 0; }
};

// One pre_physics_step + gym.simulate for one env.  q / qd / target live in the store; base and ball in registers.
template <class Store>
PP_HD void simulate_env_ta(const TAConsts& C, const StepConsts& K, Store& st, BaseState& base, Ball& ball) {
    for (int s = 0; s < C.sc.substeps; s++) {
        pass_kinematics<true>(C, st, base);
        ArmGeom<ModelG1TA::kShapes> g[1];
        V3 bound[1];
        collision_geometry(C, st, g[0], bound[0]);          // poses and velocities at the start of the substep
        pass_inertia(C, st);
        V3 alpha, a;
        solve_base(st, alpha, a);
        pass_accelerate(C, st, alpha, a);                    // also integrates the joints
        integrate_base(C, st, base, alpha, a);               // uses the start-of-substep base frame (still in the store)
        ball_substep<ModelG1TA, 1>(K, ball, g, bound);
    }
}

// rigid_body_states rows [42][13] of one env (gym.refresh_rigid_body_state_tensor) after pass_kinematics<false>
template <class Store>
PP_HD void write_body_row(float* row, const M3& R, V3 p, V3 lin, V3 ang) {
    float q[4];
    rot_to_quat(R, q);
    row[0] = p.x; row[1] = p.y; row[2] = p.z; row[3] = q[0]; row[4] = q[1]; row[5] = q[2]; row[6] = q[3];
    row[7] = lin.x; row[8] = lin.y; row[9] = lin.z; row[10] = ang.x; row[11] = ang.y; row[12] = ang.z;
}
template <class Store>
PP_HD void write_body_states(const TAConsts& C, Store& st, float* rb) {
    for (int i = 0; i < NL; i++) {
        const int o = i * LINK_STRIDE;
        M3 Rw = ldm3(st, o + F_RW);
        write_body_row<Store>(rb + C.link[i].body * 13, Rw, ldv(st, o + F_PW), mul(Rw, ldv(st, o + F_V)), mul(Rw, ldv(st, o + F_W)));
    }
    for (int f = 0; f < PPENV_TA_NUM_FIXED; f++) {
        const FixedC& F = C.fixed[f];
        const int o = F.link * LINK_STRIDE;
        M3 Rw = ldm3(st, o + F_RW);
        V3 p, v;
        point_state(st, F.link, ld3(F.xyz), p, v);
        write_body_row<Store>(rb + F.body * 13, mul(Rw, ldm(F.rot)), p, v, mul(Rw, ldv(st, o + F_W)));
    }
}

}  // namespace ta
}  // namespace pp

// ------------------------------------------------------------------ TAConsts from the model (host)
namespace pp {
namespace ta {
inline bool make_ta_consts(const ppenv_config& scene, const ppenv_ta_model& m, TAConsts& C, const char** why) {
    memset(&C, 0, sizeof C);
    *why = "";
    if (m.num_contacts < 0 || m.num_contacts > PPENV_TA_MAX_CONTACTS) { *why = "num_contacts out of range"; return false; }
    if (scene.num_shapes != ModelG1TA::kShapes) { *why = "scene.num_shapes does not match the compiled shape set"; return false; }
    int nc = 0;
    for (int i = 0; i < NL; i++) {
        const ppenv_ta_link& s = m.link[i];
        LinkC& L = C.link[i];
        if (i == 0 ? s.parent != -1 : (s.parent < 0 || s.parent >= i)) { *why = "links must list parents before children"; return false; }
        if (i > 0 && (s.axis < 0 || s.axis > 2)) { *why = "joint axis must be 0, 1 or 2"; return false; }
        if (!(s.mass > 0.f) || s.body < 0 || s.body >= PPENV_NUM_HUMANOID_BODIES) { *why = "link mass must be > 0 and body in [0, 40)"; return false; }
        L.parent = s.parent; L.axis = s.axis; L.body = s.body;
        for (int k = 0; k < 3; k++) { L.r[k] = s.origin_xyz[k]; L.mc[k] = s.mass * s.com[k]; }
        for (int k = 0; k < 9; k++) L.R0[k] = s.origin_rot[k];
        L.mass = s.mass;
        const float cx = s.com[0], cy = s.com[1], cz = s.com[2], cc = cx * cx + cy * cy + cz * cz, mm = s.mass;
        L.Io[0] = s.inertia[0] + mm * (cc - cx * cx); L.Io[1] = s.inertia[1] + mm * (cc - cy * cy); L.Io[2] = s.inertia[2] + mm * (cc - cz * cz);
        L.Io[3] = s.inertia[3] - mm * cx * cy; L.Io[4] = s.inertia[4] - mm * cx * cz; L.Io[5] = s.inertia[5] - mm * cy * cz;
        L.lo = s.lower; L.hi = s.upper; L.kp = s.kp; L.kd = s.kd; L.effort = s.effort; L.vlim = s.vel_limit; L.armature = s.armature;
        L.cfirst = nc;
        for (int k = 0; k < m.num_contacts; k++)
            if (m.contact_link[k] == i) {
                for (int t = 0; t < 3; t++) C.cpoint[nc][t] = m.contact_point[k][t];
                nc++;
            }
        L.ccount = nc - L.cfirst;
    }
    if (nc != m.num_contacts) { *why = "a contact point names a link that does not exist"; return false; }
    for (int f = 0; f < PPENV_TA_NUM_FIXED; f++) {
        const ppenv_ta_fixed& s = m.fixed[f];
        if (s.link < 0 || s.link >= NL || s.body < 0 || s.body >= PPENV_NUM_HUMANOID_BODIES) { *why = "welded body: link / body index out of range"; return false; }
    }
    int nf = 0;   // sorted by link, like the contact points: a link visits only the bodies welded to it
    for (int i = 0; i < NL; i++) {
        C.link[i].ffirst = nf;
        for (int f = 0; f < PPENV_TA_NUM_FIXED; f++) {
            const ppenv_ta_fixed& s = m.fixed[f];
            if (s.link != i) continue;
            C.fixed[nf].body = s.body; C.fixed[nf].link = s.link;
            C.fixed[nf].pad_[0] = C.fixed[nf].pad_[1] = 0;
            for (int k = 0; k < 3; k++) C.fixed[nf].xyz[k] = s.xyz[k];
            for (int k = 0; k < 9; k++) C.fixed[nf].rot[k] = s.rot[k];
            nf++;
        }
        C.link[i].fcount = nf - C.link[i].ffirst;
    }
    C.sc.num_contacts = m.num_contacts; C.sc.paddle_link = scene.paddle_link; C.sc.bound_link = m.bound_link; C.sc.num_shapes = scene.num_shapes;
    if (C.sc.paddle_link < 0 || C.sc.paddle_link >= NL || C.sc.bound_link < 0 || C.sc.bound_link >= NL) { *why = "paddle_link / bound_link out of range"; return false; }
    for (int s = 0; s < scene.num_shapes; s++) {
        if (scene.shape[s].link < 0 || scene.shape[s].link >= NL) { *why = "scene.shape[].link must index the 28-link tree"; return false; }
        if (scene.shape[s].radius != ModelG1TA::shape(s).radius) { *why = "scene.shape[].radius differs from the compiled shape set"; return false; }
        C.sc.shape_link[s] = scene.shape[s].link;
        for (int k = 0; k < 3; k++) { C.sc.shape_a[s][k] = scene.shape[s].a[k]; C.sc.shape_b[s][k] = scene.shape[s].b[k]; }
    }
    for (int s = 0; s < scene.num_shapes; s++) C.link[C.sc.shape_link[s]].geo_mask |= 1 << s;
    C.link[C.sc.paddle_link].geo_mask |= 1 << 6;
    C.link[C.sc.bound_link].geo_mask |= 1 << 7;
    for (int k = 0; k < 3; k++) { C.sc.paddle_center[k] = scene.paddle_center[k]; C.sc.paddle_normal[k] = scene.paddle_normal[k]; C.sc.bound_center[k] = m.bound_center[k]; }
    C.sc.ground_z = m.ground_z; C.sc.k_n = m.foot_stiffness; C.sc.c_n = m.foot_damping; C.sc.c_t = m.foot_tangent_damping; C.sc.mu = m.foot_friction;
    C.sc.pen_max = m.contact_max_penetration; C.sc.fade_depth = m.contact_fade_depth; C.sc.fade_force = m.contact_fade_force;
    if (!(C.sc.fade_depth > 0.f) || !(C.sc.fade_force > 0.f)) { *why = "contact_fade_depth / contact_fade_force must be positive"; return false; }
    C.sc.k_lim = m.limit_stiffness; C.sc.c_lim = m.limit_damping; C.sc.c_vlim = m.vel_limit_damping;
    C.sc.gravity_z = scene.gravity_z; C.sc.substeps = scene.substeps; C.sc.h = scene.dt / (float)scene.substeps; C.sc.clip_actions = scene.clip_actions;
    if (scene.substeps < 1 || scene.substeps > 16 || !(scene.dt > 0.f)) { *why = "dt / substeps out of range"; return false; }
    return true;
}
}  // namespace ta
}  // namespace pp
