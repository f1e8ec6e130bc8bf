// ppenv_ta_sim.hip — the 27-DoF variant's rigid-body step on gfx950: kernel + C ABI (include/ppenv.h, ppenv_ta_simulate).
//
// Mapping: one lane per env, 16 envs per workgroup.  The 28-link tree does not fit a lane's registers, so each lane keeps its
// link / dof records in its own column of an LDS array [slot][16] (1620 slots -> 101 KB per workgroup; consecutive lanes hit
// consecutive banks).  At BASELINE config 5's size (4096 envs per GPU) that is 256 workgroups, one per CU; the link loops are
// uniform across the wave, so the model tables in global memory are read with scalar loads.  The simulation state is the
// caller's Isaac-Gym-layout tensors (AoS, ~0.7 KB per env in and 2.9 KB out): the step is bound by its ~1e5 dependent fp32
// operations per env, not by these bytes.
#undef PP_STAMP   // the phase stamps of diagnostic builds belong to ppenv.hip's kernels
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ppenv_ta_device.h"
#include "ppenv_ta_task.h"
#include "ppenv_ta_chain.h"

using namespace pp;
using namespace pp::ta;
using std::min;

void ppenv_set_error(const char* msg);   // ppenv.hip
int ppenv_ta_clear_counts(int n, uint32_t* flags_dev, uint32_t* any_reset_dev, void* stream);   // ppenv_ta.hip

namespace {
// Diagnostic builds only (-DTA_STAMP, tools/gpu_ta_stamps.py): shader-clock stamps of the quad kernel's phases, lane 0 of each workgroup.
#if defined(TA_STAMP)
__device__ unsigned long long ta_stamp_buf[4096 * 32];
#define TA_STAMP_AT(k)                                                                     \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long t_;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");     \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) ta_stamp_buf[blockIdx.x * 32 + (k)] = t_; \
    } while (0)
#else
#define TA_STAMP_AT(k) do { } while (0)
#endif

constexpr int kTaLanes = 16;

struct LdsStore {
    float* col;   // &lds[lane]
    __device__ __forceinline__ float& operator()(int slot) { return col[slot * kTaLanes]; }
};

__device__ __forceinline__ void load_base(const float* root, BaseState& b) {
    b.p = mk(root[0], root[1], root[2]);
    for (int k = 0; k < 4; k++) b.quat[k] = root[3 + k];
    b.vw = mk(root[7], root[8], root[9]);
    b.ww = mk(root[10], root[11], root[12]);
}

template <bool STEP>
__global__ __launch_bounds__(kTaLanes) void ta_sim_kernel(const TAConsts* __restrict__ Cp, const StepConsts K, int n, const float* __restrict__ actions,
                                                          float* root_states, float* dof_states, float* __restrict__ rb_states,
                                                          float* __restrict__ dof_force, float* __restrict__ pre_vx) {
    __shared__ float lds[NUM_SLOTS * kTaLanes];
    const int lane = threadIdx.x;
    const int e = blockIdx.x * kTaLanes + lane;
    if (e >= n) return;   // no barrier below: lanes are independent
    const TAConsts& C = *Cp;
    LdsStore st{&lds[lane]};
    float* root = root_states + (size_t)e * 39;
    float* dofs = dof_states + (size_t)e * 2 * NDOF;
    BaseState base;
    load_base(root, base);
    for (int d = 0; d < NDOF; d++) {
        st(DOF_BASE + d * DOF_STRIDE + G_Q) = dofs[2 * d];
        st(DOF_BASE + d * DOF_STRIDE + G_QD) = dofs[2 * d + 1];
    }
    if (STEP) {
        for (int d = 0; d < NDOF; d++) {   // VecTask.step clamp + TA:1131 (offset / scale TA:729-733)
            const LinkC& L = C.link[d + 1];
            st(DOF_BASE + d * DOF_STRIDE + G_TARGET) = pd_target(actions[(size_t)e * NDOF + d], L.lo, L.hi, C.sc.clip_actions);
            st(DOF_BASE + d * DOF_STRIDE + G_FORCE) = 0.f;
        }
        float* bl = root + 26;
        Ball ball;
        ball.p = mk(bl[0], bl[1], bl[2]);
        for (int k = 0; k < 4; k++) ball.quat[k] = bl[3 + k];
        ball.v = mk(bl[7], bl[8], bl[9]);
        ball.w = mk(bl[10], bl[11], bl[12]);
        pre_vx[e] = ball.v.x;                                                   // TA:1143
        simulate_env_ta(C, K, st, base, ball);
        root[0] = base.p.x; root[1] = base.p.y; root[2] = base.p.z;
        for (int k = 0; k < 4; k++) root[3 + k] = base.quat[k];
        root[7] = base.vw.x; root[8] = base.vw.y; root[9] = base.vw.z;
        root[10] = base.ww.x; root[11] = base.ww.y; root[12] = base.ww.z;
        bl[0] = ball.p.x; bl[1] = ball.p.y; bl[2] = ball.p.z;
        for (int k = 0; k < 4; k++) bl[3 + k] = ball.quat[k];
        bl[7] = ball.v.x; bl[8] = ball.v.y; bl[9] = ball.v.z; bl[10] = ball.w.x; bl[11] = ball.w.y; bl[12] = ball.w.z;
        for (int d = 0; d < NDOF; d++) {
            dofs[2 * d] = st(DOF_BASE + d * DOF_STRIDE + G_Q);
            dofs[2 * d + 1] = st(DOF_BASE + d * DOF_STRIDE + G_QD);
            dof_force[(size_t)e * NDOF + d] = st(DOF_BASE + d * DOF_STRIDE + G_FORCE);
        }
    }
    // gym.refresh_rigid_body_state_tensor: [42][13] rows of the new state
    pass_kinematics<false>(C, st, base);
    float* rb = rb_states + (size_t)e * PPENV_NUM_BODIES * 13;
    write_body_states(C, st, rb);
    for (int k = 0; k < 13; k++) { rb[40 * 13 + k] = root[13 + k]; rb[41 * 13 + k] = root[26 + k]; }
}

// ---------------------------------------------------------------------------------------------------------------
// Four lanes per env.  The humanoid is four limbs on two hubs (pelvis: legs + the waist chain; torso: arms), so a quad of
// lanes walks the four limbs in parallel: lane role 0 = left leg, 1 = right leg, 2 = waist + left arm, 3 = waist + right arm
// (the 3-link waist chain is computed redundantly by both arm lanes).  A lane carries its chain's running quantities (pose,
// twist, articulated inertia, acceleration) in registers and keeps only what a later pass needs in its LDS column; the two
// hub accumulations travel through quad shuffles.  Serial depth: 10 joint visits per pass instead of 27, 64 active lanes
// per wave instead of 16.  The chains below are the topology of scene.build_ta_model (checked at create time; any other
// tree runs on ta_sim_kernel).
constexpr int kQuadEnvs = 16;      // envs per 64-lane workgroup
constexpr int kChainLen = 10;
// A link's record is contiguous per lane, 44 floats = eleven 16-byte groups: E9 w3 | v3 - | A6 B9 D6 pn3 pf3 - (later ua3 ub3 dinv u over the
// A.. slots), so that the compiler reads and writes it as ds_*_b128 (a lane stride of 44 floats is conflict-free for 128-bit accesses:
// eight lanes x four banks cover the 32 banks once).  Field-major storage needed one ds_read_b32 and its address per value.
constexpr int kRec = 44;
constexpr int R_E = 0, R_W = 9, R_V = 12, R_ART = 16, R_JO = 16;
// link at position k of role r's chain (-1 past its end):  0: 1..6   1: 7..12   2: 13..22   3: 13 14 15 23..27
// (arithmetic, not a table: a per-lane table read would be a vector memory load in every loop iteration)
__device__ __forceinline__ int chain_link(int role, int k) {
    if (role < 2) return k < 6 ? 1 + 6 * role + k : -1;
    if (role == 2) return 13 + k;
    return k < 3 ? 13 + k : (k < 8 ? 20 + k : -1);
}
constexpr int kHubPos = 2;         // chain position of the torso on the two arm lanes: the arms merge before it is processed

struct QuadRec {
    float* col;   // &s_rec[0][lane][0]
    __device__ __forceinline__ float& operator()(int k, int f) { return col[k * (kRec * 64) + f]; }
};
typedef float f4v __attribute__((ext_vector_type(4)));
// whole 16-byte groups of a record
__device__ __forceinline__ void rec_store4(QuadRec& rc, int k, int f, float a, float b, float c, float d) {
    *reinterpret_cast<f4v*>(&rc(k, f)) = f4v{a, b, c, d};
}
__device__ __forceinline__ void rec_store_art(QuadRec& rc, int k, const ArtI& I) {
    const float v[28] = {I.A.xx, I.A.yy, I.A.zz, I.A.xy, I.A.xz, I.A.yz, I.B.m[0], I.B.m[1], I.B.m[2], I.B.m[3], I.B.m[4], I.B.m[5], I.B.m[6], I.B.m[7], I.B.m[8],
                         I.D.xx, I.D.yy, I.D.zz, I.D.xy, I.D.xz, I.D.yz, I.pn.x, I.pn.y, I.pn.z, I.pf.x, I.pf.y, I.pf.z, 0.f};
#pragma unroll
    for (int t = 0; t < 28; t += 4) rec_store4(rc, k, R_ART + t, v[t], v[t + 1], v[t + 2], v[t + 3]);
}
__device__ __forceinline__ ArtI rec_load_art(QuadRec& rc, int k) {
    float v[27];
#pragma unroll
    for (int t = 0; t < 27; t++) v[t] = rc(k, R_ART + t);
    ArtI I = {{v[0], v[1], v[2], v[3], v[4], v[5]}, {{v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14]}},
              {v[15], v[16], v[17], v[18], v[19], v[20]}, mk(v[21], v[22], v[23]), mk(v[24], v[25], v[26])};
    return I;
}
__device__ __forceinline__ ArtI art_zero() {
    ArtI I = {{0, 0, 0, 0, 0, 0}, {{0, 0, 0, 0, 0, 0, 0, 0, 0}}, {0, 0, 0, 0, 0, 0}, mk(0, 0, 0), mk(0, 0, 0)};
    return I;
}
__device__ __forceinline__ ArtI art_shfl_xor(const ArtI& I, int m) {
    ArtI o;
    o.A.xx = __shfl_xor(I.A.xx, m); o.A.yy = __shfl_xor(I.A.yy, m); o.A.zz = __shfl_xor(I.A.zz, m);
    o.A.xy = __shfl_xor(I.A.xy, m); o.A.xz = __shfl_xor(I.A.xz, m); o.A.yz = __shfl_xor(I.A.yz, m);
#pragma unroll
    for (int t = 0; t < 9; t++) o.B.m[t] = __shfl_xor(I.B.m[t], m);
    o.D.xx = __shfl_xor(I.D.xx, m); o.D.yy = __shfl_xor(I.D.yy, m); o.D.zz = __shfl_xor(I.D.zz, m);
    o.D.xy = __shfl_xor(I.D.xy, m); o.D.xz = __shfl_xor(I.D.xz, m); o.D.yz = __shfl_xor(I.D.yz, m);
    o.pn = mk(__shfl_xor(I.pn.x, m), __shfl_xor(I.pn.y, m), __shfl_xor(I.pn.z, m));
    o.pf = mk(__shfl_xor(I.pf.x, m), __shfl_xor(I.pf.y, m), __shfl_xor(I.pf.z, m));
    return o;
}
__device__ __forceinline__ void point_of(const M3& Rw, V3 pw, V3 w, V3 v, V3 r, V3& p, V3& vel) {
    p = pw + mul(Rw, r);
    vel = mul(Rw, v + cross(w, r));
}
// the ball-collision shapes / paddle / bound centre that ride on link `li` (run-time compare against the shape table)
__device__ __forceinline__ void capture_geometry(const TAScal& C, int mask, const M3& Rw, V3 pw, V3 w, V3 v, ArmGeom<ModelG1TA::kShapes>& g, V3& bound) {
    if (mask == 0) return;
#pragma unroll
    for (int s = 0; s < ModelG1TA::kShapes; s++)
        if (mask & (1 << s)) {
            point_of(Rw, pw, w, v, ld3(C.shape_a[s]), g.a[s], g.va[s]);
            point_of(Rw, pw, w, v, ld3(C.shape_b[s]), g.b[s], g.vb[s]);
        }
    if (mask & (1 << 6)) {
        point_of(Rw, pw, w, v, ld3(C.paddle_center), g.pc, g.vpc);
        g.pn = mul(Rw, ld3(C.paddle_normal));
        g.pnd = cross(mul(Rw, w), g.pn);
    }
    if (mask & (1 << 7)) { V3 d; point_of(Rw, pw, w, v, ld3(C.bound_center), bound, d); }
}
__device__ __forceinline__ void write_row(float* row, const M3& R, V3 p, V3 lin, V3 ang) {
    float q[4];
    rot_to_quat(R, q);
    row[0] = p.x; row[1] = p.y; row[2] = p.z; row[3] = q[0]; row[4] = q[1]; row[5] = q[2]; row[6] = q[3];
    row[7] = lin.x; row[8] = lin.y; row[9] = lin.z; row[10] = ang.x; row[11] = ang.y; row[12] = ang.z;
}
// rows of link `li` and of the bodies welded to it
// The row of link L; a link that carries welded bodies also leaves its pose and twist in `pose` (20 floats per link): the welded bodies get
// their rows afterwards, three per lane (write_welded_rows).  Done inside this loop they ran one at a time — five on the torso alone —
// while the other lanes of the wave waited.
constexpr int kPoseStride = 20;
__device__ __forceinline__ void write_link_rows(float* pose, const LinkC& L, int li, const M3& Rw, V3 pw, V3 w, V3 v, float* rb) {
    const V3p la = mul(Rw, pk(v, w));        // (linear, angular) velocity in world axes
    write_row(rb + L.body * 13, Rw, pw, lo(la), hi(la));
    if (L.fcount > 0) {
        float* ps = pose + li * kPoseStride;
#pragma unroll
        for (int t = 0; t < 9; t++) ps[t] = Rw.m[t];
        ps[9] = pw.x; ps[10] = pw.y; ps[11] = pw.z; ps[12] = w.x; ps[13] = w.y; ps[14] = w.z; ps[15] = v.x; ps[16] = v.y; ps[17] = v.z;
    }
}
__device__ __forceinline__ void write_welded_rows(const FixedC* fixed, const float* pose, int role, float* rb) {
    static_assert(PPENV_TA_NUM_FIXED % 4 == 0, "welded bodies are dealt out to the four lanes of a quad");
#pragma unroll
    for (int t = 0; t < PPENV_TA_NUM_FIXED / 4; t++) {
        const FixedC F = fixed[role + 4 * t];
        const float* ps = pose + F.link * kPoseStride;
        M3 Rw;
#pragma unroll
        for (int k = 0; k < 9; k++) Rw.m[k] = ps[k];
        const V3 pw = mk(ps[9], ps[10], ps[11]), w = mk(ps[12], ps[13], ps[14]), v = mk(ps[15], ps[16], ps[17]);
        V3 p, vel;
        point_of(Rw, pw, w, v, ld3(F.xyz), p, vel);
        write_row(rb + F.body * 13, mul(Rw, ldm(F.rot)), p, vel, mul(Rw, w));
    }
}

// LDS tile -> its contiguous block of a tensor.  A full workgroup's tile (FLOATS per workgroup, a multiple of 4, 16-byte aligned on both
// sides) goes as float4 with a constant trip count, so that several LDS reads are in flight before the first store; a ragged last
// workgroup copies its `valid` floats one by one.
template <int FLOATS>
__device__ __forceinline__ void copy_tile(float* __restrict__ dst, const float* __restrict__ src, int valid, int lane) {
    static_assert(FLOATS % 4 == 0, "tile size");
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (valid == FLOATS) {
        constexpr int kVec = FLOATS / 4, kTrips = (kVec + 63) / 64;
#pragma unroll 6
        for (int it = 0; it < kTrips; it++) {
            const int k = it * 64 + lane;
            if (kVec % 64 == 0 || k < kVec) reinterpret_cast<f4v*>(dst)[k] = reinterpret_cast<const f4v*>(src)[k];
        }
        return;
    }
    for (int t = lane; t < valid; t += 64) dst[t] = src[t];
}

// what the fused launch needs beyond the rigid-body step: the task's parameters and per-env buffers (ppenv_ta_post_physics_step's)
struct TaskArgs {
    ppenv_ta_params p;
    const float* initial_rb;
    const float* reset_override;
    uint32_t* flags;
    uint32_t* episode;
    long long* progress;
    float* obs;
    float* rew;
    long long* reset;
    uint32_t* any_reset;
};

// STEP = false: forward kinematics only.  FUSE = true: post_physics_step (reward, masked reset, 313-wide observation, TA:1145-1192)
// in the same launch, on the LDS tiles, before anything goes to global memory.
template <bool STEP, bool FUSE>
__global__ __launch_bounds__(64) void ta_sim_quad_kernel(const TAConsts* __restrict__ Cp, const TAScal P, const StepConsts* __restrict__ Kp, int n, const float* __restrict__ actions,
                                                         float* root_states, float* dof_states, float* __restrict__ rb_states,
                                                         float* __restrict__ dof_force, float* __restrict__ pre_vx, const TaskArgs task) {
    __shared__ LinkC s_link[NL];
    __shared__ float s_cpoint[PPENV_TA_MAX_CONTACTS][3];
    __shared__ FixedC s_fixed[PPENV_TA_NUM_FIXED];
    // The scene constants of the ball's contact code (1.3 KB).  As a by-value kernel argument they sat in SGPRs for the whole launch and, with the
    // model scalars, overflowed them: 209 spilled SGPRs and 1.9k v_readlane in the listing.  Only the ball lane reads them, once per substep.
    __shared__ StepConsts s_K;
    __shared__ ppenv_ta_params s_tp;   // the task's parameters (100 dwords, read once at the end of the launch), likewise
    __shared__ __attribute__((aligned(16))) float s_rec[kChainLen * kRec * 64];   // later the output tiles, read back as float4
    __shared__ float s_q[NDOF][kQuadEnvs], s_qd[NDOF][kQuadEnvs], s_target[NDOF][kQuadEnvs], s_force[NDOF][kQuadEnvs];
    const TAConsts& C = *Cp;
    const int lane = threadIdx.x, role = lane & 3, es = lane >> 2;
    const int e = blockIdx.x * kQuadEnvs + es;
    const bool live = e < n;
    {   // the link table into LDS: the lanes of a quad visit different links, so its reads are per-lane from here on
        const uint32_t* src = reinterpret_cast<const uint32_t*>(C.link);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_link);
        for (int t = lane; t < (int)(sizeof(LinkC) * NL / 4); t += 64) dst[t] = src[t];
        for (int t = lane; t < PPENV_TA_MAX_CONTACTS * 3; t += 64) (&s_cpoint[0][0])[t] = (&C.cpoint[0][0])[t];
        for (int t = lane; t < (int)(sizeof(FixedC) * PPENV_TA_NUM_FIXED / 4); t += 64) reinterpret_cast<uint32_t*>(s_fixed)[t] = reinterpret_cast<const uint32_t*>(C.fixed)[t];
        if (FUSE) {
            const uint32_t* ps = reinterpret_cast<const uint32_t*>(&task.p);   // a lane-indexed read of the kernel-argument block
            uint32_t* pd = reinterpret_cast<uint32_t*>(&s_tp);
            for (int t = lane; t < (int)(sizeof(ppenv_ta_params) / 4); t += 64) pd[t] = ps[t];
        }
        if (STEP) {
            const uint32_t* ks = reinterpret_cast<const uint32_t*>(Kp);
            uint32_t* kd = reinterpret_cast<uint32_t*>(&s_K);
            for (int t = lane; t < (int)(sizeof(StepConsts) / 4); t += 64) kd[t] = ks[t];
        }
    }
    float* root = root_states + (size_t)(live ? e : 0) * 39;
    float* dofs = dof_states + (size_t)(live ? e : 0) * 2 * NDOF;
    for (int d = role; d < NDOF; d += 4) {   // the quad shares its env's dof rows
        s_q[d][es] = dofs[2 * d];
        s_qd[d][es] = dofs[2 * d + 1];
        if (STEP) {
            s_target[d][es] = pd_target(actions[(size_t)(live ? e : 0) * NDOF + d], C.link[d + 1].lo, C.link[d + 1].hi, P.clip_actions);   // VecTask.step clamp + TA:1131, 729-733
            s_force[d][es] = 0.f;
        }
    }
    TA_STAMP_AT(0);
    __syncthreads();   // one wave: orders the LDS fills above against the reads below
    BaseState base;
    load_base(root, base);
    QuadRec rc{&s_rec[lane * kRec]};
    TA_STAMP_AT(1);
    Ball ball;
    float pre_vx_reg_out = 0.f;
    if (STEP) {
        float* bl = root + 26;
        ball.p = mk(bl[0], bl[1], bl[2]);
        for (int k = 0; k < 4; k++) ball.quat[k] = bl[3 + k];
        ball.v = mk(bl[7], bl[8], bl[9]);
        ball.w = mk(bl[10], bl[11], bl[12]);
        if (live && role == 0) pre_vx[e] = ball.v.x;                                                               // TA:1143
        pre_vx_reg_out = ball.v.x;

        for (int sub = 0; sub < P.substeps; sub++) {
            // ---- pass 1: kinematics + each link's own inertia / bias / contacts, base -> tip of this lane's limb
            M3 Rw = quat_to_m3(base.quat);
            V3 pw = base.p, w = tmul(Rw, base.ww), v = tmul(Rw, base.vw);
            const M3 R0 = Rw;
            const V3 w0 = w, v0 = v;
            ArtI I0 = link_dynamics(P, s_link[0], s_cpoint, Rw, pw, w, v);     // the pelvis: every lane of the quad computes it
            ArmGeom<ModelG1TA::kShapes> g[1];
            V3 bound[1];
            capture_geometry(P, s_link[0].geo_mask, Rw, pw, w, v, g[0], bound[0]);
            for (int k = 0; k < kChainLen; k++) {
                const int li = chain_link(role, k);
                if (li < 0) continue;
                const LinkC L = s_link[li];   // by value: plain LDS reads into registers
                M3 E;
                link_kinematics(L, s_q[li - 1][es], s_qd[li - 1][es], Rw, pw, w, v, E);
                rec_store4(rc, k, 0, E.m[0], E.m[1], E.m[2], E.m[3]);
                rec_store4(rc, k, 4, E.m[4], E.m[5], E.m[6], E.m[7]);
                rec_store4(rc, k, 8, E.m[8], w.x, w.y, w.z);
                rec_store4(rc, k, 12, v.x, v.y, v.z, 0.f);
                rec_store_art(rc, k, link_dynamics(P, L, s_cpoint, Rw, pw, w, v));
                capture_geometry(P, L.geo_mask, Rw, pw, w, v, g[0], bound[0]);   // complete on the right-arm lane (pelvis, torso, right arm)
            }
            TA_STAMP_AT(2 + 5 * (sub & 1));
            // ---- pass 2: articulated inertias tip -> base; the arms meet at the torso, everything at the pelvis
            ArtI acc = art_zero();
            for (int k = kChainLen - 1; k >= 0; k--) {
                if (k == kHubPos) {
                    ArtI other = art_shfl_xor(acc, 1);                   // lanes 2 <-> 3 (and, unused, 0 <-> 1)
                    if (role >= 2) add_art(acc, other);
                }
                const int li = chain_link(role, k);
                if (li < 0) continue;
                const LinkC L = s_link[li];   // by value: plain LDS reads into registers
                ArtI I = rec_load_art(rc, k);
                add_art(I, acc);
                JointOut jo;
                M3 E;
#pragma unroll
                for (int t = 0; t < 9; t++) E.m[t] = rc(k, R_E + t);
                inward_step(P, L, I, mk(rc(k, R_W), rc(k, R_W + 1), rc(k, R_W + 2)), mk(rc(k, R_V), rc(k, R_V + 1), rc(k, R_V + 2)), E,
                            s_q[li - 1][es], s_qd[li - 1][es], s_target[li - 1][es], jo);
                rec_store4(rc, k, R_JO, jo.ua.x, jo.ua.y, jo.ua.z, jo.ub.x);
                rec_store4(rc, k, R_JO + 4, jo.ub.y, jo.ub.z, jo.dinv, jo.u);
                acc = I;
            }
            TA_STAMP_AT(3 + 5 * (sub & 1));
            if (role == 3) acc = art_zero();                             // its waist chain duplicates lane 2's
            {
                ArtI t = art_shfl_xor(acc, 1);
                add_art(acc, t);
                t = art_shfl_xor(acc, 2);
                add_art(acc, t);
            }
            add_art(I0, acc);
            V3 alpha, a;
            solve_base_art(I0, alpha, a);
            TA_STAMP_AT(4 + 5 * (sub & 1));
            // ---- pass 3: accelerations base -> tip, joints integrated on the way
            V3 aw = alpha, av = a;
            for (int k = 0; k < kChainLen; k++) {
                const int li = chain_link(role, k);
                if (li < 0) continue;
                const LinkC L = s_link[li];   // by value: plain LDS reads into registers
                JointOut jo = {mk(rc(k, R_JO), rc(k, R_JO + 1), rc(k, R_JO + 2)), mk(rc(k, R_JO + 3), rc(k, R_JO + 4), rc(k, R_JO + 5)), rc(k, R_JO + 6),
                               rc(k, R_JO + 7)};
                float q = s_q[li - 1][es], qd = s_qd[li - 1][es], force;
                M3 E;
#pragma unroll
                for (int t = 0; t < 9; t++) E.m[t] = rc(k, R_E + t);
                outward_step(P, L, E, mk(rc(k, R_W), rc(k, R_W + 1), rc(k, R_W + 2)), mk(rc(k, R_V), rc(k, R_V + 1), rc(k, R_V + 2)), jo,
                             aw, av, s_target[li - 1][es], q, qd, force);
                if (!(role == 3 && k < 3)) {                             // the waist dofs are written by lane 2
                    s_q[li - 1][es] = q; s_qd[li - 1][es] = qd; s_force[li - 1][es] = force;
                }
            }
            integrate_base_regs(P, R0, w0, v0, base, alpha, a);
            TA_STAMP_AT(5 + 5 * (sub & 1));
            if (role == 3) ball_substep<ModelG1TA, 1>(s_K, ball, g, bound);
            TA_STAMP_AT(6 + 5 * (sub & 1));
            __builtin_amdgcn_wave_barrier();                             // lane 2's waist dofs before every lane's next pass 1
        }
    }
    // ---- outputs.  Everything a workgroup writes is a contiguous block of its tensor (16 envs): it is assembled in LDS tiles
    // (the link records are dead by now) and leaves coalesced.  Tiles: rigid_body_states [16][42 x 13], root_states [16][39],
    // dof_states [16][54], dof_force [16][27] and, fused, the observation rows [16][313].
    __builtin_amdgcn_wave_barrier();
    constexpr int kRb = PPENV_NUM_BODIES * 13, kRoot = PPENV_NUM_ACTORS * 13, kDofs = 2 * NDOF;
    float* t_rb = s_rec;
    float* t_root = t_rb + kQuadEnvs * kRb;
    float* t_dofs = t_root + kQuadEnvs * kRoot;
    float* t_frc = t_dofs + kQuadEnvs * kDofs;
    float* t_obs = t_frc + kQuadEnvs * NDOF;
    float* t_pvx = t_obs + kQuadEnvs * PPENV_TA_NUM_OBS;
    float* t_pose = t_pvx + kQuadEnvs;   // [16][NL][20]: pose and twist of the links that carry welded bodies
    static_assert(kQuadEnvs * (kRb + kRoot + kDofs + NDOF + PPENV_TA_NUM_OBS + 1 + NL * kPoseStride) <= kChainLen * kRec * 64, "output tiles exceed the record area");
    {
        float* rb = t_rb + es * kRb;
        M3 Rw = quat_to_m3(base.quat);
        V3 pw = base.p, w = tmul(Rw, base.ww), v = tmul(Rw, base.vw);
        float table_row[13];               // untouched by the step; fetched here so that the chain loop below hides the latency
        if (role == 1) {
#pragma unroll
            for (int k = 0; k < 13; k++) table_row[k] = root[13 + k];
        }
        float* pose = t_pose + es * (NL * kPoseStride);
        if (role == 0) write_link_rows(pose, s_link[0], 0, Rw, pw, w, v, rb);
        for (int k = 0; k < kChainLen; k++) {
            const int li = chain_link(role, k);
            if (li < 0) continue;
            const LinkC L = s_link[li];
            M3 E;
            link_kinematics(L, s_q[li - 1][es], s_qd[li - 1][es], Rw, pw, w, v, E);
            if (!(role == 3 && k < 3)) write_link_rows(pose, L, li, Rw, pw, w, v, rb);
        }
        __builtin_amdgcn_wave_barrier();   // the poses above (written by other lanes of the quad) before the welded rows
        write_welded_rows(s_fixed, pose, role, rb);
        TA_STAMP_AT(16);
        float* tr = t_root + es * kRoot;
        if (role == 0) {
            const float br[13] = {base.p.x, base.p.y, base.p.z, base.quat[0], base.quat[1], base.quat[2], base.quat[3],
                                  base.vw.x, base.vw.y, base.vw.z, base.ww.x, base.ww.y, base.ww.z};
            for (int k = 0; k < 13; k++) tr[k] = br[k];
        }
        if (role == 1) {
#pragma unroll
            for (int k = 0; k < 13; k++) { tr[13 + k] = table_row[k]; rb[40 * 13 + k] = table_row[k]; }
        }
        if (STEP && role == 3) t_pvx[es] = pre_vx_reg_out;
        if (role == 3) {                   // ball row: this lane holds the stepped ball
            float bl[13];
            if (STEP) {
                const float t[13] = {ball.p.x, ball.p.y, ball.p.z, ball.quat[0], ball.quat[1], ball.quat[2], ball.quat[3],
                                     ball.v.x, ball.v.y, ball.v.z, ball.w.x, ball.w.y, ball.w.z};
                for (int k = 0; k < 13; k++) bl[k] = t[k];
            } else {
                for (int k = 0; k < 13; k++) bl[k] = root[26 + k];
            }
            for (int k = 0; k < 13; k++) { tr[26 + k] = bl[k]; rb[41 * 13 + k] = bl[k]; }
        }
        if (STEP) {
#pragma unroll
            for (int t = 0; t < (NDOF + 3) / 4; t++) {     // constant trip count: the LDS reads of all seven rounds are in flight together
                const int d = role + 4 * t;
                if (d < NDOF) {
                    t_dofs[es * kDofs + 2 * d] = s_q[d][es];
                    t_dofs[es * kDofs + 2 * d + 1] = s_qd[d][es];
                    t_frc[es * NDOF + d] = s_force[d][es];
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    TA_STAMP_AT(12);
    const int nv = min(kQuadEnvs, n - blockIdx.x * kQuadEnvs);
    const size_t e0 = (size_t)blockIdx.x * kQuadEnvs;
    // rigid_body_states are the PRE-reset ones (the reference refreshes them before reward / reset, TA:1150-1160)
    copy_tile<kQuadEnvs * kRb>(rb_states + e0 * kRb, t_rb, nv * kRb, lane);
    TA_STAMP_AT(13);
    if (STEP) {
        copy_tile<kQuadEnvs * NDOF>(dof_force + e0 * NDOF, t_frc, nv * NDOF, lane);
        if (FUSE) {
            {   // the four lanes of the quad share the env's task arithmetic (lanes past the last env compute on env n - 1, store nothing)
                const int ec = live ? e : n - 1;
                tatask::ta_task_env<4>(s_tp, ec, t_rb + es * kRb, task.initial_rb + (size_t)(s_tp.initial_rb_shared ? 0 : ec) * kRb, t_root + es * kRoot, t_dofs + es * kDofs,
                                       t_frc + es * NDOF, t_pvx[es], task.reset_override ? task.reset_override + (size_t)ec * 5 : nullptr, &task.flags[ec],
                                       &task.episode[ec], &task.progress[ec], t_obs + es * PPENV_TA_NUM_OBS, &task.rew[ec], &task.reset[ec], task.any_reset,
                                       role, live);
            }
            __builtin_amdgcn_wave_barrier();
            TA_STAMP_AT(14);
            copy_tile<kQuadEnvs * PPENV_TA_NUM_OBS>(task.obs + e0 * PPENV_TA_NUM_OBS, t_obs, nv * PPENV_TA_NUM_OBS, lane);
        }
        copy_tile<kQuadEnvs * kRoot>(root_states + e0 * kRoot, t_root, nv * kRoot, lane);
        copy_tile<kQuadEnvs * kDofs>(dof_states + e0 * kDofs, t_dofs, nv * kDofs, lane);
    }
    TA_STAMP_AT(15);
}

// pre_physics_step alone (ppenv_ta_pd_targets) and TA's serve on explicit draws (ppenv_ta_serve_from_draws): the step's own device functions
__global__ void ta_pd_targets_kernel(const TAConsts* __restrict__ Cp, int n, const float* __restrict__ actions, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * NDOF) return;
    const LinkC& L = Cp->link[t % NDOF + 1];
    out[t] = pd_target(actions[t], L.lo, L.hi, Cp->sc.clip_actions);
}
__global__ void ta_serve_from_draws_kernel(int m, const float* __restrict__ draws, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const V3 v = serve_from_draws(PPENV_VARIANT_TN, draws[3 * t], draws[3 * t + 1], draws[3 * t + 2]);   // TA:370-375 is TN's form
    out[3 * t] = v.x; out[3 * t + 1] = v.y; out[3 * t + 2] = v.z;
}

// does the model have the tree the quad kernel's chains are written for?
bool quad_topology(const TAConsts& C) {
    static const int parents[NL] = {-1, 0, 1, 2, 3, 4, 5, 0, 7, 8, 9, 10, 11, 0, 13, 14, 15, 16, 17, 18, 19, 20, 21, 15, 23, 24, 25, 26};
    for (int i = 0; i < NL; i++)
        if (C.link[i].parent != parents[i]) return false;
    return true;
}
}  // namespace

struct ppenv_ta_sim {
    TAConsts host;
    TAConsts* dev;
    StepConsts K;
    StepConsts* devK;      // the same in device memory (the quad kernel stages it in LDS)
    int device;
    int quad;     // 1: ta_sim_quad_kernel (four lanes per env), 0: ta_sim_kernel (one lane per env; any tree)
    int chain;    // 1: ppenv_ta_step runs the chain-wave kernel (ppenv_ta_chain.hip: one lane per env, one wave per limb; the compiled G1 model only)
    uint32_t* status_host;   // PPENV_STATUS_* bits, pinned host memory the kernels write through (cf. ppenv::status_host)
    uint32_t* status_dev;
    const float* pin_mean;   // ppenv_ta_sim_set_policy_input (chain-wave kernel only); pin_out NULL: off
    const float* pin_inv_std;
    float pin_clip;
    unsigned short* pin_out;
    int pin_ld;
    ppenv_ta_randomization dr;   // ppenv_ta_sim_set_randomization (chain-wave kernel only); all NULL / 0: off
};

namespace {
// every entry point launches on the device the handle was created on, whatever the caller's current device is
int ta_use_device(const ppenv_ta_sim* s) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || (cur != s->device && hipSetDevice(s->device) != hipSuccess)) {
        ppenv_set_error("selecting the 27-dof handle's device failed");
        return PPENV_EHIP;
    }
    return PPENV_OK;
}
}  // namespace

extern "C" {

int ppenv_ta_sim_device(const ppenv_ta_sim* s) { return s ? s->device : -1; }
uint32_t ppenv_ta_sim_status(const ppenv_ta_sim* s) { return s ? *(volatile uint32_t*)s->status_host : 0u; }
/* the policy's first-layer input written by ppenv_ta_step itself (chain-wave kernel): out NULL switches it off */
int ppenv_ta_sim_set_policy_input(ppenv_ta_sim* s, const float* mean_dev, const float* inv_std_dev, float clip, void* out_f16_dev, int32_t ld_out) {
    if (!s) { ppenv_set_error("ppenv_ta_sim_set_policy_input: NULL handle"); return PPENV_EINVAL; }
    if (!out_f16_dev) { s->pin_out = nullptr; return PPENV_OK; }
    if (!s->chain) { ppenv_set_error("ppenv_ta_sim_set_policy_input: only the chain-wave kernel writes the policy input (ppenv_ta_sim_kernel() == 2)"); return PPENV_EINVAL; }
    if (!mean_dev || !inv_std_dev || ld_out < PPENV_TA_NUM_OBS || (ld_out & 1) || (reinterpret_cast<uintptr_t>(out_f16_dev) & 3)) {
        ppenv_set_error("ppenv_ta_sim_set_policy_input: need mean, inv_std, an even ld_out >= 313 and a 4-byte aligned output");
        return PPENV_EINVAL;
    }
    s->pin_mean = mean_dev; s->pin_inv_std = inv_std_dev; s->pin_clip = clip; s->pin_out = reinterpret_cast<unsigned short*>(out_f16_dev); s->pin_ld = ld_out;
    return PPENV_OK;
}
/* domain randomisation tables of the 27-DoF step (chain-wave kernel): dr NULL switches it off */
int ppenv_ta_sim_set_randomization(ppenv_ta_sim* s, const ppenv_ta_randomization* dr) {
    if (!s) { ppenv_set_error("ppenv_ta_sim_set_randomization: NULL handle"); return PPENV_EINVAL; }
    if (!dr) { s->dr = ppenv_ta_randomization{}; return PPENV_OK; }
    if (!s->chain) { ppenv_set_error("ppenv_ta_sim_set_randomization: only the chain-wave kernel reads the tables (ppenv_ta_sim_kernel() == 2: the compiled G1 model)"); return PPENV_EINVAL; }
    if (!(dr->action_noise_sigma >= 0.f) || !(dr->observation_noise_sigma >= 0.f)) { ppenv_set_error("ppenv_ta_sim_set_randomization: noise amplitudes must be >= 0"); return PPENV_EINVAL; }
    s->dr = *dr;
    return PPENV_OK;
}
/* which kernel ppenv_ta_step launches: 2 chain-wave, 1 quad, 0 one lane per env */
int ppenv_ta_sim_kernel(const ppenv_ta_sim* s) { return s ? (s->chain ? 2 : (s->quad ? 1 : 0)) : -1; }
/* host-only (no GPU call): 1 when the model equals, bit for bit, the tables compiled into the chain-wave kernel; 0 when it differs; < 0 on a bad model */
int ppenv_ta_model_is_compiled(const ppenv_config* scene, const ppenv_ta_model* model) {
    static TAConsts C;
    const char* why = "";
    if (!scene || !model || !make_ta_consts(*scene, *model, C, &why)) { ppenv_set_error(why); return PPENV_EINVAL; }
    char msg[160] = "";
    const bool same = ta_chain_model_matches(C, msg, sizeof msg);
    if (!same) ppenv_set_error(msg);
    return same ? 1 : 0;
}

#if defined(TA_STAMP)
int ppenv_ta_debug_read_stamps(unsigned long long* dst, size_t count) {
    if (hipDeviceSynchronize() != hipSuccess) return PPENV_EHIP;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ta_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : PPENV_EHIP;
}
#endif

int ppenv_ta_sim_create(const ppenv_config* scene, const ppenv_ta_model* model, void* stream, ppenv_ta_sim** out) {
    if (!scene || !model || !out) { ppenv_set_error("ppenv_ta_sim_create: NULL argument"); return PPENV_EINVAL; }
    *out = nullptr;
    if (scene->abi_version != PPENV_ABI_VERSION) { ppenv_set_error("scene.abi_version does not match this library"); return PPENV_EINVAL; }
    ppenv_ta_sim* s = new (std::nothrow) ppenv_ta_sim;
    if (!s) { ppenv_set_error("out of host memory"); return PPENV_ENOMEM; }
    const char* why = "";
    if (!make_ta_consts(*scene, *model, s->host, &why)) {
        char msg[256];
        snprintf(msg, sizeof msg, "ppenv_ta_sim_create: %s", why);
        ppenv_set_error(msg);
        delete s;
        return PPENV_EINVAL;
    }
    s->K = make_step_consts(*scene);
    s->dev = nullptr;
    s->devK = nullptr;
    {   // PPENV_TA_KERNEL=lane|quad|chain forces a mapping (same arithmetic; quad needs the G1 tree, chain the compiled G1 model)
        const char* k = getenv("PPENV_TA_KERNEL");
        s->quad = quad_topology(s->host) && !(k && strcmp(k, "lane") == 0);
        s->chain = ta_chain_model_matches(s->host) && !(k && (strcmp(k, "lane") == 0 || strcmp(k, "quad") == 0));
        if (k && strcmp(k, "chain") == 0 && !s->chain) {
            ppenv_set_error("PPENV_TA_KERNEL=chain, but the model differs from the one compiled into the chain-wave kernel (run python -m isaacgym_amd.modelgen_ta and rebuild)");
            delete s;
            return PPENV_EINVAL;
        }
    }
    s->status_host = s->status_dev = nullptr;
    s->pin_mean = s->pin_inv_std = nullptr; s->pin_out = nullptr; s->pin_clip = 0.f; s->pin_ld = 0;
    s->dr = ppenv_ta_randomization{};
    // the handle lives on scene->device_id when that names a visible GPU (the caller's current device otherwise)
    int ndev = 0;
    s->device = -1;
    if (hipGetDeviceCount(&ndev) == hipSuccess && scene->device_id >= 0 && scene->device_id < ndev && hipSetDevice(scene->device_id) == hipSuccess) s->device = scene->device_id;
    if ((s->device < 0 && hipGetDevice(&s->device) != hipSuccess) || hipMalloc((void**)&s->dev, sizeof(TAConsts)) != hipSuccess ||
        hipMalloc((void**)&s->devK, sizeof(StepConsts)) != hipSuccess) {
        ppenv_set_error("ppenv_ta_sim_create: hipMalloc of the model constants failed");
        if (s->dev) (void)hipFree(s->dev);
        delete s;
        return PPENV_EHIP;
    }
    if (hipHostMalloc((void**)&s->status_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&s->status_dev, s->status_host, 0) != hipSuccess) {
        ppenv_set_error("ppenv_ta_sim_create: allocating the device status word failed");
        if (s->status_host) (void)hipHostFree(s->status_host);
        (void)hipFree(s->dev); (void)hipFree(s->devK);
        delete s;
        return PPENV_ENOMEM;
    }
    *s->status_host = 0u;
    // pageable host source: the copy is staged before the call returns, the struct may be reused by the caller
    if (hipMemcpyAsync(s->dev, &s->host, sizeof(TAConsts), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipMemcpyAsync(s->devK, &s->K, sizeof(StepConsts), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        (void)hipFree(s->dev);
        (void)hipFree(s->devK);
        (void)hipHostFree(s->status_host);
        delete s;
        ppenv_set_error("ppenv_ta_sim_create: uploading the model constants failed");
        return PPENV_EHIP;
    }
    *out = s;
    return PPENV_OK;
}

void ppenv_ta_sim_destroy(ppenv_ta_sim* s) {
    if (!s) return;
    (void)ta_use_device(s);
    if (s->dev) (void)hipFree(s->dev);
    if (s->devK) (void)hipFree(s->devK);
    if (s->status_host) (void)hipHostFree(s->status_host);
    delete s;
}

// sim_params.gravity of this simulation (cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml:123-124 randomises it): the humanoid's links read it from the
// model constants (by value in the chain-wave kernel's argument, in device memory for the table-driven kernels), the ball from StepConsts — both
// copies are refreshed in stream order, so every launch enqueued after this call on `stream` steps under the new value.
int ppenv_ta_sim_set_gravity(ppenv_ta_sim* s, float gravity_z, void* stream) {
    if (!s) { ppenv_set_error("ppenv_ta_sim_set_gravity: NULL handle"); return PPENV_EINVAL; }
    if (!(gravity_z <= 0.f)) { ppenv_set_error("gravity_z must be <= 0 (the world's up axis is z)"); return PPENV_EINVAL; }
    if (int rc = ta_use_device(s)) return rc;
    s->host.sc.gravity_z = gravity_z;
    s->K.gdv = gravity_z * s->K.hb;
    // pageable host sources: staged before the calls return
    if (hipMemcpyAsync(s->dev, &s->host, sizeof(TAConsts), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipMemcpyAsync(s->devK, &s->K, sizeof(StepConsts), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        ppenv_set_error("ppenv_ta_sim_set_gravity: uploading the constants failed");
        return PPENV_EHIP;
    }
    return PPENV_OK;
}

// The kernel ppenv_ta_step launches for this handle now (demangled, as rocprofv3's kernel trace shows it).
const char* ppenv_ta_sim_kernel_name(const ppenv_ta_sim* s) {
    if (!s) return "";
    const bool dr = s->dr.dof_stiffness_scale || s->dr.dof_damping_scale || s->dr.link_mass_scale || s->dr.restitution_scale || s->dr.friction_scale ||
                    s->dr.action_noise_sigma > 0.f || s->dr.observation_noise_sigma > 0.f;
    if (s->chain) return dr ? "ta_chain_kernel<true>" : "ta_chain_kernel<false>";
    return s->quad ? "ta_sim_quad_kernel<true, true>" : "ta_sim_kernel<true> + ta_post_physics_kernel";
}

int ppenv_ta_simulate(ppenv_ta_sim* s, int32_t n, const float* actions_dev, float* root_states_dev, float* dof_states_dev, float* rb_states_dev,
                      float* dof_force_dev, float* pre_ball_vx_dev, void* stream) {
    if (!s || n <= 0 || !actions_dev || !root_states_dev || !dof_states_dev || !rb_states_dev || !dof_force_dev || !pre_ball_vx_dev) {
        ppenv_set_error("ppenv_ta_simulate: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    if (s->dr.dof_stiffness_scale || s->dr.dof_damping_scale || s->dr.link_mass_scale || s->dr.restitution_scale || s->dr.friction_scale ||
        s->dr.action_noise_sigma > 0.f || s->dr.observation_noise_sigma > 0.f) {
        ppenv_set_error("ppenv_ta_simulate: a randomisation is set; its tables are read by ppenv_ta_step (the chain-wave kernel) only");
        return PPENV_EINVAL;
    }
    if (int rc = ta_use_device(s)) return rc;
    if (s->quad)
        hipLaunchKernelGGL((ta_sim_quad_kernel<true, false>), dim3((n + kQuadEnvs - 1) / kQuadEnvs), dim3(64), 0, (hipStream_t)stream, s->dev, s->host.sc, s->devK, n, actions_dev,
                           root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev, TaskArgs{});
    else
        hipLaunchKernelGGL(ta_sim_kernel<true>, dim3((n + kTaLanes - 1) / kTaLanes), dim3(kTaLanes), 0, (hipStream_t)stream, s->dev, s->K, n, actions_dev,
                           root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_sim_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

int ppenv_ta_step(ppenv_ta_sim* s, const ppenv_ta_params* params, const float* actions_dev, const float* initial_rb_states_dev, float* root_states_dev,
                  float* dof_states_dev, float* rb_states_dev, float* dof_force_dev, float* pre_ball_vx_dev, const float* reset_override_dev,
                  uint32_t* flags_dev, uint32_t* episode_dev, int64_t* progress_dev, float* obs_dev, float* rew_dev, int64_t* reset_dev,
                  uint32_t* scratch_any_reset_dev, void* stream) {
    if (!s || !params || params->num_envs <= 0 || !actions_dev || !initial_rb_states_dev || !root_states_dev || !dof_states_dev ||
        !dof_force_dev || !pre_ball_vx_dev || !flags_dev || !episode_dev || !progress_dev || !obs_dev || !rew_dev || !reset_dev || !scratch_any_reset_dev) {
        ppenv_set_error("ppenv_ta_step: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    if (*(volatile uint32_t*)s->status_host != 0u) {
        ppenv_set_error("ppenv_ta_step: an earlier launch reported a hand-off time-out (status word set): the state tensors of that launch were not stored; destroy the handle");
        return PPENV_EDEVICE;
    }
    const int n = params->num_envs;
    if (int rc = ta_use_device(s)) return rc;
    if (s->chain) {   // one lane per env, one wave per limb; rigid_body_states only on request
        TAChainArgs a{*params, s->devK, actions_dev, initial_rb_states_dev, root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev,
                      reset_override_dev, flags_dev, episode_dev, (long long*)progress_dev, obs_dev, rew_dev, (long long*)reset_dev, scratch_any_reset_dev, s->status_dev,
                      s->pin_mean, s->pin_inv_std, s->pin_clip, s->pin_out, s->pin_ld,
                      s->dr.dof_stiffness_scale, s->dr.dof_damping_scale, s->dr.link_mass_scale, s->dr.restitution_scale, s->dr.friction_scale,
                      s->dr.action_noise_sigma, s->dr.observation_noise_sigma};
        return ta_chain_launch(s->host.sc, a, stream);
    }
    if (!rb_states_dev) { ppenv_set_error("ppenv_ta_step: rb_states may only be NULL with the chain-wave kernel (the compiled G1 model)"); return PPENV_EINVAL; }
    if (!s->quad) {   // another tree, or PPENV_TA_KERNEL=lane: the two launches
        int rc = ppenv_ta_simulate(s, n, actions_dev, root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev, stream);
        if (rc) return rc;
        return ppenv_ta_post_physics_step(params, rb_states_dev, initial_rb_states_dev, root_states_dev, dof_states_dev, dof_force_dev, pre_ball_vx_dev,
                                          reset_override_dev, flags_dev, episode_dev, progress_dev, obs_dev, rew_dev, reset_dev, scratch_any_reset_dev, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    TaskArgs t{*params, initial_rb_states_dev, reset_override_dev, flags_dev, episode_dev, (long long*)progress_dev, obs_dev, rew_dev, (long long*)reset_dev,
               scratch_any_reset_dev};
    hipLaunchKernelGGL((ta_sim_quad_kernel<true, true>), dim3((n + kQuadEnvs - 1) / kQuadEnvs), dim3(64), 0, st, s->dev, s->host.sc, s->devK, n, actions_dev,
                       root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev, t);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching the fused 27-dof step failed"); return PPENV_EHIP; }
    return ppenv_ta_clear_counts(n, flags_dev, scratch_any_reset_dev, stream);
}

int ppenv_ta_pd_targets(ppenv_ta_sim* s, int32_t n, const float* actions_dev, float* pd_tar_dev, void* stream) {
    if (!s || n <= 0 || !actions_dev || !pd_tar_dev) { ppenv_set_error("ppenv_ta_pd_targets: NULL argument or num_envs <= 0"); return PPENV_EINVAL; }
    if (int rc = ta_use_device(s)) return rc;
    hipLaunchKernelGGL(ta_pd_targets_kernel, dim3((n * NDOF + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->dev, n, actions_dev, pd_tar_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_pd_targets_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

int ppenv_ta_serve_from_draws(ppenv_ta_sim* s, const float* draws_dev, int32_t m, float* vel_dev, void* stream) {
    if (!s || m <= 0 || !draws_dev || !vel_dev) { ppenv_set_error("ppenv_ta_serve_from_draws: NULL argument or m <= 0"); return PPENV_EINVAL; }
    if (int rc = ta_use_device(s)) return rc;
    hipLaunchKernelGGL(ta_serve_from_draws_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, draws_dev, vel_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_serve_from_draws_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

int ppenv_ta_forward_kinematics(ppenv_ta_sim* s, int32_t n, const float* root_states_dev, const float* dof_states_dev, float* rb_states_dev,
                                void* stream) {
    if (!s || n <= 0 || !root_states_dev || !dof_states_dev || !rb_states_dev) {
        ppenv_set_error("ppenv_ta_forward_kinematics: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    if (int rc = ta_use_device(s)) return rc;
    if (s->quad)
        hipLaunchKernelGGL((ta_sim_quad_kernel<false, false>), dim3((n + kQuadEnvs - 1) / kQuadEnvs), dim3(64), 0, (hipStream_t)stream, s->dev, s->host.sc, s->devK, n,
                           (const float*)nullptr, const_cast<float*>(root_states_dev), const_cast<float*>(dof_states_dev), rb_states_dev,
                           (float*)nullptr, (float*)nullptr, TaskArgs{});
    else
        hipLaunchKernelGGL(ta_sim_kernel<false>, dim3((n + kTaLanes - 1) / kTaLanes), dim3(kTaLanes), 0, (hipStream_t)stream, s->dev, s->K, n,
                           (const float*)nullptr, const_cast<float*>(root_states_dev), const_cast<float*>(dof_states_dev), rb_states_dev,
                           (float*)nullptr, (float*)nullptr);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_sim_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

}  // extern "C"
