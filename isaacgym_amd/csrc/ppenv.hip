// ppenv.hip — HIP kernels (gfx950) and the C ABI of include/ppenv.h.
//
// Data layout in HBM: simulation state is SoA, [field][num_envs] fp32 (7 dof_pos rows,
// 7 dof_vel, 7 dof_force, 13 ball rows) + u32 flags/episode + i64 progress/reset, so lane
// e of a wave reads element e of every row: each state access of a 64-lane wave is one
// contiguous 256-byte (or 512-byte for i64) segment.  The two row-major tensors the
// VecTask surface fixes — actions [N,7] in, obs_buf [N,80] out — are transposed through
// LDS so that they, too, move as contiguous 16-byte-per-lane segments.
//
// Mapping: one lane per env, 64-thread workgroups (one wave), so N=16384 gives 256
// workgroups = one per CU.  There is no inter-env communication and no MFMA: the step is
// ~10^4 dependent fp32 VALU operations per env on ~0.6 KB of state.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using std::min;
using std::max;

#ifndef PP_ABLATE
#define PP_ABLATE 0
#endif
#if defined(PP_STAMP)
__device__ unsigned long long pp_stamp_buf[4096 * 32];
#endif
#include "ppenv_device.h"

using namespace pp;

namespace {

constexpr int kBlock = 64;   // exactly one wave per workgroup: step_kernel relies on it (no barrier around its LDS tile)
constexpr int kObsStride = PPENV_NUM_OBS + 1;   // +1 float: lanes write LDS rows bank-conflict-free
constexpr int kMaxSplitSubsteps = 4;
constexpr int kDefaultBallWaves = 1;            // ball waves per 64 envs (PPENV_BALL_WAVES overrides; see step_kernel_split's BW)            // the multi-wave step kernels keep one LDS hand-off slot per substep boundary

struct DevBuffers {
    float* obs;
    float* rew;
    long long* reset;
    long long* progress;
    float* dof_pos;
    float* dof_vel;
    float* dof_force;
    float* ball;
    uint32_t* flags;
    uint32_t* episode;
    float* serve;
};

// State stores of the step kernels.  PP_STORE_MODE (profiling builds) selects how they leave the CU:
// 0 plain, 1 non-temporal (default), 2 system-scope (write-through).  Measured at N = 16384 / 65536 (tools/gpu_storemode.sh):
// 13.03 / 19.50 us, 12.92 / 19.18 us, 13.13 / 20.28 us — nothing in the launch reads the state again, so it may stream out.
#ifndef PP_STORE_MODE
#define PP_STORE_MODE 1
#endif
template <class V>
__device__ __forceinline__ void st_state(V* p, V v) {
#if PP_STORE_MODE == 1
    __builtin_nontemporal_store(v, p);
#elif PP_STORE_MODE == 2
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *p = v;
#endif
}

// ------------------------------------------------------------- SoA state <-> registers
template <int A>
__device__ __forceinline__ void load_state(const DevBuffers& b, int n, int i, EnvStateT<A>& st) {
#pragma unroll
    for (int d = 0; d < A * ND; d++) {
        st.q[d] = b.dof_pos[(size_t)d * n + i];
        st.qd[d] = b.dof_vel[(size_t)d * n + i];
    }
    float bl[13];
#pragma unroll
    for (int k = 0; k < 13; k++) bl[k] = b.ball[(size_t)k * n + i];
    st.ball.p = mk(bl[0], bl[1], bl[2]);
#pragma unroll
    for (int k = 0; k < 4; k++) st.ball.quat[k] = bl[3 + k];
    st.ball.v = mk(bl[7], bl[8], bl[9]);
    st.ball.w = mk(bl[10], bl[11], bl[12]);
    st.progress = b.progress[(size_t)i * A];   // rows A*e .. A*e + A-1 carry the same value
#pragma unroll
    for (int a = 0; a < A; a++) st.flags[a] = b.flags[(size_t)a * n + i];
    st.episode = b.episode[i];
}
__device__ __forceinline__ void store_ball(const DevBuffers& b, int n, int i, const Ball& ball) {
    const float bl[13] = {ball.p.x, ball.p.y, ball.p.z, ball.quat[0], ball.quat[1], ball.quat[2], ball.quat[3],
                          ball.v.x, ball.v.y, ball.v.z, ball.w.x, ball.w.y, ball.w.z};
#pragma unroll
    for (int k = 0; k < 13; k++) st_state(&b.ball[(size_t)k * n + i], bl[k]);
}
// progress / flags / episode and the per-agent rows of rew / reset
template <int A>
__device__ __forceinline__ void store_task(const DevBuffers& b, int n, int i, const EnvStateT<A>& st, const float* rew, long long reset) {
#pragma unroll
    for (int a = 0; a < A; a++) {
        st_state(&b.progress[(size_t)i * A + a], st.progress);
        st_state(&b.flags[(size_t)a * n + i], st.flags[a]);
        st_state(&b.rew[(size_t)i * A + a], rew[a]);
        st_state(&b.reset[(size_t)i * A + a], reset);
    }
    st_state(&b.episode[i], st.episode);
}
template <int A>
__device__ __forceinline__ void store_state(const DevBuffers& b, int n, int i, const EnvStateT<A>& st, const float* rew, long long reset) {
#pragma unroll
    for (int d = 0; d < A * ND; d++) {
        b.dof_pos[(size_t)d * n + i] = st.q[d];
        b.dof_vel[(size_t)d * n + i] = st.qd[d];
        b.dof_force[(size_t)d * n + i] = st.dof_force[d];
    }
    store_ball(b, n, i, st.ball);
    store_task<A>(b, n, i, st, rew, reset);
}

struct LdsRowStore {
    float* row;
    __device__ __forceinline__ void operator()(int k, float v) { row[k] = v; }
};
// the same with the observation noise of the domain randomisation added (yaml:106-109); index 16 + k keeps clear of the action draws
struct NoisyRowStore {
    float* row;
    float sigma;
    uint64_t seed;
    uint32_t gid, episode, progress;
    uint32_t base = 16u;      // first noise index of this row: 16 for the one agent of the 3-actor variants, 16 + 80 a for agent a of the 4-actor one
    __device__ __forceinline__ void operator()(int k, float v) { row[k] = v + sigma * dr_gauss(seed, gid, episode, progress, base + (uint32_t)k); }
};
// device pointers of the randomisation tables (ppenv_randomization), by value in the kernel argument
struct DRTables {
    const float* kp; const float* kd; const float* ms; const float* es; const float* fs;
    float act_sigma, obs_sigma;
};

// obs rows of one workgroup: LDS [kBlock][kObsStride] -> obs_buf[base*80 ...], 16 B per lane, contiguous
__device__ __forceinline__ void flush_obs(const float* s_obs, float* obs, int base, int nvalid, int lane) {
    const int total = nvalid * PPENV_NUM_OBS;   // multiple of 4; a float4 never straddles rows (80 % 4 == 0)
    float4* dst = reinterpret_cast<float4*>(obs + (size_t)base * PPENV_NUM_OBS);
    for (int k = lane * 4; k < total; k += kBlock * 4) {
        int r = k / PPENV_NUM_OBS, c = k - r * PPENV_NUM_OBS;
        const float* src = &s_obs[r * kObsStride + c];
        // non-temporal: the env never reads obs_buf back, so the rows need not displace its state in L2
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v val = {src[0], src[1], src[2], src[3]};
        __builtin_nontemporal_store(val, reinterpret_cast<f4v*>(&dst[k >> 2]));
    }
}

// ------------------------------------------------------------------ the fused step
// K1..K8 of SURVEY.md §2 in one launch: TT:1002-1052.
template <class T, bool DR = false>
__global__ __launch_bounds__(kBlock) void step_kernel(const StepConsts K, DevBuffers b, const float* __restrict__ actions, int serve_on, const DRTables drt = DRTables{}) {
    __shared__ float s_obs[kBlock * kObsStride];
    const int lane = threadIdx.x;
    PP_STAMP_AT(0);
    const int n = K.num_envs;
    const int base = blockIdx.x * kBlock;
    const int i = base + lane;
    const int nvalid = min(kBlock, n - base);

    const bool active = i < n;
    EnvState st;
    float rew = 0.f;
    long long reset = 0;
    if (active) {
        // actions [N,7] row-major: the wave's 64 rows are one contiguous 1792-byte run, so the seven
        // strided dword loads of a lane hit the same 28 cache lines (L1 serves the re-touches)
        float act[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) act[d] = actions[(size_t)i * ND + d];
        load_state<1>(b, n, i, st);
#if defined(PP_STAMP)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // attribute the load latency to phase 0->1
#endif
        PP_STAMP_AT(1);
        BodyState bodies[NB];
        float pre_vx;
#if PP_ABLATE >= 2   // profiling builds only (tools/gpu_ablate.sh): skip the physics
        pre_vx = st.ball.v.x;
#pragma unroll
        for (int j = 0; j < NB; j++) { bodies[j].pos = mk(act[0], act[1], (float)j); bodies[j].lin = mk(act[2], act[3], act[4]); }
#else
        EnvDR dr;
        const uint32_t gid = (uint32_t)(K.env_id_offset + i), ep0 = st.episode;
        if (DR) {   // this env's entries of the randomisation tables (a NULL table = scale 1)
#pragma unroll
            for (int d = 0; d < ND; d++) {
                dr.kp[d] = drt.kp ? drt.kp[(size_t)d * n + i] : 1.f;
                dr.kd[d] = drt.kd ? drt.kd[(size_t)d * n + i] : 1.f;
                dr.ms[d] = drt.ms ? drt.ms[(size_t)d * n + i] : 1.f;
            }
            dr.es = drt.es ? drt.es[i] : 1.f;
            dr.fs = drt.fs ? drt.fs[i] : 1.f;
            dr.act_sigma = drt.act_sigma; dr.obs_sigma = drt.obs_sigma;
            dr.key_progress = (uint32_t)st.progress;
        }
        simulate_env<T, 1, DR>(K, act, st, bodies, pre_vx, &dr, gid);
#endif
        V3 ov = mk(0, 0, 0);
        if (serve_on) ov = mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i]);
        LdsRowStore store{&s_obs[lane * kObsStride]};
        NoisyRowStore nstore{&s_obs[lane * kObsStride], dr.obs_sigma, K.seed, gid, ep0, dr.key_progress};
#if PP_ABLATE >= 3   // skip reward / reset / observations as well: loads + stores only
        rew = pre_vx; reset = 0;
#pragma unroll
        for (int k = 0; k < PPENV_NUM_OBS; k++) store(k, bodies[k % NB].pos.x);
#else
        if (DR && drt.obs_sigma > 0.f) post_physics_env<1>(K, gid, st, bodies, pre_vx, serve_on ? &ov : nullptr, &rew, reset, &nstore);
        else post_physics_env<1>(K, gid, st, bodies, pre_vx, serve_on ? &ov : nullptr, &rew, reset, &store);
#endif
        PP_STAMP_AT(9);
    }
    // One wave per workgroup: DS operations of a wave execute in order, so the tile written above is visible
    // to the reads below without a barrier, and nothing forces the vector-memory queue to drain.  Every lane
    // takes part in the flush (a ragged last workgroup has lanes without an env of their own).
    __builtin_amdgcn_wave_barrier();
    flush_obs(s_obs, b.obs, base, nvalid, lane);
    if (active) {
        store_state<1>(b, n, i, st, &rew, reset);   // fire and forget: nothing in this launch reads the state again
    }
    PP_STAMP_AT(10);
}

// ------------------------------------------------- the fused step, two waves per 64 envs
// Same arithmetic as step_kernel, different schedule.  At N = 16384 step_kernel puts one wave on each CU
// and leaves three of its four SIMDs idle, and a lone wave is bound by its own serial instruction
// stream (one VALU instruction per >= 4 cycles).  Here a workgroup is two waves that own the same 64 envs.
// Within a substep the arm and the ball depend only on the state at the substep's start (ball_substep), so:
//   wave 0 ("arm")  : joint space.  actions -> PD targets; per substep the velocity recursion, ABA and
//                     integration, publishing (q, qd) in LDS after each one.
//   wave 1 ("ball") : world space.  Per substep its own FK sweep of the published (q, qd) for the collision
//                     geometry, then the ball's micro-stepped contacts; substep 0 starts straight from the
//                     loaded state, substep s >= 1 as soon as the arm wave has published boundary s.
// After the last integration the arm wave runs the FK of the final state, publishes the paddle position and
// writes the 60 body-observation values; the ball wave computes reward, masked reset and the last 20 values;
// then each flushes the obs columns it wrote and stores its half of the state.  Every hand-off is one-way (an
// LDS sequence number the consumer polls: the producer never waits); the only s_barrier is the one after the
// flags are initialised.  (q, qd) of substep boundary b travel through LDS slot b - 1 (one per boundary, never reused).
struct NullVisitor {
    __device__ __forceinline__ void operator()(int, const M3&, V3, V3, V3) {}
};
// obs tile -> obs_buf for the columns [C0, C1) (both multiples of 4) of `nrows` tile rows starting at tile row
// `t0`, as float4.  Tile row t0 + r goes to obs_buf row `g0 + r * gstep` (the tile is [agent][lane], obs_buf rows
// are A * env + agent).
template <int C0, int C1>
__device__ __forceinline__ void flush_obs_cols(const float* s_obs, float* obs, int t0, int nrows, size_t g0, int gstep, int lane) {
    constexpr int per_row = (C1 - C0) / 4;
    const int total = nrows * per_row;
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (nrows == kBlock) {
        // full workgroup (every one but a ragged last): the trip count is a constant, so all LDS reads of the flush are in
        // flight before the first store instead of one read-wait-store round trip per iteration
        f4v val[per_row];
#pragma unroll
        for (int it = 0; it < per_row; it++) {
            const int k = it * kBlock + lane;
            const int r = k / per_row, c = C0 + 4 * (k - r * per_row);
            const float* src = &s_obs[(t0 + r) * kObsStride + c];
            val[it] = f4v{src[0], src[1], src[2], src[3]};
        }
#pragma unroll
        for (int it = 0; it < per_row; it++) {
            const int k = it * kBlock + lane;
            const int r = k / per_row, c = C0 + 4 * (k - r * per_row);
            __builtin_nontemporal_store(val[it], reinterpret_cast<f4v*>(obs + ((g0 + (size_t)r * gstep) * PPENV_NUM_OBS + c)));
        }
        return;
    }
    for (int k = lane; k < total; k += kBlock) {
        int r = k / per_row, c = C0 + 4 * (k - r * per_row);
        const float* src = &s_obs[(t0 + r) * kObsStride + c];
        f4v val = {src[0], src[1], src[2], src[3]};
        __builtin_nontemporal_store(val, reinterpret_cast<f4v*>(obs + ((g0 + (size_t)r * gstep) * PPENV_NUM_OBS + c)));
    }
}

// One-way hand-off inside the workgroup: the arm wave publishes data in LDS and then a sequence number; the
// ball wave polls it.  (An s_barrier would make the producer wait for the consumer as well.)  Both waves of a
// workgroup are co-resident, so the producer always makes progress; the poll is bounded all the same.
__device__ __forceinline__ void publish(int* flag, int value) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Returns false when the partner never publishes (2^22 polls, ~0.2 s): the caller then reports it in the handle's status word
// and leaves WITHOUT storing anything — stale LDS must never reach HBM as a plausible-looking state.
__device__ __forceinline__ bool await(int* flag, int value) {
    for (int spin = 0; spin < (1 << 22); spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= value) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
// the status word lives in pinned host memory (ppenv::status_host): the host reads it at its next call without a synchronisation
__device__ __forceinline__ void report_fault(uint32_t* status, uint32_t bit) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#define PP_AWAIT(flag, value)                                                                        \
    do {                                                                                             \
        if (!await(flag, value)) { report_fault(status, PPENV_STATUS_HANDOFF_TIMEOUT); return; }     \
    } while (0)

// the parts of an arm's collision geometry that move with it: paddle centre / normal / their rates, then end points and
// velocities of the link-attached shapes
template <class T>
struct MovingGeom {
    static constexpr int count() {
        int c = 12;
        for (int s = 0; s < T::kShapes; s++) c += T::shape_link(s) >= 0 ? 12 : 0;
        return c;
    }
    template <class F>
    __device__ __forceinline__ static void each(ArmGeom<T::kShapes>& g, F&& f) {
        int k = 0;
        V3* head[4] = {&g.pc, &g.pn, &g.vpc, &g.pnd};
#pragma unroll
        for (int t = 0; t < 4; t++) { f(k, head[t]->x); f(k + 1, head[t]->y); f(k + 2, head[t]->z); k += 3; }
#pragma unroll
        for (int s = 0; s < T::kShapes; s++)
            if (T::shape_link(s) >= 0) {
                V3* v[4] = {&g.a[s], &g.b[s], &g.va[s], &g.vb[s]};
#pragma unroll
                for (int t = 0; t < 4; t++) { f(k, v[t]->x); f(k + 1, v[t]->y); f(k + 2, v[t]->z); k += 3; }
            }
    }
};

// A = humanoids per env, G = who sweeps the collision geometry.  <A=1, G=0>: two waves (arm, ball); the arm wave is the
// critical path, so it stays in joint space and the ball wave runs the world-space FK sweep for the geometry itself.
// <A=2, G=0>: the same with three waves, one per arm on its own base (K.site[arm]).  <A=2, G=1> (the 4-actor default): with
// two humanoids to collide against the ball wave is the critical path (35k cycles against the arm waves' 21k), so each
// arm wave's start-of-substep sweep is the world-space one and hands its geometry to the ball wave through two alternating
// LDS slots; the ball wave is left with the contacts.  (Tried and dropped: separate geometry waves — five waves on four
// SIMDs slow each other more than the hand-off saves.)
// BW = ball waves per 64 envs (round 3).  What a ball wave pays per micro-step is the UNION of its lanes' contact branches (DESIGN.md §6:
// 250-420 instructions where a ball in free flight needs 40, because with 64 envs in a wave some lane is near nearly every shape).  At
// N = 16384 half of the chip's SIMDs have no wave at all, so the 64 envs of a workgroup can be dealt to BW ball waves of 64 / BW envs
// each (the upper lanes idle): fewer lanes per wave = a smaller union = a shorter chain, on SIMDs that were empty anyway.  The arm wave
// keeps all 64 envs (its instruction stream has no divergence to shrink).
// DR (round 3): the domain-randomisation tables read by the wave that owns the quantity — drive gains and link masses, the
// action noise and the body block's observation noise on the arm wave; restitution / friction scales and the rest of the row's noise on the
// ball wave.  Same arithmetic as step_kernel<T, true> (the one-wave instantiation: 427 VGPRs + spills), on the fast schedule.
template <class T, int A, int G, int BW = 1, bool DR = false>
__global__ __launch_bounds__((A + BW) * kBlock) void step_kernel_split(const StepConsts K, DevBuffers b, const float* __restrict__ actions, int serve_on,
                                                                       uint32_t* status, int dbg_drop_handoff, const DRTables drt = DRTables{}) {
    static_assert(BW == 1 || G == 0, "narrow ball waves: each sweeps its own geometry (the s_bflag slot protocol has one consumer)");
    // (DR with two humanoids: both are instances of the yaml's one "humanoid" actor — an env's [7][N] table entries apply to both arms)
    constexpr int kGeo = MovingGeom<T>::count();
    // Who writes dof_pos / dof_vel / dof_force.  With one humanoid the arm wave is the critical path and would sit waiting
    // for the ball wave's reset decision just to pick between q and the initial pose: the ball wave, which has both, stores
    // them instead.  With two humanoids the ball wave is the critical path and the arm waves keep the stores.
#ifndef PP_BALL_STORES_DOFS
#define PP_BALL_STORES_DOFS 1      // profiling builds: 0 = the arm wave stores the dof state of the one-humanoid variants too (it waits for the reset decision)
#endif
    constexpr bool kBallStoresDofs = A == 1 && PP_BALL_STORES_DOFS;
    // With several ball waves each publishes s_flag_ball = 1 on its own and the arm wave's wait is satisfied by the FIRST of them: were the arm wave to
    // read the reset decisions of all 64 envs after that wait (the PP_BALL_STORES_DOFS=0 diagnostic build), it would race with the slower ball waves.
    static_assert(A != 1 || BW == 1 || kBallStoresDofs, "BW > 1 needs PP_BALL_STORES_DOFS: the arm wave must not read s_reset behind a flag any one ball wave sets");
    __shared__ float s_geom[G ? 2 : 1][G ? A : 1][G ? kGeo : 1][G ? kBlock : 1];   // geometry of boundary s in slot s & 1
    __shared__ int s_gflag[A];                         // arm -> ball: boundaries whose geometry is in LDS
    __shared__ int s_bflag;                            // ball -> arm: substeps the ball has finished (slot reuse)
    __shared__ float s_obs[A * kBlock * kObsStride];   // tile row = agent * kBlock + lane
    __shared__ float s_q[kMaxSplitSubsteps][A * 2 * ND][kBlock];   // (q, qd) at substep boundary b = 1 .. substeps, slot b - 1: never reused
    __shared__ float s_tau[A * ND][kBlock];            // drive torques of the last substep (dof_force)
    __shared__ float s_paddle[A * 3][kBlock];          // paddle position of the final state (the reward reads it)
    __shared__ int s_reset[kBlock];                    // the ball wave's reset decision, for the arm waves' dof stores
    __shared__ float s_serve[3][kBlock];               // arm wave 0 -> ball: the serve of a possible reset (published with the paddle position)
    __shared__ int s_flag[A];                          // arm -> ball: boundaries published so far; substeps + 1 = paddle position too
    __shared__ int s_flag_ball;                        // ball -> arm: 1 once the reset decision is in s_reset
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n = K.num_envs;
    const int base = blockIdx.x * kBlock;
    const int i = base + lane;
    const int nvalid = min(kBlock, n - base);
    const bool active = i < n;
    const int substeps = K.substeps;
    if (threadIdx.x <= A) { if (threadIdx.x < A) { s_flag[threadIdx.x] = 0; s_gflag[threadIdx.x] = 0; } else { s_flag_ball = 0; s_bflag = 0; } }
    __syncthreads();   // the only rendezvous of the launch: the flags are initialised

    if (wave < A) {
        // ------------------------------------------------------------------ arm wave
        const int arm = wave;
        const ArmSite& S = K.site[A == 1 ? 0 : arm];
        PP_STAMP_AT(0);
        float q[ND], qd[ND], target[ND], tau[ND];
        JointSave js[ND];
        EnvDR dr;
        uint32_t dr_ep0 = 0;
        if (active) {
            if (DR) {   // this env's gains / masses (a NULL table = scale 1) and the keys of its noise draws: episode and progress at the step's start
#pragma unroll
                for (int d = 0; d < ND; d++) {
                    dr.kp[d] = drt.kp ? drt.kp[(size_t)d * n + i] : 1.f;
                    dr.kd[d] = drt.kd ? drt.kd[(size_t)d * n + i] : 1.f;
                    dr.ms[d] = drt.ms ? drt.ms[(size_t)d * n + i] : 1.f;
                }
                dr.act_sigma = drt.act_sigma; dr.obs_sigma = drt.obs_sigma;
                dr.key_progress = (uint32_t)b.progress[(size_t)i * A];
                dr_ep0 = b.episode[i];
            }
#pragma unroll
            for (int d = 0; d < ND; d++) {
                q[d] = b.dof_pos[(size_t)(arm * ND + d) * n + i];
                qd[d] = b.dof_vel[(size_t)(arm * ND + d) * n + i];
                tau[d] = 0.f;
                float act = actions[((size_t)i * A + arm) * ND + d];
                if (DR) { if (dr.act_sigma > 0.f) act += dr.act_sigma * dr_gauss(K.seed, (uint32_t)(K.env_id_offset + i), dr_ep0, dr.key_progress, (uint32_t)(arm * ND + d)); }   // as simulate_env: before the clamp
                target[d] = pd_target(act, T::drive(d).lower, T::drive(d).upper, K.clip_actions);   // VecTask.step clamp + TT:1008
            }
        }
        PP_STAMP_AT(1);
        for (int s = 0; s < substeps; s++) {
            if (G) {
                if (s >= 2) PP_AWAIT(&s_bflag, s - 1);     // the ball is done with the slot's previous content (substep s - 2)
                if (active) {
                    ArmGeom<T::kShapes> gg;             // world-space sweep: velocity recursion + the collision geometry of boundary s
                    GeomVisitor<T> gv(gg);
                    fk_sweep<T>(S, q, qd, js, gv);
                    MovingGeom<T>::each(gg, [&](int k, float& v) { s_geom[s & 1][arm][k][lane] = v; });
                }
                publish(&s_gflag[arm], s + 1);
            }
            if (active) {
                if (!G) {
                    NullVisitor nv;   // velocity recursion only: the world transforms of this sweep are dead code
                    fk_sweep<T>(S, q, qd, js, nv);
                }
                arm_substep<T, DR>(S, js, q, qd, target, K.h, tau, &dr);
#pragma unroll
                for (int d = 0; d < ND; d++) { s_q[s][arm * 2 * ND + d][lane] = q[d]; s_q[s][arm * 2 * ND + ND + d][lane] = qd[d]; }
                if (s + 1 == substeps) {
#pragma unroll
                    for (int d = 0; d < ND; d++) s_tau[arm * ND + d][lane] = tau[d];
                }
            }
            publish(&s_flag[arm], s + 1);   // (q, qd) of boundary s+1 (and, last time, the drive torques)
            PP_STAMP_AT(2 + 2 * s);
        }
        // FK of the final state: paddle position for the reward, then the body block of the observation row,
        // obs[0:60] (TT:1696-1697) — it depends on the pre-reset body states only (TT:1039)
        // (a bare pose-chain sweep publishing the paddle position ~1k cycles before the full sweep was tried: no effect on the
        // step time once the ball wave's substeps ended later than the full sweep, so there is one sweep)
        BodyState bodies[NB];
        PP_STAMP_AT(5);
        if (arm == 0 && active) {
            // The serve this env gets if it resets at the end of the step: the counter RNG is a pure function of (seed, env id,
            // episode + 1).  The ball wave is the longer one, so the draw sits here, between this wave's last substep and its
            // final sweep, and travels with the paddle position.
            V3 sv = serve_on ? mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i])
                             : serve_velocity(K, (uint32_t)(K.env_id_offset + i), b.episode[i] + 1u);
            s_serve[0][lane] = sv.x; s_serve[1][lane] = sv.y; s_serve[2][lane] = sv.z;
        }
        if (active) {
            ArmGeom<T::kShapes> g;
            BodyVisitor<T, false> bv(g, bodies);
            fk_sweep<T>(S, q, qd, js, bv);
            static_body<false>(S, bodies[0]);
            s_paddle[arm * 3 + 0][lane] = bodies[NB - 1].pos.x; s_paddle[arm * 3 + 1][lane] = bodies[NB - 1].pos.y; s_paddle[arm * 3 + 2][lane] = bodies[NB - 1].pos.z;
        }
        if (!dbg_drop_handoff) publish(&s_flag[arm], substeps + 1);   // (dbg: tests force the partner's time-out path, PPENV_DEBUG_DROP_HANDOFF)
        if (active) {
            V3 bpos[NB], bvel[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) { bpos[j] = bodies[j].pos; bvel[j] = bodies[j].lin; }
            LdsRowStore store{&s_obs[(arm * kBlock + lane) * kObsStride]};
            NoisyRowStore nstore{&s_obs[(arm * kBlock + lane) * kObsStride], dr.obs_sigma, K.seed, (uint32_t)(K.env_id_offset + i), dr_ep0, dr.key_progress,
                                 16u + (uint32_t)(arm * PPENV_NUM_OBS)};
            if (DR && drt.obs_sigma > 0.f) write_obs_bodies(bpos, bvel, S.hinv, nstore);
            else write_obs_bodies(bpos, bvel, S.hinv, store);
        }
        PP_STAMP_AT(6);
        __builtin_amdgcn_wave_barrier();   // columns [0,60) of this agent's rows were written by this wave only: no rendezvous needed
        flush_obs_cols<0, 6 * NB>(s_obs, b.obs, arm * kBlock, nvalid, (size_t)base * A + arm, A, lane);
        if (kBallStoresDofs) return;       // (A = 1) the ball wave holds the final dof state and the reset decision: it stores them
        PP_AWAIT(&s_flag_ball, 1);         // the ball wave's reset decision
        PP_STAMP_AT(7);
        if (active) {
            const bool rst = s_reset[lane] != 0 && K.rc.variant != PPENV_VARIANT_TN;   // TN:888-901 keeps the dof state
#pragma unroll
            for (int d = 0; d < ND; d++) {
                st_state(&b.dof_pos[(size_t)(arm * ND + d) * n + i], rst ? K.init_dof_pos[d] : q[d]);
                st_state(&b.dof_vel[(size_t)(arm * ND + d) * n + i], rst ? K.init_dof_vel[d] : qd[d]);
                st_state(&b.dof_force[(size_t)(arm * ND + d) * n + i], tau[d]);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- ball wave(s)
    PP_STAMP_AT(16);
    constexpr int EB = kBlock / BW;                 // envs per ball wave
    const int bw = wave - A;                        // which of the BW ball waves
    const int env = bw * EB + lane;                 // this lane's env within the workgroup (lanes >= EB of a narrow ball wave idle)
    const int bi = base + env;
    const bool bactive = lane < EB && bi < n;
    EnvStateT<A> st;
    float rew[A], pre_vx = 0.f;
    long long reset = 0;
    V3 next_serve = mk(0, 0, 0);
    ArmGeom<T::kShapes> g[A];
    V3 bound[A];
    float qs[A * ND], qds[A * ND];
    EnvDR bdr;
    uint32_t bdr_ep0 = 0;
#pragma unroll
    for (int a = 0; a < A; a++) rew[a] = 0.f;
    if (bactive) {
        if (!G) {
#pragma unroll
            for (int d = 0; d < A * ND; d++) { qs[d] = b.dof_pos[(size_t)d * n + bi]; qds[d] = b.dof_vel[(size_t)d * n + bi]; }
        }
        float bl[13];
#pragma unroll
        for (int k = 0; k < 13; k++) bl[k] = b.ball[(size_t)k * n + bi];
        st.ball.p = mk(bl[0], bl[1], bl[2]);
#pragma unroll
        for (int k = 0; k < 4; k++) st.ball.quat[k] = bl[3 + k];
        st.ball.v = mk(bl[7], bl[8], bl[9]);
        st.ball.w = mk(bl[10], bl[11], bl[12]);
        st.progress = b.progress[(size_t)bi * A];
#pragma unroll
        for (int a = 0; a < A; a++) st.flags[a] = b.flags[(size_t)a * n + bi];
        st.episode = b.episode[bi];
        pre_vx = st.ball.v.x;   // TT:1020
#pragma unroll
        for (int a = 0; a < A; a++) { static_geometry<T>(K.site[a], g[a]); bound[a] = ld3(K.site[a].bound_center); }
        if (DR) {
            bdr.es = drt.es ? drt.es[bi] : 1.f;
            bdr.fs = drt.fs ? drt.fs[bi] : 1.f;
            bdr.obs_sigma = drt.obs_sigma;
            bdr.key_progress = (uint32_t)st.progress;
            bdr_ep0 = st.episode;
        }
    }
    PP_STAMP_AT(17);
    for (int s = 0; s < substeps; s++) {
        if (G) {
#pragma unroll
            for (int a = 0; a < A; a++) PP_AWAIT(&s_gflag[a], s + 1);   // the geometry waves have boundary s in LDS
        } else if (s > 0) {
#pragma unroll
            for (int a = 0; a < A; a++) PP_AWAIT(&s_flag[a], s);   // the arm waves have published boundary s
            if (bactive) {
#pragma unroll
                for (int a = 0; a < A; a++)
#pragma unroll
                    for (int d = 0; d < ND; d++) { qs[a * ND + d] = s_q[s - 1][a * 2 * ND + d][env]; qds[a * ND + d] = s_q[s - 1][a * 2 * ND + ND + d][env]; }
            }
        }
        PP_STAMP_AT(18 + 2 * s);
        if (bactive) {
#pragma unroll
            for (int a = 0; a < A; a++) {
                if (G) {
                    MovingGeom<T>::each(g[a], [&](int k, float& v) { v = s_geom[s & 1][a][k][env]; });
                } else {
                    JointSave js[ND];        // dead: only the geometry (points + velocities) of this sweep is used
                    GeomVisitor<T> gv(g[a]);
                    fk_sweep<T>(K.site[a], &qs[a * ND], &qds[a * ND], js, gv);
                }
            }
            PP_STAMP_AT(25);
            ball_substep<T, A, DR>(K, st.ball, g, bound, &bdr);
        }
        if (G) publish(&s_bflag, s + 1);
        PP_STAMP_AT(19 + 2 * s);
    }
#pragma unroll
    for (int a = 0; a < A; a++) PP_AWAIT(&s_flag[a], substeps + 1);   // final dof state, drive torques and paddle position
    if (bactive) {
        BodyState bodies[A * NB];   // the task part reads the pelvis (row 0) and the paddle (row 9) only
        LdsRowStore stores[A];
#pragma unroll
        for (int a = 0; a < A; a++) {
#pragma unroll
            for (int d = 0; d < ND; d++) {
                st.q[a * ND + d] = s_q[substeps - 1][a * 2 * ND + d][env];
                st.qd[a * ND + d] = s_q[substeps - 1][a * 2 * ND + ND + d][env];
                st.dof_force[a * ND + d] = s_tau[a * ND + d][env];
            }
            static_body<false>(K.site[a], bodies[a * NB]);
            bodies[a * NB + NB - 1].pos = mk(s_paddle[a * 3 + 0][env], s_paddle[a * 3 + 1][env], s_paddle[a * 3 + 2][env]);
            stores[a].row = &s_obs[(a * kBlock + env) * kObsStride];
        }
        next_serve = mk(s_serve[0][env], s_serve[1][env], s_serve[2][env]);   // drawn by arm wave 0
        if (DR && drt.obs_sigma > 0.f) {
            NoisyRowStore nstores[A];
#pragma unroll
            for (int a = 0; a < A; a++)
                nstores[a] = NoisyRowStore{stores[a].row, bdr.obs_sigma, K.seed, (uint32_t)(K.env_id_offset + bi), bdr_ep0, bdr.key_progress, 16u + (uint32_t)(a * PPENV_NUM_OBS)};
            post_physics_env<A, false>(K, (uint32_t)(K.env_id_offset + bi), st, bodies, pre_vx, &next_serve, rew, reset, nstores);
        } else
            post_physics_env<A, false>(K, (uint32_t)(K.env_id_offset + bi), st, bodies, pre_vx, &next_serve, rew, reset, stores);
        s_reset[env] = (int)reset;
    }
    publish(&s_flag_ball, 1);          // hands the reset decision to the arm waves
    PP_STAMP_AT(22);
    __builtin_amdgcn_wave_barrier();   // columns [60,80) were written by this wave only
    PP_STAMP_AT(23);
#pragma unroll
    for (int a = 0; a < A; a++)
        flush_obs_cols<6 * NB, PPENV_NUM_OBS>(s_obs, b.obs, a * kBlock + bw * EB, BW == 1 ? nvalid : max(0, min(EB, nvalid - bw * EB)),
                                              (size_t)(base + bw * EB) * A + a, A, lane);
    if (bactive) {
        if (kBallStoresDofs) {             // st.q / st.qd already show the reset state where the env reset (TN keeps its dof state)
#pragma unroll
            for (int d = 0; d < A * ND; d++) {
                st_state(&b.dof_pos[(size_t)d * n + bi], st.q[d]);
                st_state(&b.dof_vel[(size_t)d * n + bi], st.qd[d]);
                st_state(&b.dof_force[(size_t)d * n + bi], st.dof_force[d]);
            }
        }
        store_ball(b, n, bi, st.ball);
        store_task<A>(b, n, bi, st, rew, reset);
    }
    PP_STAMP_AT(24);
}

// ---- the single-humanoid step on FOUR waves per 64 envs (round 2; PPENV_STEP_KERNEL=quad) ---------------------------------------------
//
// step_kernel_split<T, 1, 0> is two co-critical serial chains: the arm wave (two substeps, then the final-state sweep, the
// body block of the observation row and its flush) and the ball wave (per substep a kinematics sweep for the collision geometry, then
// four micro-steps; then reward / reset / the rest of the row, its flush and the state stores).  Here two more waves — on the two SIMDs
// the launch leaves idle — take what is not inherently on either chain:
//   wave 2 (geometry): the ball wave's kinematics sweep, from the substep boundary the arm wave publishes, into an LDS slot; and, after
//           the final sweep, a third of the observation row's body block and of the flush;
//   wave 3 (auxiliary): the serve draw of a possible reset (a pure function of seed, env id, episode), then a third of the body block
//           and of the flush.
// Hand-offs during the substeps are one-way flags as in the two-wave kernel (the consumers wait anyway).  The tail is two workgroup
// barriers, at which a waiting wave costs no issue slot: X — final sweep done (paddle position, raw body states, final dof state in
// LDS; the ball wave's substeps done): the three helpers transform the bodies, the ball wave runs the task arithmetic; Y — the whole
// observation tile is in LDS: waves 0, 2, 3 flush it, the ball wave stores the state.  Nothing reaches HBM before Y, so a hand-off
// time-out (bounded wait, as elsewhere) simply marks the workgroup dead and nobody stores.
template <class T>
__global__ __launch_bounds__(4 * kBlock) void step_kernel_quad(const StepConsts K, DevBuffers b, const float* __restrict__ actions, int serve_on,
                                                               uint32_t* status, int dbg_drop_handoff) {
    constexpr int A = 1, kGeo = MovingGeom<T>::count(), kHelpFirst = 4;   // bodies 0 .. kHelpFirst-1: arm wave; then wave 2 and wave 3 split the rest
    constexpr int kMid = kHelpFirst + (NB - kHelpFirst) / 2;
    __shared__ float s_geom[2][kGeo][kBlock];          // geometry of boundary s in slot s & 1
    __shared__ float s_obs[kBlock * kObsStride];
    __shared__ float s_q[kMaxSplitSubsteps][2 * ND][kBlock];
    __shared__ float s_tau[ND][kBlock];
    __shared__ float s_paddle[3][kBlock];
    __shared__ float s_body[6 * (NB - kHelpFirst)][kBlock];   // final-state position / velocity of the bodies the helpers transform
    __shared__ float s_serve[3][kBlock];
    __shared__ int s_flag, s_gflag, s_bflag, s_dead;   // arm: boundaries published; geometry: boundaries swept; ball: substeps done; a wait timed out
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n = K.num_envs;
    const int base = blockIdx.x * kBlock;
    const int i = base + lane;
    const int nvalid = min(kBlock, n - base);
    const bool active = i < n;
    const int substeps = K.substeps;
    const ArmSite& S = K.site[0];
    if (threadIdx.x == 0) { s_flag = 0; s_gflag = 0; s_bflag = 0; s_dead = 0; }
    __syncthreads();
    bool dead = false;
#define QD_AWAIT(flag, value)                                                                             \
    do {                                                                                                  \
        if (!dead && !await(flag, value)) {                                                               \
            dead = true;                                                                                  \
            report_fault(status, PPENV_STATUS_HANDOFF_TIMEOUT);                                           \
            if (lane == 0) __hip_atomic_store(&s_dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }                                                                                                 \
    } while (0)
    // rows of the tile this wave flushes after Y (waves 0, 2, 3 = parts 0, 1, 2)
    auto flush_part = [&](int part) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        constexpr int per_row = PPENV_NUM_OBS / 4;
        const int total = nvalid * per_row;
        for (int k = part * kBlock + lane; k < total; k += 3 * kBlock) {
            const int r = k / per_row, c = 4 * (k - r * per_row);
            const float* src = &s_obs[r * kObsStride + c];
            const f4v val = {src[0], src[1], src[2], src[3]};
            __builtin_nontemporal_store(val, reinterpret_cast<f4v*>(b.obs + ((size_t)(base + r) * PPENV_NUM_OBS + c)));
        }
    };
    // a helper's share of the body block: bodies J0 .. J1-1 from s_body, the pelvis (the fixed base) from the site constants
    auto help_bodies = [&](auto j0c, auto j1c) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        if (!active) return;
        V3 bpos[NB], bvel[NB];
        BodyState root;
        static_body<false>(S, root);
        bpos[0] = root.pos;
#pragma unroll
        for (int j = J0; j < J1; j++) {
            const int o = 6 * (j - kHelpFirst);
            bpos[j] = mk(s_body[o][lane], s_body[o + 1][lane], s_body[o + 2][lane]);
            bvel[j] = mk(s_body[o + 3][lane], s_body[o + 4][lane], s_body[o + 5][lane]);
        }
        LdsRowStore store{&s_obs[lane * kObsStride]};
        write_obs_bodies<J0, J1>(bpos, bvel, S.hinv, store);
    };

    if (wave == 0) {
        // ------------------------------------------------------------------ arm wave
        PP_STAMP_AT(0);
        float q[ND], qd[ND], target[ND], tau[ND];
        JointSave js[ND];
        if (active) {
#pragma unroll
            for (int d = 0; d < ND; d++) {
                q[d] = b.dof_pos[(size_t)d * n + i];
                qd[d] = b.dof_vel[(size_t)d * n + i];
                tau[d] = 0.f;
                target[d] = pd_target(actions[(size_t)i * ND + d], T::drive(d).lower, T::drive(d).upper, K.clip_actions);   // VecTask.step clamp + TT:1008
            }
        }
        PP_STAMP_AT(1);
        for (int s = 0; s < substeps; s++) {
            if (active) {
                NullVisitor nv;   // velocity recursion only
                fk_sweep<T>(S, q, qd, js, nv);
                arm_substep<T>(S, js, q, qd, target, K.h, tau);
#pragma unroll
                for (int d = 0; d < ND; d++) { s_q[s][d][lane] = q[d]; s_q[s][ND + d][lane] = qd[d]; }
                if (s + 1 == substeps) {
#pragma unroll
                    for (int d = 0; d < ND; d++) s_tau[d][lane] = tau[d];
                }
            }
            if (!dbg_drop_handoff) publish(&s_flag, s + 1);   // (dbg: tests force the partners' time-out path, PPENV_DEBUG_DROP_HANDOFF)
            PP_STAMP_AT(2 + 2 * s);
        }
        BodyState bodies[NB];
        PP_STAMP_AT(5);
        if (active) {
            ArmGeom<T::kShapes> g;
            BodyVisitor<T, false> bv(g, bodies);
            fk_sweep<T>(S, q, qd, js, bv);
            static_body<false>(S, bodies[0]);
            s_paddle[0][lane] = bodies[NB - 1].pos.x; s_paddle[1][lane] = bodies[NB - 1].pos.y; s_paddle[2][lane] = bodies[NB - 1].pos.z;
#pragma unroll
            for (int j = kHelpFirst; j < NB; j++) {
                const int o = 6 * (j - kHelpFirst);
                s_body[o][lane] = bodies[j].pos.x; s_body[o + 1][lane] = bodies[j].pos.y; s_body[o + 2][lane] = bodies[j].pos.z;
                s_body[o + 3][lane] = bodies[j].lin.x; s_body[o + 4][lane] = bodies[j].lin.y; s_body[o + 5][lane] = bodies[j].lin.z;
            }
        }
        PP_STAMP_AT(6);
        __syncthreads();   // X
        if (active) {
            V3 bpos[NB], bvel[NB];
#pragma unroll
            for (int j = 0; j < kHelpFirst; j++) { bpos[j] = bodies[j].pos; bvel[j] = bodies[j].lin; }
            LdsRowStore store{&s_obs[lane * kObsStride]};
            write_obs_bodies<0, kHelpFirst>(bpos, bvel, S.hinv, store);
        }
        __syncthreads();   // Y
        PP_STAMP_AT(7);
        if (!s_dead) flush_part(0);
        PP_STAMP_AT(8);
        return;
    }
    if (wave == 2) {
        // ------------------------------------------------------------------ geometry wave
        float qs[ND], qds[ND];
        ArmGeom<T::kShapes> gg;
        if (active) {
#pragma unroll
            for (int d = 0; d < ND; d++) { qs[d] = b.dof_pos[(size_t)d * n + i]; qds[d] = b.dof_vel[(size_t)d * n + i]; }
        }
        for (int s = 0; s < substeps; s++) {
            if (s > 0) {
                QD_AWAIT(&s_flag, s);                          // the arm wave has published boundary s
                if (active) {
#pragma unroll
                    for (int d = 0; d < ND; d++) { qs[d] = s_q[s - 1][d][lane]; qds[d] = s_q[s - 1][ND + d][lane]; }
                }
            }
            if (s >= 2) QD_AWAIT(&s_bflag, s - 1);            // the ball is done with the slot's previous content (substep s - 2)
            if (active && !dead) {
                JointSave js[ND];        // dead code: only the geometry (points + velocities) of this sweep is used
                GeomVisitor<T> gv(gg);
                fk_sweep<T>(S, qs, qds, js, gv);
                MovingGeom<T>::each(gg, [&](int k, float& v) { s_geom[s & 1][k][lane] = v; });
            }
            if (!dead) publish(&s_gflag, s + 1);
        }
        __syncthreads();   // X
        help_bodies(std::integral_constant<int, kHelpFirst>{}, std::integral_constant<int, kMid>{});
        __syncthreads();   // Y
        if (!s_dead) flush_part(1);
        return;
    }
    if (wave == 3) {
        // ------------------------------------------------------------------ auxiliary wave
        if (active) {
            // The serve this env gets if it resets at the end of the step: the counter RNG is a pure function of (seed, env id, episode + 1)
            V3 sv = serve_on ? mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i])
                             : serve_velocity(K, (uint32_t)(K.env_id_offset + i), b.episode[i] + 1u);
            s_serve[0][lane] = sv.x; s_serve[1][lane] = sv.y; s_serve[2][lane] = sv.z;
        }
        __syncthreads();   // X
        help_bodies(std::integral_constant<int, kMid>{}, std::integral_constant<int, NB>{});
        __syncthreads();   // Y
        if (!s_dead) flush_part(2);
        return;
    }

    // ---------------------------------------------------------------------- ball wave
    PP_STAMP_AT(16);
    EnvStateT<A> st;
    float rew[A], pre_vx = 0.f;
    long long reset = 0;
    V3 next_serve = mk(0, 0, 0);
    ArmGeom<T::kShapes> g[A];
    V3 bound[A];
    rew[0] = 0.f;
    if (active) {
        float bl[13];
#pragma unroll
        for (int k = 0; k < 13; k++) bl[k] = b.ball[(size_t)k * n + i];
        st.ball.p = mk(bl[0], bl[1], bl[2]);
#pragma unroll
        for (int k = 0; k < 4; k++) st.ball.quat[k] = bl[3 + k];
        st.ball.v = mk(bl[7], bl[8], bl[9]);
        st.ball.w = mk(bl[10], bl[11], bl[12]);
        st.progress = b.progress[(size_t)i];
        st.flags[0] = b.flags[i];
        st.episode = b.episode[i];
        pre_vx = st.ball.v.x;   // TT:1020
        static_geometry<T>(S, g[0]);
        bound[0] = ld3(S.bound_center);
    }
    PP_STAMP_AT(17);
    for (int s = 0; s < substeps; s++) {
        QD_AWAIT(&s_gflag, s + 1);                             // the geometry wave has boundary s in LDS
        PP_STAMP_AT(18 + 2 * s);
        if (active && !dead) {
            MovingGeom<T>::each(g[0], [&](int k, float& v) { v = s_geom[s & 1][k][lane]; });
            ball_substep<T, A>(K, st.ball, g, bound);
        }
        if (!dead) publish(&s_bflag, s + 1);
        PP_STAMP_AT(19 + 2 * s);
    }
    QD_AWAIT(&s_flag, substeps);                               // (only fails when the arm wave withheld a publish: it then never reaches X either way)
    __syncthreads();   // X: final dof state, drive torques, paddle position, serve
    if (active && !dead && !s_dead) {
        BodyState bodies[NB];   // the task part reads the pelvis (row 0) and the paddle (row 9) only
        LdsRowStore stores[A];
#pragma unroll
        for (int d = 0; d < ND; d++) {
            st.q[d] = s_q[substeps - 1][d][lane];
            st.qd[d] = s_q[substeps - 1][ND + d][lane];
            st.dof_force[d] = s_tau[d][lane];
        }
        static_body<false>(S, bodies[0]);
        bodies[NB - 1].pos = mk(s_paddle[0][lane], s_paddle[1][lane], s_paddle[2][lane]);
        stores[0].row = &s_obs[lane * kObsStride];
        next_serve = mk(s_serve[0][lane], s_serve[1][lane], s_serve[2][lane]);
        post_physics_env<A, false>(K, (uint32_t)(K.env_id_offset + i), st, bodies, pre_vx, &next_serve, rew, reset, stores);
    }
    PP_STAMP_AT(22);
    __syncthreads();   // Y
    PP_STAMP_AT(23);
    if (active && !s_dead) {   // st.q / st.qd already show the reset state where the env reset (TN keeps its dof state)
#pragma unroll
        for (int d = 0; d < ND; d++) {
            st_state(&b.dof_pos[(size_t)d * n + i], st.q[d]);
            st_state(&b.dof_vel[(size_t)d * n + i], st.qd[d]);
            st_state(&b.dof_force[(size_t)d * n + i], st.dof_force[d]);
        }
        store_ball(b, n, i, st.ball);
        store_task<A>(b, n, i, st, rew, reset);
    }
    PP_STAMP_AT(24);
#undef QD_AWAIT
}

// create (mode 0: creation is episode 0) / reset_all (mode 1: next episode): state as after
// _create_envs (TT:512-643) plus the observations of that state
template <class T, int A>
__global__ __launch_bounds__(kBlock) void init_kernel(const StepConsts K, DevBuffers b, int mode, int serve_on) {
    __shared__ float s_obs[A * kBlock * kObsStride];   // tile row = agent * kBlock + lane
    const int n = K.num_envs;
    const int base = blockIdx.x * kBlock;
    const int lane = threadIdx.x;
    const int i = base + lane;
    const int nvalid = min(kBlock, n - base);
    if (i < n) {
        EnvStateT<A> st;
        st.episode = mode == 0 ? 0u : b.episode[i] + 1u;
        V3 serve = serve_on ? mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i])
                            : serve_velocity(K, (uint32_t)(K.env_id_offset + i), st.episode);
        reset_state(K, st, serve, true);
#pragma unroll
        for (int d = 0; d < A * ND; d++) st.dof_force[d] = 0.f;
        st.progress = 0;
        float rew[A];
#pragma unroll
        for (int a = 0; a < A; a++) {
            st.flags[a] = PPENV_FLAG_NO_BOUNCE;
            rew[a] = 0.f;
            BodyState bodies[NB];
            bodies_of_state<T>(K.site[a], &st.q[a * ND], &st.qd[a * ND], bodies);
            V3 bpos[NB], bvel[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) { bpos[j] = bodies[j].pos; bvel[j] = bodies[j].lin; }
            LdsRowStore store{&s_obs[(a * kBlock + lane) * kObsStride]};
            write_obs(bpos, bvel, K.site[a].hinv, &st.q[a * ND], &st.qd[a * ND], st.ball.p, st.ball.v, store);
        }
        store_state<A>(b, n, i, st, rew, 1);   // reset_buf: upstream VecTask.allocate_buffers has ones; overwritten by the first step (TT:740)
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < A; a++)
        flush_obs_cols<0, PPENV_NUM_OBS>(s_obs, b.obs, a * kBlock, nvalid, (size_t)base * A + a, A, lane);
}

// reset_idx(env_ids) -> _reset_idx (TT:809-812, 847-906): one lane per listed env; everything it writes is that env's own
// (scattered SoA elements and, with refresh_obs, its obs rows), so duplicate ids only repeat the same stores.
template <class T, int A>
__global__ __launch_bounds__(kBlock) void reset_idx_kernel(const StepConsts K, DevBuffers b, const long long* __restrict__ ids, int count, int refresh_obs,
                                                           int serve_on, uint32_t* status) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= count) return;
    const int n = K.num_envs;
    const long long id = ids[t];
    if (id < 0 || id >= n) return;                                         // validated on the host when the ids are host-visible; never index out of range
    const int i = (int)id;
    EnvStateT<A> st;
#pragma unroll
    for (int d = 0; d < A * ND; d++) { st.q[d] = b.dof_pos[(size_t)d * n + i]; st.qd[d] = b.dof_vel[(size_t)d * n + i]; }
    st.episode = b.episode[i] + 1u;
    const V3 serve = serve_on ? mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i])
                              : serve_velocity(K, (uint32_t)(K.env_id_offset + i), st.episode);
    reset_state(K, st, serve, K.rc.variant != PPENV_VARIANT_TN);        // TN:888-901 keeps the dof state
#pragma unroll
    for (int d = 0; d < A * ND; d++) { b.dof_pos[(size_t)d * n + i] = st.q[d]; b.dof_vel[(size_t)d * n + i] = st.qd[d]; }
    store_ball(b, n, i, st.ball);
    b.episode[i] = st.episode;
#pragma unroll
    for (int a = 0; a < A; a++) {
        b.progress[(size_t)i * A + a] = 0;                                  // TT:902
        b.flags[(size_t)a * n + i] = PPENV_FLAG_NO_BOUNCE;                  // TT:903-905
        if (refresh_obs) {
            BodyState bodies[NB];
            bodies_of_state<T>(K.site[a], &st.q[a * ND], &st.qd[a * ND], bodies);
            V3 bpos[NB], bvel[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) { bpos[j] = bodies[j].pos; bvel[j] = bodies[j].lin; }
            LdsRowStore store{b.obs + ((size_t)i * A + a) * PPENV_NUM_OBS};   // a plain row pointer here: a rare, scattered write
            write_obs(bpos, bvel, K.site[a].hinv, &st.q[a * ND], &st.qd[a * ND], st.ball.p, st.ball.v, store);
        }
    }
}

// pre_physics_step alone: PD targets of [rows, 7] actions (ppenv_pd_targets) — the step kernels' own pd_target()
template <class T>
__global__ void pd_targets_kernel(int rows, float clip, const float* __restrict__ actions, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * ND) return;
    const int d = t % ND;
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int k = 0; k < ND; k++)
        if (k == d) { lo = T::drive(k).lower; hi = T::drive(k).upper; }
    out[t] = pd_target(actions[t], lo, hi, clip);
}
// generate_random_speed_for_ball on explicit draws (ppenv_serve_from_draws) — the reset path's own serve_from_draws()
__global__ void serve_from_draws_kernel(int form, int m, const float* __restrict__ draws, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const V3 v = serve_from_draws(form, draws[3 * t], draws[3 * t + 1], draws[3 * t + 2]);
    out[3 * t] = v.x; out[3 * t + 1] = v.y; out[3 * t + 2] = v.z;
}

// ---------------------------------------- Isaac-Gym tensor-API mode: TT:1022-1039 on caller tensors
__global__ __launch_bounds__(kBlock) void post_physics_kernel(const StepConsts K, DevBuffers b, const float* __restrict__ rb,
                                                               float* root, float* dofs, const float* __restrict__ dof_force,
                                                               const float* __restrict__ pre_vx, int serve_on) {
    __shared__ float s_obs[kBlock * kObsStride];
    const int n = K.num_envs;
    const int base = blockIdx.x * kBlock;
    const int lane = threadIdx.x;
    const int i = base + lane;
    const int nvalid = min(kBlock, n - base);
    if (i < n) {
        const int body_ids[NB] = {0, 31, 32, 33, 34, 35, 36, 37, 38, 39};   // bodyStatesId, HumanoidPingpongTiltG1.yaml:47
        const float* rbe = rb + (size_t)i * PPENV_NUM_BODIES * 13;
        float* roote = root + (size_t)i * PPENV_NUM_ACTORS * 13;
        float* dofe = dofs + (size_t)i * ND * 2;
        V3 bpos[NB], bvel[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const float* r = rbe + body_ids[j] * 13;
            bpos[j] = mk(r[0], r[1], r[2]);
            bvel[j] = mk(r[7], r[8], r[9]);
        }
        float root_quat[4] = {rbe[3], rbe[4], rbe[5], rbe[6]};
        EnvState st;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            st.q[d] = dofe[2 * d];
            st.qd[d] = dofe[2 * d + 1];
            st.dof_force[d] = dof_force[(size_t)i * ND + d];
        }
        float* be = roote + 2 * 13;
        st.ball.p = mk(be[0], be[1], be[2]);
        st.ball.v = mk(be[7], be[8], be[9]);
        st.progress = b.progress[i] + 1;                                   // TT:1023
        st.flags[0] = b.flags[i];
        st.episode = b.episode[i];
        RewardIn in;
        in.humanoid_x = roote[0];
        in.paddle = bpos[NB - 1];
        in.pre_vx = pre_vx[i];
        in.bp = st.ball.p;
        in.vx = st.ball.v.x;
        float power = 0.f;
#pragma unroll
        for (int d = 0; d < ND; d++) power += fabsf(st.dof_force[d] * st.qd[d]);
        in.power = power;
        in.progress = st.progress;
        long long reset;
        float rew = compute_reward(K.rc, in, st.flags[0], reset);
        if (reset) {                                                        // TT:847-906
            st.episode += 1;
            V3 serve = serve_on ? mk(b.serve[i], b.serve[(size_t)n + i], b.serve[2 * (size_t)n + i])
                                : serve_velocity(K, (uint32_t)(K.env_id_offset + i), st.episode);
            const float* ipos[3] = {K.site[0].root_pos, K.table_pos, K.ball_init_pos};
            const float* iquat[3] = {K.site[0].root_quat, K.table_quat, K.ball_init_quat};
#pragma unroll
            for (int a = 0; a < 3; a++) {                                   // TT:853-855
#pragma unroll
                for (int k = 0; k < 3; k++) roote[a * 13 + k] = ipos[a][k];
#pragma unroll
                for (int k = 0; k < 4; k++) roote[a * 13 + 3 + k] = iquat[a][k];
#pragma unroll
                for (int k = 7; k < 13; k++) roote[a * 13 + k] = 0.f;
            }
            be[7] = serve.x; be[8] = serve.y; be[9] = serve.z;              // TT:857-862
            st.ball.p = ld3(K.ball_init_pos);
            st.ball.v = serve;
            if (K.rc.variant != PPENV_VARIANT_TN) {                          // TN:888-901 keeps the dof state
#pragma unroll
                for (int d = 0; d < ND; d++) {
                    st.q[d] = K.init_dof_pos[d]; st.qd[d] = K.init_dof_vel[d];
                    dofe[2 * d] = st.q[d]; dofe[2 * d + 1] = st.qd[d];
                }
            }
            st.progress = 0;
            st.flags[0] = PPENV_FLAG_NO_BOUNCE;
        }
        LdsRowStore store{&s_obs[lane * kObsStride]};
        float hinv[4];
        heading_quat_inv(root_quat, hinv);
        write_obs(bpos, bvel, hinv, st.q, st.qd, st.ball.p, st.ball.v, store);
        b.progress[i] = st.progress;
        b.flags[i] = st.flags[0];
        b.episode[i] = st.episode;
        b.rew[i] = rew;
        b.reset[i] = reset;
    }
    __syncthreads();
    flush_obs(s_obs, b.obs, base, nvalid, lane);
}

// ------------------------------------------------------------ gym.refresh_* equivalents
// Layouts: 3-actor [N,3,13] / [N,7,2] / [N,7] / [N,42,13] (TT:166-183,208-211); 4-actor [N,4,13] (humanoid1, humanoid2,
// table, ball: T4:181-185) / [N,14,2] / [N,14] / [N,82,13] (humanoid1 0-39, humanoid2 40-79, table 80, ball 81: T4:169-172).
__global__ void refresh_root_kernel(const StepConsts K, DevBuffers b, float* out) {
    const int n = K.num_envs, A = K.num_arms;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rows = A + 2;
    float* r = out + (size_t)i * rows * 13;
    for (int k = 0; k < rows * 13; k++) r[k] = 0.f;
    for (int a = 0; a < A; a++) {
        for (int k = 0; k < 3; k++) r[a * 13 + k] = K.site[a].root_pos[k];
        for (int k = 0; k < 4; k++) r[a * 13 + 3 + k] = K.site[a].root_quat[k];
    }
    for (int k = 0; k < 3; k++) r[A * 13 + k] = K.table_pos[k];
    for (int k = 0; k < 4; k++) r[A * 13 + 3 + k] = K.table_quat[k];
    for (int k = 0; k < 13; k++) r[(A + 1) * 13 + k] = b.ball[(size_t)k * n + i];
}
__global__ void refresh_dof_kernel(const StepConsts K, DevBuffers b, float* out) {
    const int n = K.num_envs, D = K.num_arms * ND;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int d = 0; d < D; d++) {
        out[((size_t)i * D + d) * 2] = b.dof_pos[(size_t)d * n + i];
        out[((size_t)i * D + d) * 2 + 1] = b.dof_vel[(size_t)d * n + i];
    }
}
__global__ void refresh_dof_force_kernel(const StepConsts K, DevBuffers b, float* out) {
    const int n = K.num_envs, D = K.num_arms * ND;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int d = 0; d < D; d++) out[(size_t)i * D + d] = b.dof_force[(size_t)d * n + i];
}
template <class T>
__global__ __launch_bounds__(kBlock) void refresh_rb_kernel(const StepConsts K, DevBuffers b, float* out) {
    const int n = K.num_envs, A = K.num_arms;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* rb = out + (size_t)i * (A * PPENV_NUM_HUMANOID_BODIES + 2) * 13;
    for (int a = 0; a < A; a++) {
        const ArmSite& S = K.site[a];
        float q[ND], qd[ND];
        for (int d = 0; d < ND; d++) { q[d] = b.dof_pos[(size_t)(a * ND + d) * n + i]; qd[d] = b.dof_vel[(size_t)(a * ND + d) * n + i]; }
        BodyState bodies[NB];
        bodies_of_state<T>(S, q, qd, bodies);
        float* rba = rb + a * PPENV_NUM_HUMANOID_BODIES * 13;
        for (int body = 0; body < PPENV_NUM_HUMANOID_BODIES; body++) {
            float* r = rba + body * 13;
            for (int k = 0; k < 3; k++) r[k] = S.root_pos[k];
            for (int k = 0; k < 4; k++) r[3 + k] = S.root_quat[k];
            for (int k = 7; k < 13; k++) r[k] = 0.f;
        }
        const int body_ids[NB] = {0, 31, 32, 33, 34, 35, 36, 37, 38, 39};
#pragma unroll
        for (int j = 0; j < NB; j++) {
            float* r = rba + body_ids[j] * 13;
            r[0] = bodies[j].pos.x; r[1] = bodies[j].pos.y; r[2] = bodies[j].pos.z;
            float qt[4];
            rot_to_quat(bodies[j].rot, qt);
            r[3] = qt[0]; r[4] = qt[1]; r[5] = qt[2]; r[6] = qt[3];
            r[7] = bodies[j].lin.x; r[8] = bodies[j].lin.y; r[9] = bodies[j].lin.z;
            r[10] = bodies[j].ang.x; r[11] = bodies[j].ang.y; r[12] = bodies[j].ang.z;
        }
    }
    float* t = rb + A * PPENV_NUM_HUMANOID_BODIES * 13;
    for (int k = 0; k < 13; k++) t[k] = 0.f;
    for (int k = 0; k < 3; k++) t[k] = K.table_pos[k];
    for (int k = 0; k < 4; k++) t[3 + k] = K.table_quat[k];
    float* bl = t + 13;
    for (int k = 0; k < 13; k++) bl[k] = b.ball[(size_t)k * n + i];
}

// sums of rew / progress / episode over the envs (ppenv_reduce_stats): wave reduction + one atomic per wave
__global__ __launch_bounds__(256) void stats_kernel(int n, int agents, const float* __restrict__ rew, const long long* __restrict__ progress,
                                                     const uint32_t* __restrict__ episode, double* out) {
    double r = 0.0, p = 0.0, e = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {   // i = agents * env + agent
        r += (double)rew[i];
        p += (double)progress[i];
        e += (double)episode[i / agents];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off, 64);
        p += __shfl_down(p, off, 64);
        e += __shfl_down(e, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], r);
        atomicAdd(&out[1], p);
        atomicAdd(&out[2], e);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[3] = (double)n;
}

// The same sums by ONE workgroup of 16 waves, written (not accumulated) into out: no memset before it, no atomics, a fixed
// summation order.  Used up to 32768 rows, where one CU streams the 0.5 MB in a few microseconds.
__global__ __launch_bounds__(1024) void stats_kernel_single(int n, int agents, const float* __restrict__ rew, const long long* __restrict__ progress,
                                                             const uint32_t* __restrict__ episode, double* out) {
    __shared__ double s_part[16][3];
    // integer partial sums per thread where exact; all loads of an 8-row batch in flight together
    double rf = 0.0;
    long long pi = 0;
    unsigned int ei = 0;
    const int sh = agents - 1;   // agents is 1 or 2: row -> env
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += 1024) {
        rf += (double)rew[i];
        pi += progress[i];
        ei += episode[i >> sh];
    }
    double r = rf, p = (double)pi, e = (double)ei;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off, 64);
        p += __shfl_down(p, off, 64);
        e += __shfl_down(e, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6][0] = r; s_part[threadIdx.x >> 6][1] = p; s_part[threadIdx.x >> 6][2] = e; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += s_part[w][threadIdx.x];
        out[threadIdx.x] = t;
    }
    if (threadIdx.x == 3) out[3] = (double)n;
}

// serve override [N,3] row-major -> SoA [3][N]
__global__ void serve_transpose_kernel(int n, const float* in, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; k++) out[(size_t)k * n + i] = in[(size_t)i * 3 + k];
}

// ------------------------------------------------------------------------- host side
thread_local char g_err[512] = "";
void set_err(const char* fmt, const char* a = "", const char* b = "") { snprintf(g_err, sizeof g_err, fmt, a, b); }
}  // namespace
void ppenv_set_error(const char* msg) { set_err("%s", msg); }   // for the other translation units of the library
namespace {

#define PP_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) {                                        \
            set_err("%s failed: %s", #call, hipGetErrorString(e_));    \
            return PPENV_EHIP;                                         \
        }                                                              \
    } while (0)

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t obs, rew, reset, progress, dof_pos, dof_vel, dof_force, ball, flags, episode, serve, total;
};
int agents_of(const ppenv_config* c) { return c->variant == PPENV_VARIANT_T4 ? 2 : 1; }
Layout layout_for(int n, int A) {
    Layout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
    l.obs = take((size_t)n * A * PPENV_NUM_OBS * 4);
    l.rew = take((size_t)n * A * 4);
    l.reset = take((size_t)n * A * 8);
    l.progress = take((size_t)n * A * 8);
    l.dof_pos = take((size_t)n * A * ND * 4);
    l.dof_vel = take((size_t)n * A * ND * 4);
    l.dof_force = take((size_t)n * A * ND * 4);
    l.ball = take((size_t)n * 13 * 4);
    l.flags = take((size_t)n * A * 4);
    l.episode = take((size_t)n * 4);
    l.serve = take((size_t)n * 3 * 4);
    l.total = o;
    return l;
}

bool validate(const ppenv_config* c) {
    if (!c) { set_err("config is NULL"); return false; }
    if (c->abi_version != PPENV_ABI_VERSION) { set_err("config.abi_version does not match this library"); return false; }
    if (c->num_envs <= 0) { set_err("num_envs must be positive"); return false; }
    if (c->variant < PPENV_VARIANT_T3 || c->variant > PPENV_VARIANT_T4) { set_err("unknown task variant"); return false; }
    if (c->num_humanoids != agents_of(c)) { set_err("num_humanoids must be 2 for PPENV_VARIANT_T4 and 1 otherwise"); return false; }
    if (agents_of(c) == 2 && c->substeps > kMaxSplitSubsteps) { set_err("the 4-actor variant supports at most 4 substeps"); return false; }
    if (c->substeps < 1 || c->substeps > 16 || c->ball_substeps < 1 || c->ball_substeps > 64) { set_err("substeps / ball_substeps out of range"); return false; }
    if (!(c->dt > 0.f)) { set_err("dt must be positive"); return false; }
    {   // the serve's sines and cosines are a degree-9 / degree-8 series (sincos_small): fp32-exact up to ~1 rad, not beyond
        const float lim = 57.0f;
        if (fabsf(c->serve_tilt_lo_deg) > lim || fabsf(c->serve_tilt_hi_deg) > lim || fabsf(c->serve_tilt_z_lo_deg) > lim || fabsf(c->serve_tilt_z_hi_deg) > lim) {
            set_err("serve tilt angles must lie within +-57 degrees (the kernel's small-angle sine / cosine series)");
            return false;
        }
    }
    if (!model_matches<ModelG1>(*c)) {
        set_err("the config's arm model (joint frames / inertials / gains / limits / link shapes / paddle / observed-body frames) differs from "
                "the one compiled into this library (csrc/ppenv_model_g1.h); generate its header with isaacgym_amd.modelgen and rebuild "
                "(isaacgym_amd._lib.build_for_arm_model)");
        return false;
    }
    for (int d = 0; d < ND; d++)
        if (!(c->joint[d].mass > 0.f) || !(c->joint[d].lower <= c->joint[d].upper)) { set_err("joint table: mass must be > 0 and lower <= upper"); return false; }
    return true;
}

int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

struct ppenv {
    ppenv_config cfg;
    StepConsts K;            // run-time constants derived from cfg; passed to every kernel by value (kernarg)
    DevBuffers buf;
    Layout lay;
    void* arena;
    bool owns_arena;
    int serve_on;
    int split;               // 1: step_kernel_split (two waves per 64 envs; three for the 4-actor variant), 2: 4-actor with the arm waves sweeping the geometry, 0: step_kernel
    int agents;              // 1, or 2 for PPENV_VARIANT_T4
    uint32_t* status_host;   // PPENV_STATUS_* bits, pinned host memory mapped into the device: kernels write it through, the host reads it without a sync
    uint32_t* status_dev;
    int ball_waves;          // ball waves per 64 envs of the two-wave schedule (1, 2 or 4: step_kernel_split's BW)
    int dbg_drop_handoff;    // PPENV_DEBUG_DROP_HANDOFF=1 at create (tests): the arm wave withholds its last hand-off, so the partner waves time out
    int dr_on;               // a randomisation is set: ppenv_step launches step_kernel<ModelG1, true> with these tables
    DRTables dr;
};

namespace {
int use_device(const ppenv* e) {
    int cur = -1;
    PP_HIP(hipGetDevice(&cur));
    if (cur != e->cfg.device_id) PP_HIP(hipSetDevice(e->cfg.device_id));
    return PPENV_OK;
}
// every entry point that reads or advances the state: refuse to go on once a kernel has reported a fault
int check_status(const ppenv* e) {
    const uint32_t st = *(volatile uint32_t*)e->status_host;
    if (st == 0) return PPENV_OK;
    char msg[200];
    snprintf(msg, sizeof msg, "device status 0x%x: %s; the environment state is no longer valid (destroy the handle)", st,
             (st & PPENV_STATUS_HANDOFF_TIMEOUT) ? "a step-kernel wave timed out waiting for its partner wave's LDS hand-off and did not store its envs" : "unknown fault");
    set_err("%s", msg);
    return PPENV_EDEVICE;
}
}  // namespace

extern "C" {

int ppenv_abi_version(void) { return PPENV_ABI_VERSION; }
const char* ppenv_last_error(void) { return g_err; }

size_t ppenv_arena_bytes(const ppenv_config* cfg) {
    if (!cfg || cfg->num_envs <= 0) return 0;
    return layout_for(cfg->num_envs, agents_of(cfg)).total;
}

int ppenv_create(const ppenv_config* cfg, void* arena_dev, size_t arena_bytes, void* stream, ppenv** out) {
    if (!out) { set_err("out is NULL"); return PPENV_EINVAL; }
    *out = nullptr;
    if (!validate(cfg)) return PPENV_EINVAL;
    int ndev = 0;
    PP_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { set_err("device_id out of range"); return PPENV_EINVAL; }
    ppenv* e = new (std::nothrow) ppenv;
    if (!e) { set_err("out of host memory"); return PPENV_ENOMEM; }
    e->cfg = *cfg;
    e->agents = agents_of(cfg);
    e->lay = layout_for(cfg->num_envs, e->agents);
    e->serve_on = 0;
    {   // PPENV_STEP_KERNEL=fused|split forces a schedule (same arithmetic either way)
        // The two-wave schedule wins at every size measured (us per step, split vs one-wave: N = 16384 13.1 / 20.5,
        // 65536 19.6 / 22.8, 131072 35.3 / 41.5): it needs 187 VGPRs (two waves per SIMD) against 256 + 79 AGPRs.
        const char* k = getenv("PPENV_STEP_KERNEL");
        e->split = k ? (strcmp(k, "fused") != 0) : 1;
        if (k && strcmp(k, "quad") == 0) e->split = 3;          // four waves per 64 envs (single-humanoid variants)
        if (k && strcmp(k, "split_g1") == 0) e->split = 4;      // two waves, the ARM wave sweeps the collision geometry (round-3 experiment, single-humanoid variants)
        if (cfg->substeps > kMaxSplitSubsteps) e->split = 0;   // one LDS hand-off slot per substep boundary
        if (e->agents == 2) e->split = (k && strcmp(k, "split3") == 0) ? 1 : 2;
        else if (e->split == 3 && cfg->substeps > kMaxSplitSubsteps) e->split = 0;   // 4-actor: arm waves sweep the geometry (default), or the ball wave
    }
    {   // PPENV_BALL_WAVES=1|2|4: ball waves per 64 envs (single-humanoid two-wave schedule only)
        const char* bwv = getenv("PPENV_BALL_WAVES");
        e->ball_waves = bwv ? atoi(bwv) : kDefaultBallWaves;
        if (e->ball_waves != 1 && e->ball_waves != 2 && e->ball_waves != 4) e->ball_waves = kDefaultBallWaves;
    }
    e->arena = nullptr;
    e->owns_arena = false;
    e->status_host = e->status_dev = nullptr;
    e->dr_on = 0;
    e->dr = DRTables{};
    {
        const char* d = getenv("PPENV_DEBUG_DROP_HANDOFF");
        e->dbg_drop_handoff = (d && d[0] == '1') ? 1 : 0;
    }
    if (hipSetDevice(cfg->device_id) != hipSuccess) { delete e; set_err("hipSetDevice failed"); return PPENV_EHIP; }
    if (hipHostMalloc((void**)&e->status_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&e->status_dev, e->status_host, 0) != hipSuccess) {
        if (e->status_host) (void)hipHostFree(e->status_host);
        delete e;
        set_err("allocating the device status word failed");
        return PPENV_ENOMEM;
    }
    *e->status_host = 0u;
    if (arena_dev) {
        if (arena_bytes < e->lay.total || ((uintptr_t)arena_dev & 255u)) {
            (void)hipHostFree(e->status_host);
            delete e;
            set_err("arena too small or not 256-byte aligned");
            return PPENV_EINVAL;
        }
        e->arena = arena_dev;
    } else {
        if (hipMalloc(&e->arena, e->lay.total) != hipSuccess) { (void)hipHostFree(e->status_host); delete e; set_err("hipMalloc of the env arena failed"); return PPENV_ENOMEM; }
        e->owns_arena = true;
    }
    char* a = (char*)e->arena;
    const Layout& l = e->lay;
    e->buf = DevBuffers{(float*)(a + l.obs), (float*)(a + l.rew), (long long*)(a + l.reset), (long long*)(a + l.progress),
                        (float*)(a + l.dof_pos), (float*)(a + l.dof_vel), (float*)(a + l.dof_force), (float*)(a + l.ball),
                        (uint32_t*)(a + l.flags), (uint32_t*)(a + l.episode), (float*)(a + l.serve)};
    e->K = make_step_consts(e->cfg);
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(e->arena, 0, l.total, s);
    if (err == hipSuccess) {
        if (e->agents == 2) hipLaunchKernelGGL((init_kernel<ModelG1, 2>), dim3(grid_for(cfg->num_envs)), dim3(kBlock), 0, s, e->K, e->buf, 0, 0);
        else hipLaunchKernelGGL((init_kernel<ModelG1, 1>), dim3(grid_for(cfg->num_envs)), dim3(kBlock), 0, s, e->K, e->buf, 0, 0);
        err = hipGetLastError();
    }
    if (err != hipSuccess) {
        set_err("initialising the env state failed: %s", hipGetErrorString(err));
        if (e->owns_arena) (void)hipFree(e->arena);
        (void)hipHostFree(e->status_host);
        delete e;
        return PPENV_EHIP;
    }
    *out = e;
    return PPENV_OK;
}

void ppenv_destroy(ppenv* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    if (e->owns_arena && e->arena) (void)hipFree(e->arena);
    if (e->status_host) (void)hipHostFree(e->status_host);
    delete e;
}

int ppenv_buffers_of(ppenv* e, ppenv_buffers* out) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    out->num_envs = e->cfg.num_envs;
    out->num_agents = e->agents;
    out->obs_buf = e->buf.obs; out->rew_buf = e->buf.rew;
    out->reset_buf = (int64_t*)e->buf.reset; out->progress_buf = (int64_t*)e->buf.progress;
    out->dof_pos = e->buf.dof_pos; out->dof_vel = e->buf.dof_vel; out->dof_force = e->buf.dof_force; out->ball = e->buf.ball;
    out->flags = e->buf.flags; out->episode = e->buf.episode; out->serve_override = e->buf.serve;
    return PPENV_OK;
}

int ppenv_config_of(ppenv* e, ppenv_config* out) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    *out = e->cfg;
    return PPENV_OK;
}

// Which kernel ppenv_step launches for this handle NOW (the randomisation state included).  One decision, two readers: launch_step
// switches on it, ppenv_step_kernel_name reports it — so what bench.py prints as `roofline.kernel` is what ran, not a guess from the
// environment.  The names are the demangled ones rocprofv3's kernel trace shows (default template arguments written out).
enum StepSchedule { SCH_DR_T4, SCH_DR_SPLIT, SCH_DR_FUSED, SCH_T4_ARMS_SWEEP, SCH_T4_BALL_SWEEPS, SCH_SPLIT_ARM_SWEEPS, SCH_QUAD, SCH_SPLIT_BW4, SCH_SPLIT_BW2,
                    SCH_SPLIT, SCH_FUSED };
static const char* const kScheduleNames[] = {
    "step_kernel_split<pp::ModelG1, 2, 1, 1, true>", "step_kernel_split<pp::ModelG1, 1, 0, 1, true>", "step_kernel<pp::ModelG1, true>",
    "step_kernel_split<pp::ModelG1, 2, 1, 1, false>", "step_kernel_split<pp::ModelG1, 2, 0, 1, false>", "step_kernel_split<pp::ModelG1, 1, 1, 1, false>",
    "step_kernel_quad<pp::ModelG1>", "step_kernel_split<pp::ModelG1, 1, 0, 4, false>", "step_kernel_split<pp::ModelG1, 1, 0, 2, false>",
    "step_kernel_split<pp::ModelG1, 1, 0, 1, false>", "step_kernel<pp::ModelG1, false>"};
static StepSchedule schedule_of(const ppenv* e) {
    if (e->dr_on)   // domain randomisation: the table-reading instantiation — of the two-wave schedule (default) or of the one-wave kernel (PPENV_STEP_KERNEL=fused)
        return e->agents == 2 ? SCH_DR_T4 : e->split ? SCH_DR_SPLIT : SCH_DR_FUSED;
    if (e->agents == 2) return e->split == 2 ? SCH_T4_ARMS_SWEEP : SCH_T4_BALL_SWEEPS;
    if (e->split == 4) return SCH_SPLIT_ARM_SWEEPS;
    if (e->split == 3) return SCH_QUAD;
    if (e->split) return e->ball_waves == 4 ? SCH_SPLIT_BW4 : e->ball_waves == 2 ? SCH_SPLIT_BW2 : SCH_SPLIT;
    return SCH_FUSED;
}

static int launch_step(ppenv* e, const DevBuffers& buf, const float* actions_dev, void* stream) {
    if (!e || !actions_dev) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = check_status(e)) return rc;
    if (int rc = use_device(e)) return rc;
    const dim3 grid(grid_for(e->cfg.num_envs));
    hipStream_t st = (hipStream_t)stream;
    switch (schedule_of(e)) {
    case SCH_DR_T4:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 2, 1, 1, true>), grid, dim3(3 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff, e->dr);
        break;
    case SCH_DR_SPLIT:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 1, 0, 1, true>), grid, dim3(2 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff, e->dr);
        break;
    case SCH_DR_FUSED:
        hipLaunchKernelGGL((step_kernel<ModelG1, true>), grid, dim3(kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->dr);
        break;
    case SCH_T4_ARMS_SWEEP:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 2, 1>), grid, dim3(3 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_T4_BALL_SWEEPS:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 2, 0>), grid, dim3(3 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_SPLIT_ARM_SWEEPS:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 1, 1>), grid, dim3(2 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_QUAD:
        hipLaunchKernelGGL((step_kernel_quad<ModelG1>), grid, dim3(4 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_SPLIT_BW4:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 1, 0, 4>), grid, dim3(5 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_SPLIT_BW2:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 1, 0, 2>), grid, dim3(3 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_SPLIT:
        hipLaunchKernelGGL((step_kernel_split<ModelG1, 1, 0>), grid, dim3(2 * kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, e->status_dev, e->dbg_drop_handoff);
        break;
    case SCH_FUSED:
        hipLaunchKernelGGL((step_kernel<ModelG1, false>), grid, dim3(kBlock), 0, st, e->K, buf, actions_dev, e->serve_on, DRTables{});
        break;
    }
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

const char* ppenv_step_kernel_name(ppenv* e) { return e ? kScheduleNames[schedule_of(e)] : ""; }

int ppenv_step(ppenv* e, const float* actions_dev, void* stream) { return launch_step(e, e ? e->buf : DevBuffers{}, actions_dev, stream); }

/* `count` steps in ONE call: step i on actions_dev[i], launched back to back from native code (include/ppenv.h).  Measured on the driver's 20-step run at
 * 16 384 envs (profiles/r04_d_smallk.txt): 1.03 G env-steps/s, the same as one replay of a 20-step graph (so the ~85 us such a short run loses are not
 * the graph launch but the first launch after an idle GPU and the final synchronisation), against 0.8-0.9 G for twenty Python calls. */
int ppenv_step_sequence(ppenv* e, const float* const* actions_dev, int32_t count, void* stream) {
    if (!e || !actions_dev || count < 0) { set_err("NULL argument or negative count"); return PPENV_EINVAL; }
    for (int32_t i = 0; i < count; i++)
        if (int rc = launch_step(e, e->buf, actions_dev[i], stream)) return rc;
    return PPENV_OK;
}

/* the same launch with this step's observations / rewards / reset flags going to the caller's tensors (NULL: the handle's own) */
int ppenv_step_into(ppenv* e, const float* actions_dev, float* obs_dev, float* rew_dev, int64_t* reset_dev, void* stream) {
    if (!e) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (obs_dev && (reinterpret_cast<uintptr_t>(obs_dev) & 15)) { set_err("ppenv_step_into: obs must be 16-byte aligned"); return PPENV_EINVAL; }
    DevBuffers b = e->buf;
    if (obs_dev) b.obs = obs_dev;
    if (rew_dev) b.rew = rew_dev;
    if (reset_dev) b.reset = reinterpret_cast<long long*>(reset_dev);
    return launch_step(e, b, actions_dev, stream);
}

int ppenv_reset_all(ppenv* e, void* stream) {
    if (!e) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = check_status(e)) return rc;
    if (int rc = use_device(e)) return rc;
    if (e->agents == 2) hipLaunchKernelGGL((init_kernel<ModelG1, 2>), dim3(grid_for(e->cfg.num_envs)), dim3(kBlock), 0, (hipStream_t)stream, e->K, e->buf, 1, e->serve_on);
    else hipLaunchKernelGGL((init_kernel<ModelG1, 1>), dim3(grid_for(e->cfg.num_envs)), dim3(kBlock), 0, (hipStream_t)stream, e->K, e->buf, 1, e->serve_on);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

uint32_t ppenv_status(ppenv* e) { return e ? *(volatile uint32_t*)e->status_host : 0u; }

int ppenv_set_randomization(ppenv* e, const ppenv_randomization* dr) {
    if (!e) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (!dr) { e->dr_on = 0; e->dr = DRTables{}; return PPENV_OK; }
    if (!(dr->action_noise_sigma >= 0.f) || !(dr->observation_noise_sigma >= 0.f)) { set_err("noise amplitudes must be >= 0"); return PPENV_EINVAL; }
    e->dr = DRTables{dr->dof_stiffness_scale, dr->dof_damping_scale, dr->link_mass_scale, dr->restitution_scale, dr->friction_scale,
                     dr->action_noise_sigma, dr->observation_noise_sigma};
    e->dr_on = 1;
    return PPENV_OK;
}

int ppenv_set_gravity(ppenv* e, float gravity_z) {
    if (!e) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (!(gravity_z <= 0.f)) { set_err("gravity_z must be <= 0 (the world's up axis is z)"); return PPENV_EINVAL; }
    e->cfg.gravity_z = gravity_z;
    e->K = make_step_consts(e->cfg);   // host-side: the next launch carries the new constants in its kernel argument
    return PPENV_OK;
}

int ppenv_reset_idx(ppenv* e, const int64_t* env_ids_dev, int32_t count, int refresh_obs, void* stream) {
    if (!e || (count > 0 && !env_ids_dev) || count < 0) { set_err("ppenv_reset_idx: NULL ids or negative count"); return PPENV_EINVAL; }
    if (count == 0) return PPENV_OK;
    if (int rc = check_status(e)) return rc;
    if (int rc = use_device(e)) return rc;
    const dim3 grid((count + kBlock - 1) / kBlock), block(kBlock);
    if (e->agents == 2)
        hipLaunchKernelGGL((reset_idx_kernel<ModelG1, 2>), grid, block, 0, (hipStream_t)stream, e->K, e->buf, (const long long*)env_ids_dev, count, refresh_obs, e->serve_on, e->status_dev);
    else
        hipLaunchKernelGGL((reset_idx_kernel<ModelG1, 1>), grid, block, 0, (hipStream_t)stream, e->K, e->buf, (const long long*)env_ids_dev, count, refresh_obs, e->serve_on, e->status_dev);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_pd_targets(ppenv* e, const float* actions_dev, float* pd_tar_dev, void* stream) {
    if (!e || !actions_dev || !pd_tar_dev) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    const int rows = e->cfg.num_envs * e->agents;
    hipLaunchKernelGGL(pd_targets_kernel<ModelG1>, dim3((rows * ND + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, e->K.clip_actions, actions_dev, pd_tar_dev);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_serve_from_draws(ppenv* e, const float* draws_dev, int32_t m, float* vel_dev, void* stream) {
    if (!e || !draws_dev || !vel_dev || m <= 0) { set_err("NULL argument or m <= 0"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(serve_from_draws_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->cfg.variant, m, draws_dev, vel_dev);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_reduce_stats(ppenv* e, double* out_dev, void* stream) {
    if (!e || !out_dev) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = check_status(e)) return rc;
    if (int rc = use_device(e)) return rc;
    const int rows = e->cfg.num_envs * e->agents;   // one row per agent: out[3] counts agents
    if (rows <= 32768) {
        hipLaunchKernelGGL(stats_kernel_single, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, e->agents, e->buf.rew, e->buf.progress,
                           e->buf.episode, out_dev);
    } else {
        PP_HIP(hipMemsetAsync(out_dev, 0, 4 * sizeof(double), (hipStream_t)stream));
        const int blocks = min(256, (rows + 255) / 256);
        hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, e->agents, e->buf.rew, e->buf.progress,
                           e->buf.episode, out_dev);
    }
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_post_physics_step(ppenv* e, const float* rigid_body_states_dev, float* root_states_dev, float* dof_states_dev,
                            const float* dof_force_dev, const float* pre_ball_vx_dev, void* stream) {
    if (!e || !rigid_body_states_dev || !root_states_dev || !dof_states_dev || !dof_force_dev || !pre_ball_vx_dev) {
        set_err("NULL argument");
        return PPENV_EINVAL;
    }
    if (e->agents != 1) { set_err("ppenv_post_physics_step serves the 3-actor variants; the 4-actor entry is ppenv_t4_rewards"); return PPENV_EINVAL; }
    if (int rc = check_status(e)) return rc;
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(post_physics_kernel, dim3(grid_for(e->cfg.num_envs)), dim3(kBlock), 0, (hipStream_t)stream, e->K, e->buf,
                       rigid_body_states_dev, root_states_dev, dof_states_dev, dof_force_dev, pre_ball_vx_dev, e->serve_on);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_refresh_root_states(ppenv* e, float* out, void* stream) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(refresh_root_kernel, dim3((e->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->K, e->buf, out);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}
int ppenv_refresh_dof_states(ppenv* e, float* out, void* stream) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(refresh_dof_kernel, dim3((e->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->K, e->buf, out);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}
int ppenv_refresh_dof_force(ppenv* e, float* out, void* stream) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(refresh_dof_force_kernel, dim3((e->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->K, e->buf, out);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}
int ppenv_refresh_rigid_body_states(ppenv* e, float* out, void* stream) {
    if (!e || !out) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    hipLaunchKernelGGL(refresh_rb_kernel<ModelG1>, dim3(grid_for(e->cfg.num_envs)), dim3(kBlock), 0, (hipStream_t)stream, e->K, e->buf, out);
    PP_HIP(hipGetLastError());
    return PPENV_OK;
}

int ppenv_set_serve_override(ppenv* e, const float* serve_dev, int on, void* stream) {
    if (!e) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (int rc = use_device(e)) return rc;
    if (on && serve_dev) {
        hipLaunchKernelGGL(serve_transpose_kernel, dim3((e->cfg.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->cfg.num_envs,
                           serve_dev, e->buf.serve);
        PP_HIP(hipGetLastError());
    }
    e->serve_on = on ? 1 : 0;
    return PPENV_OK;
}

#if defined(PP_STAMP)
// diagnostic builds only: copy the phase stamps of the last launch to the host
int ppenv_debug_read_stamps(unsigned long long* dst, size_t count) {
    if (hipDeviceSynchronize() != hipSuccess) return PPENV_EHIP;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pp_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : PPENV_EHIP;
}
#endif

size_t ppenv_state_bytes(ppenv* e) {
    if (!e) return 0;
    const size_t A = (size_t)e->agents;
    return (size_t)e->cfg.num_envs * ((A * ND * 3 + 13) * 4 + A * 4 + 4 + A * 8 + A * 8);
}

static int state_io(ppenv* e, char* blob, size_t nbytes, bool to_host) {
    if (!e || !blob) { set_err("NULL argument"); return PPENV_EINVAL; }
    if (nbytes != ppenv_state_bytes(e)) { set_err("state blob size does not match this handle"); return PPENV_ESTATE; }
    if (int rc = use_device(e)) return rc;
    PP_HIP(hipDeviceSynchronize());
    if (int rc = check_status(e)) return rc;       // after the synchronisation: a fault of the step still in flight is seen here
    const size_t n = (size_t)e->cfg.num_envs, A = (size_t)e->agents;
    struct Part { void* dev; size_t bytes; };
    const Part parts[] = {{e->buf.dof_pos, n * A * ND * 4}, {e->buf.dof_vel, n * A * ND * 4}, {e->buf.dof_force, n * A * ND * 4}, {e->buf.ball, n * 13 * 4},
                          {e->buf.flags, n * A * 4}, {e->buf.episode, n * 4}, {e->buf.progress, n * A * 8}, {e->buf.reset, n * A * 8}};
    for (const Part& p : parts) {
        if (to_host) PP_HIP(hipMemcpy(blob, p.dev, p.bytes, hipMemcpyDeviceToHost));
        else PP_HIP(hipMemcpy(p.dev, blob, p.bytes, hipMemcpyHostToDevice));
        blob += p.bytes;
    }
    return PPENV_OK;
}
int ppenv_get_state(ppenv* e, void* dst_host, size_t n) { return state_io(e, (char*)dst_host, n, true); }
int ppenv_set_state(ppenv* e, const void* src_host, size_t n) { return state_io(e, (char*)src_host, n, false); }

}  // extern "C"
