// ppenv_ta_chain.hip — the 27-DoF VecTask step, "chain-wave" schedule (gfx950).
//
// Mapping: one LANE per env, one WAVE per limb.  A workgroup is six waves that own the same 64 envs:
//     wave 0  left leg   links 1..6        wave 3  left arm   links 16..22
//     wave 1  right leg  links 7..12       wave 4  right arm  links 23..27 (+ the paddle and forearm collision shapes)
//     wave 2  pelvis + waist links 13..15  wave 5  ball, then reward / reset / observations of the 64 envs
// Every wave walks a COMPILE-TIME chain of the tree (ppenv_model_g1_ta.h, generated): all lanes of a wave visit the same link,
// so joint axes, frames, inertias, gains, contact points and welded bodies are instruction literals — identity joint frames
// and zero offsets fold away, nothing is read from a model table, and the per-link state between the three passes of the
// articulated-body algorithm stays in registers.  (The quad kernel of ppenv_ta_sim.hip puts the four limbs of one env on
// four lanes: its lanes visit different links at the same time, so its tables are run-time LDS data — 27.6k VALU
// instructions per wave, 45 % of them issuing, one wave per CU at 4096 envs.)
//
// The limbs only meet at the two hubs of the tree, and every hand-off is one-way through LDS (producer publishes data,
// then a sequence number; the consumer polls it with a bounded wait, cf. step_kernel_split in ppenv.hip):
//   pass 1 (kinematics, outwards)        waist -> arms: pose and twist of the torso
//   pass 2 (articulated inertia, inwards) arms -> waist: what they add to the torso; legs -> waist: what they add to the pelvis
//   base solve                            waist -> legs: acceleration of the pelvis;  waist -> arms: acceleration of the torso
//   integration                           waist -> everybody: the base state of the next substep
//   collision geometry                    waist / right arm -> ball wave: the shapes the ball can hit, at the substep's start
// The arithmetic is ppenv_ta_device.h's (link_dynamics, inward_step, outward_step, solve_base_art ... — the functions the
// other two mappings and the CPU host shim run); the task part restates ppenv_ta_task.h for one lane per env.
//
// Outputs leave coalesced: dof_states / dof_force / root_states / obs rows are assembled in LDS tiles and flushed by all six
// waves.  rigid_body_states [N,42,13] is only written when the caller passes a buffer for it (the reward and the observation
// read the body states from registers; gym.refresh_rigid_body_state_tensor has ppenv_ta_forward_kinematics).
#undef PP_STAMP
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>
#include <utility>

#include "ppenv_ta_device.h"
#include "ppenv_ta_task.h"
// PPENV_TA_MODEL_HEADER: a build for another 27-dof asset names the header isaacgym_amd/modelgen_ta.py generated from ITS tree
// (isaacgym_amd/_lib.py build_for_ta_model); the struct keeps its name, the static_asserts below hold it to the G1's topology.
#ifndef PPENV_TA_MODEL_HEADER
#define PPENV_TA_MODEL_HEADER "ppenv_model_g1_ta.h"
#endif
#include PPENV_TA_MODEL_HEADER
#include "ppenv_ta_chain.h"

using namespace pp;
using namespace pp::ta;

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
using T = ModelG1Tree;
constexpr int kE = 64;                 // lanes per wave = envs of a workgroup = columns of every LDS tile
// Row pitch of the [row][env] tiles.  A role reads and writes a tile with lane = env (conflict-free at any pitch); the staging at the start and
// the flush at the end move the Isaac-Gym-layout tensors ([env][row]) as flat float4 — lane = a run of four ROWS of one env — where a pitch of 64
// floats puts the lanes of an env into one LDS bank.  Round 4 measured the odd pitch (-DTA_TILE_PAD=1): the staging 5.9k -> 5.2k cycles and the state
// tiles' flush 4.6k -> 3.7k (stamps), the launch 30.4 -> 30.7 us in a same-box A/B (the roles' row addresses lose their shifts): not adopted.
#ifndef TA_TILE_PAD
#define TA_TILE_PAD 0
#endif
constexpr int kP = kE + TA_TILE_PAD;
// Envs per workgroup.  64 in the product.  Diagnostic builds (tools/gpu_ta_narrow.sh, round 3) set 32 or 16: the upper lanes of every wave idle (they chew on
// a copy of the workgroup's last env, like the lanes of a ragged last workgroup) and the grid has 2x / 4x the workgroups — at 4096 envs 128 / 256 CUs get one
// instead of 64.  Measured: the chain does not get shorter (a wave issues an instruction in four passes whatever its EXEC mask), see DESIGN.md §9.
#ifndef TA_ENVS_PER_WG
#define TA_ENVS_PER_WG 64
#endif
constexpr int kEPW = TA_ENVS_PER_WG;
static_assert(kEPW == 64 || kEPW == 32 || kEPW == 16, "envs per workgroup");
constexpr int kWaves = 6;
// Role -> wave index.  A workgroup's waves are dealt round-robin over the CU's four SIMDs (wave i on SIMD i & 3), so six waves leave two SIMDs with two waves
// each.  The critical path is waist -> arms -> waist, with the legs close behind.  Round 3 (tools/gpu_ta_rolemap.sh, profiles/r03_d_ta_rolemap.txt): the waist
// shares SIMD 0 with the RIGHT ARM — the two take turns by construction (the arm waits for the torso's pose, the waist for the arms' inertias, the arm for the
// torso's acceleration), so they hardly ever want the same issue slot — the right leg shares SIMD 1 with the (short) ball wave, and the left leg and the left arm
// (seven links) get SIMDs 2 and 3 to themselves: 30.6 us at 4096 envs.  Round 2's placement (TA_ROLE_MAP=0: the waist with the LEFT LEG, both arms alone) made
// the left leg the last to deliver its inertia (24.2k cycles into the step against 20.8k for the right leg): 31.4 us.  The waist with the left arm
// (TA_ROLE_MAP=2): 32.5 us.  (Round 2, first numbering — right arm on the left leg's SIMD: the right arm's five links took 15.7k cycles against 11k for the left's seven.)
#ifndef TA_ROLE_MAP
#define TA_ROLE_MAP 1
#endif
#if TA_ROLE_MAP == 1
enum { W_WAIST = 0, W_RL = 1, W_LL = 2, W_LA = 3, W_RA = 4, W_BALL = 5 };
#elif TA_ROLE_MAP == 2
enum { W_WAIST = 0, W_RL = 1, W_LL = 2, W_RA = 3, W_LA = 4, W_BALL = 5 };
#else
enum { W_WAIST = 0, W_RL = 1, W_LA = 2, W_RA = 3, W_LL = 4, W_BALL = 5 };
#endif
constexpr int kTorso = 15;
constexpr int kGeoW = 39, kGeoRA = 48; // floats of collision geometry the waist / right-arm wave hand to the ball wave

// Diagnostic builds only (-DTA_STAMP, tools/gpu_ta_chain_stamps.py): shader-clock stamps per workgroup and wave.
#if defined(TA_STAMP)
__device__ unsigned long long ta_chain_stamp_buf[1024 * 6 * 32];
#define CH_STAMP(k)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (lane == 0 && blockIdx.x < 1024) ta_chain_stamp_buf[(blockIdx.x * 6 + wave) * 32 + (k)] = t_; \
    } while (0)
#else
#define CH_STAMP(k) do { } while (0)
#endif

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr bool chain_ok(int first, int n, int parent) {
    for (int k = 0; k < n; k++)
        if (T::link(first + k).parent != (k == 0 ? parent : first + k - 1)) return false;
    return true;
}
static_assert(chain_ok(1, 6, 0) && chain_ok(7, 6, 0) && chain_ok(13, 3, 0) && chain_ok(16, 7, kTorso) && chain_ok(23, 5, kTorso), "chains of the G1 tree");
static_assert(T::kPaddleLink == 27 && T::kBoundLink == kTorso, "collision geometry owners");

// ---- LDS ----------------------------------------------------------------------------------------------------------------
// The tiles are [row][64]: lane e reads / writes element e of a row, so an access of a wave is one conflict-free 256-byte row.
// Two of the three Isaac-Gym-layout tensors (round 4) live here IN THEIR OWN LAYOUT instead, env-major — the workgroup's block of dof_states
// [64][27][2] and of actions / dof_force [64][27] — so that their staging at the start and their flush at the end are flat 16-byte copies (as
// [row][env] tiles they are transposed on the way, lane = four rows of one env: all the lanes of an env in one LDS bank).  A role touches them
// only when it loads / stores its limb's state, lane = env at a pitch of 54 / 27 floats (2-way / conflict-free).  root_states keeps the
// [row][env] form: the waist and the ball wave read and write it every substep, and env-major it cost the roles more (14 more spilled
// registers, substeps +1.5k cycles) than its flush gained.  Stamped timeline at 4096 envs (profiles/r04_c_TA_chain_stamps.txt): inputs staged
// 5.9k -> 4.1k cycles, the state tiles' flush 4.6k -> 1.9k, workgroup span 67.7k -> 64.6k; the launch itself, same box, 30.3 us before and after
// (tools/gpu_r4_ta_ab.sh) — what the stamps gain the un-stamped build does not show.  Kept for the simpler copies, not for speed.
struct __attribute__((aligned(16))) Shared {
    float dof[kE * 2 * NDOF];         // (q, qd) pairs: element (e, d, k) at e * 54 + 2 d + k
    float act[kE * NDOF];             // raw actions at the start, the reported drive torques at the end: (e, d) at e * 27 + d
    float root[39][kP];               // rows 0..12: the base state (start of the current substep / final), 13..25 table, 26..38 ball ([row][env]: the waist and the ball wave read and write it every substep)
    float torso[18][kP];              // pass 1 hand-off: Rw 9, pw 3, w 3, v 3 of the torso (link 15)
    float sums[5][3][kP];             // final phase: each chain wave's share of the balance sums (pos, vel, norm)
    float paddle[3][kP];              // final phase: paddle position (body 39)
    float pre_vx[kE];                 // the ball's vx before the step (TA:1143)
    union {
        struct {
            float art_leg[2][27][kP]; // pass 2: what a leg adds to the pelvis (A6 B9 D6 pn3 pf3)
            float art_arm[2][27][kP]; // pass 2: what an arm adds to the torso
            float acc_base[6][kP];    // pelvis acceleration (alpha, a)
            float acc_torso[6][kP];   // torso acceleration
            float geo_w[2][kGeoW][kP];   // collision geometry of the pelvis / torso shapes + bound centre, slot = substep & 1
            float geo_ra[2][kGeoRA][kP]; // ... of the forearm / hand shapes and the paddle blade
        } hub;
        float obs[kE * PPENV_TA_NUM_OBS];   // final phase: the observation rows, row-major (flushed as one contiguous block)
    } u;
    int f_torso, f_leg[2], f_arm[2], f_accb, f_acct, f_base, f_geo_w, f_geo_ra, f_ball;   // sequence numbers of the hand-offs (11 words, zeroed together)
    int dead;                         // some wave of this workgroup gave up waiting: nobody stores
};
static_assert(sizeof(Shared) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void publish(int* flag, int value) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef TA_POLL_SLEEP
#define TA_POLL_SLEEP 1      // s_sleep argument between two polls of a hand-off flag (x 64 cycles)
#endif
__device__ __forceinline__ bool await(int* flag, int value) {
    for (int spin = 0; spin < (1 << 22); spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= value) return true;
        __builtin_amdgcn_s_sleep(TA_POLL_SLEEP);
    }
    return false;
}
// A wave whose partner never shows up must not store anything: it marks the launch as failed (the word the host checks) and
// keeps going through the remaining barriers with `dead` set, so that the workgroup still drains.
#define TA_AWAIT(flag, value)                                                                                        \
    do {                                                                                                             \
        if (!dead && !await(flag, value)) {                                                                          \
            dead = true;                                                                                             \
            if (lane == 0) { __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&S.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } \
        }                                                                                                            \
    } while (0)

__device__ __forceinline__ V3 row3(const float (*r)[kP], int k0, int e) { return mk(r[k0][e], r[k0 + 1][e], r[k0 + 2][e]); }
__device__ __forceinline__ void put3(float (*r)[kP], int k0, int e, V3 v) { r[k0][e] = v.x; r[k0 + 1][e] = v.y; r[k0 + 2][e] = v.z; }
__device__ __forceinline__ void put_art(float (*r)[kP], int e, const ArtI& I) {
    const float v[27] = {I.A.xx, I.A.yy, I.A.zz, I.A.xy, I.A.xz, I.A.yz, I.B.m[0], I.B.m[1], I.B.m[2], I.B.m[3], I.B.m[4], I.B.m[5], I.B.m[6], I.B.m[7], I.B.m[8],
                         I.D.xx, I.D.yy, I.D.zz, I.D.xy, I.D.xz, I.D.yz, I.pn.x, I.pn.y, I.pn.z, I.pf.x, I.pf.y, I.pf.z};
#pragma unroll
    for (int t = 0; t < 27; t++) r[t][e] = v[t];
}
__device__ __forceinline__ ArtI get_art(const float (*r)[kP], int e) {
    float v[27];
#pragma unroll
    for (int t = 0; t < 27; t++) v[t] = r[t][e];
    ArtI I = {{v[0], v[1], v[2], v[3], v[4], v[5]}, {{v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14]}},
              {v[15], v[16], v[17], v[18], v[19], v[20]}, mk(v[21], v[22], v[23]), mk(v[24], v[25], v[26])};
    return I;
}
// Keeps the compiler from sinking the computation of I below this point: a bounded wait on a hand-off flag is no obstacle to moving
// register arithmetic across it, and whatever is only consumed after the wait would otherwise be scheduled there — on the critical
// path, instead of inside the time the wave spends waiting anyway.
__device__ __forceinline__ void pin_art(ArtI& I) {
    asm volatile("" : "+v"(I.A.xx), "+v"(I.A.yy), "+v"(I.A.zz), "+v"(I.A.xy), "+v"(I.A.xz), "+v"(I.A.yz));
    asm volatile("" : "+v"(I.B.m[0]), "+v"(I.B.m[1]), "+v"(I.B.m[2]), "+v"(I.B.m[3]), "+v"(I.B.m[4]), "+v"(I.B.m[5]), "+v"(I.B.m[6]), "+v"(I.B.m[7]), "+v"(I.B.m[8]));
    asm volatile("" : "+v"(I.D.xx), "+v"(I.D.yy), "+v"(I.D.zz), "+v"(I.D.xy), "+v"(I.D.xz), "+v"(I.D.yz));
    asm volatile("" : "+v"(I.pn.x), "+v"(I.pn.y), "+v"(I.pn.z), "+v"(I.pf.x), "+v"(I.pf.y), "+v"(I.pf.z));
}
__device__ __forceinline__ ArtI art_zero() {
    ArtI I = {{0, 0, 0, 0, 0, 0}, {{0, 0, 0, 0, 0, 0, 0, 0, 0}}, {0, 0, 0, 0, 0, 0}, mk(0, 0, 0), mk(0, 0, 0)};
    return I;
}

// what a link leaves for the later passes (the compiler keeps only the members a link's later code reads)
struct LinkSave { float c, s; V3 w, v; M3 Rw; V3 pw; };
struct Frame { M3 Rw; V3 pw, w, v; };   // pose (world <- link, origin) and link-frame twist

template <int LI>
__device__ __forceinline__ M3 joint_E(float c, float s) {
    constexpr LinkC L = T::link(LI);
    return L.axis == 0 ? joint_rot<0>(L.R0, c, s) : (L.axis == 1 ? joint_rot<1>(L.R0, c, s) : joint_rot<2>(L.R0, c, s));
}
// link_kinematics (ppenv_ta_device.h) for a compile-time link
template <int LI>
__device__ __forceinline__ void kin_step(float q, float qd, Frame& f, float& c, float& s) {
    constexpr LinkC L = T::link(LI);
    sincos_joint(q, s, c);
    const M3 E = joint_E<LI>(c, s);
    const V3 r = ld3(L.r);
    const V3p wv = tmul(E, pk(f.w, f.v + cross(f.w, r)));
    V3 wn = lo(wv), vn = hi(wv);
    if (L.axis == 0) wn.x += qd; else if (L.axis == 1) wn.y += qd; else wn.z += qd;
    f.pw = f.pw + mul(f.Rw, r);
    f.Rw = mul(f.Rw, E);
    f.w = wn; f.v = vn;
}
__device__ __forceinline__ Frame base_frame(const float (*root)[kP], int e) {
    float quat[4] = {root[3][e], root[4][e], root[5][e], root[6][e]};
    Frame f;
    f.Rw = quat_to_m3(quat);
    f.pw = row3(root, 0, e);
    f.w = tmul(f.Rw, row3(root, 10, e));
    f.v = tmul(f.Rw, row3(root, 7, e));
    return f;
}

// the contact-point table of the compiled model in the shape link_dynamics expects
struct CPTable { float p[T::kContacts][3]; };
__device__ __forceinline__ constexpr CPTable cp_table() {
    CPTable t{};
    for (int k = 0; k < T::kContacts; k++)
        for (int x = 0; x < 3; x++) t.p[k][x] = T::cpoint(k).x[x];
    return t;
}

// point fixed in a link: world position and velocity
__device__ __forceinline__ void point_of(const Frame& f, V3 r, V3& p, V3& vel) {
    p = f.pw + mul(f.Rw, r);
    vel = mul(f.Rw, f.v + cross(f.w, r));
}

// ---- domain randomisation (cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml carries the same randomization_params block as the
// 7-dof yamls) -----------------------------------------------------------------------------------------------------------------------------
// A link with this env's scales applied: the literals of the compiled model times three run-time factors (mass, m c and the inertia about the
// origin all scale with the mass, as recomputeInertia does; the drive gains on their own).  DR = false returns the literal link untouched.
template <bool DR>
__device__ __forceinline__ LinkC dr_link(const LinkC& L, float ms, float kps, float kds) {
    if (!DR) return L;
    LinkC R = L;
    R.mass *= ms;
#pragma unroll
    for (int t = 0; t < 3; t++) R.mc[t] *= ms;
#pragma unroll
    for (int t = 0; t < 6; t++) R.Io[t] *= ms;
    R.kp *= kps; R.kd *= kds;
    return R;
}
// the noise draws of this task: index < 512 (27 action draws, then 32 + k for observation value k < 313) folded into dr_gauss's 256-per-step key space
__device__ __forceinline__ float ta_dr_gauss(uint64_t seed, uint32_t gid, uint32_t episode, uint32_t progress, uint32_t index) {
    return dr_gauss(seed, gid, episode, 2u * progress + (index >> 8), index & 255u);
}
// Round 4: the per-env scales live in LDS, not in registers.  A limb's 3 N scales (21 for the seven-link arm) used to stay in VGPRs across the
// whole role — on a kernel that sits at the 256-register limit six waves on four SIMDs impose, that was 245 spilled VGPRs (300 B of scratch per lane)
// in the <true> instantiation.  Now load() parks them in [row][64] tiles like every other per-env quantity of this kernel (each lane reads back only
// what it wrote itself: DS operations of a wave execute in order, no hand-off involved) and link_of() reads the three a link needs where it is used.
struct DrShared { float kp[NDOF][kE], kd[NDOF][kE], ms[NL][kE]; };    // 82 rows x 256 B = 21 KB, DR instantiation only
typedef __attribute__((address_space(3))) DrShared* DrLds;
struct DrKeys { const TAChainArgs* a; int env, n; uint32_t gid, ep, prog; DrLds tab; };   // what a limb's load() needs to fetch its table entries and draw its action noise

// ---- one limb: the three passes over a compile-time chain ---------------------------------------------------------------
template <int FIRST, int N, bool DR = false>
struct Limb {
    float q[N], qd[N], target[N], force[N];
    DrLds tab;                                                  // DR: this env's scales of the limb's links, column `lane` of the LDS tables
    int lane;
    LinkSave sv[N];
    JointOut jo[N];
    template <int K>
    __device__ __forceinline__ LinkC link_of() const {
        if constexpr (!DR) return T::link(FIRST + K);
        else return dr_link<true>(T::link(FIRST + K), tab->ms[FIRST + K][lane], tab->kp[FIRST - 1 + K][lane], tab->kd[FIRST - 1 + K][lane]);
    }

    // pass 1: kinematics outwards from the parent's frame; GEO(link, frame) sees every link (collision geometry capture)
    template <class GEO>
    __device__ __forceinline__ void pass1(Frame f, GEO&& geo) {
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            float c, s;
            kin_step<FIRST + k>(q[k], qd[k], f, c, s);
            sv[k].c = c; sv[k].s = s; sv[k].w = f.w; sv[k].v = f.v; sv[k].Rw = f.Rw; sv[k].pw = f.pw;
            geo(std::integral_constant<int, FIRST + k>{}, f);
        });
    }
    // pass 2: articulated inertias inwards; `acc` = what the chain's tip receives from outside (zero, or the arms at the torso);
    // returns the chain's contribution to its parent, in the parent's coordinates
    // The links' own dynamics do not depend on what arrives at the tip: a chain that WAITS for its tip's input (the waist, for the arms)
    // computes them first (pass2_own into `own`, then pass2 with OWN = true) instead of after the wait, on the critical path.
    __device__ __forceinline__ void pass2_own(const TAScal& P, ArtI (&own)[N]) {
        constexpr CPTable cp = cp_table();
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const LinkC L = link_of<k>();
            own[k] = link_dynamics(P, L, cp.p, sv[k].Rw, sv[k].pw, sv[k].w, sv[k].v);
            pin_art(own[k]);
        });
    }
    template <bool OWN = false>
    __device__ __forceinline__ ArtI pass2(const TAScal& P, ArtI acc, const ArtI* own = nullptr) {
        constexpr CPTable cp = cp_table();
        static_for<N>([&](auto kc) {
            constexpr int k = N - 1 - decltype(kc)::value;
            const LinkC L = link_of<k>();
            ArtI I = OWN ? own[k] : link_dynamics(P, L, cp.p, sv[k].Rw, sv[k].pw, sv[k].w, sv[k].v);
            add_art(I, acc);
            const M3 E = joint_E<FIRST + k>(sv[k].c, sv[k].s);
            inward_step(P, L, I, sv[k].w, sv[k].v, E, q[k], qd[k], target[k], jo[k]);
            acc = I;
        });
        return acc;
    }
    // pass 3: accelerations outwards from the parent's (in: aw, av of the parent; out: of the chain's last link), joints integrated on the way
    __device__ __forceinline__ void pass3(const TAScal& P, V3& aw, V3& av) {
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const LinkC L = link_of<k>();
            const M3 E = joint_E<FIRST + k>(sv[k].c, sv[k].s);
            outward_step(P, L, E, sv[k].w, sv[k].v, jo[k], aw, av, target[k], q[k], qd[k], force[k]);
        });
    }
    // (the staging leaves the raw actions in S.act: the clamp and the map onto the joint range happen here, where the joint is a
    // compile-time constant — in the staging loop the lanes of a wave look at different dofs and the limits were a table lookup)
    __device__ __forceinline__ void load(const Shared& S, int e, float clip_actions, const DrKeys& dk = DrKeys{}) {
        tab = dk.tab; lane = e;
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr LinkC L = T::link(FIRST + k);
            constexpr int d = FIRST - 1 + k;                          // the dof of link FIRST + k
            q[k] = S.dof[(e) * (2 * NDOF) + 2 * (d)]; qd[k] = S.dof[(e) * (2 * NDOF) + 2 * (d) + 1]; force[k] = 0.f;
            float act = S.act[(e) * NDOF + (d)];
            if constexpr (DR) {                                       // this env's table entries (a NULL table = scale 1); the action noise goes in before the clamp
                const TAChainArgs& a = *dk.a;
                tab->kp[d][e] = a.dr_kp ? a.dr_kp[(size_t)d * dk.n + dk.env] : 1.f;
                tab->kd[d][e] = a.dr_kd ? a.dr_kd[(size_t)d * dk.n + dk.env] : 1.f;
                tab->ms[FIRST + k][e] = a.dr_ms ? a.dr_ms[(size_t)(FIRST + k) * dk.n + dk.env] : 1.f;
                if (a.dr_act_sigma > 0.f) act += a.dr_act_sigma * ta_dr_gauss(a.p.seed, dk.gid, dk.ep, dk.prog, (uint32_t)d);
            }
            target[k] = pd_target(act, L.lo, L.hi, clip_actions);   // VecTask.step clamp + TA:1131, 729-733
        });
    }
    // the final phase only needs the new (q, qd): S.act holds the drive torques by then
    __device__ __forceinline__ void load_state(const Shared& S, int e) {
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            q[k] = S.dof[(e) * (2 * NDOF) + 2 * (FIRST - 1 + k)]; qd[k] = S.dof[(e) * (2 * NDOF) + 2 * (FIRST - 1 + k) + 1]; target[k] = 0.f; force[k] = 0.f;
        });
    }
    __device__ __forceinline__ void store(Shared& S, int e) const {
        static_for<N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            S.dof[(e) * (2 * NDOF) + 2 * (FIRST - 1 + k)] = q[k]; S.dof[(e) * (2 * NDOF) + 2 * (FIRST - 1 + k) + 1] = qd[k]; S.act[(e) * NDOF + (FIRST - 1 + k)] = force[k];
        });
    }
};

// ---- final phase: what a rigid body contributes to the task (ppenv_ta_task.h, one lane per env) ---------------------------
struct TaskCtx {
    const TAChainArgs& a;
    Shared& S;
    int e, env;               // lane, global env index (clamped to a valid env for lanes past the end)
    float hinv[4];
    V3 rootp;
    float pos_acc, vel_acc, norm_acc;
    float irb[tatask::TA_NBAL][6];   // initial position / velocity of the balance bodies THIS wave owns (the other entries are never touched)
};
// issue the loads of the initial body states of link LI's bodies (uncoalesced: 52-byte rows at a 2184-byte stride); they are consumed by body_out
template <int BODY>
__device__ __forceinline__ void prefetch_body(TaskCtx& c) {
    constexpr int jb = tatask::ta_bal_index(BODY);
    if constexpr (jb >= 0) {
        // (a shared block: the address is the same in every lane, so these are scalar loads)
        const float* r = c.a.initial_rb + ((size_t)(c.a.p.initial_rb_shared ? 0 : c.env) * PPENV_NUM_BODIES + BODY) * 13;
        c.irb[jb][0] = r[0]; c.irb[jb][1] = r[1]; c.irb[jb][2] = r[2]; c.irb[jb][3] = r[7]; c.irb[jb][4] = r[8]; c.irb[jb][5] = r[9];
    }
}
template <int FIRST, int N>
__device__ __forceinline__ void prefetch_limb(TaskCtx& c) {
    static_for<N>([&](auto kc) {
        constexpr LinkC L = T::link(FIRST + decltype(kc)::value);
        prefetch_body<L.body>(c);
        static_for<L.fcount>([&](auto fc) { prefetch_body<T::fixed(L.ffirst + decltype(fc)::value).body>(c); });
    });
}
// body `BODY` (Isaac Gym rigid-body index) at world position p, linear velocity v; R / w only feed the optional rigid_body_states row
template <int BODY>
__device__ __forceinline__ void body_out(TaskCtx& c, const M3& R, V3 p, V3 v, V3 wworld) {
    constexpr int jb = tatask::ta_bal_index(BODY), jo = tatask::ta_obs_index(BODY);
    float* o = &c.S.u.obs[c.e * PPENV_TA_NUM_OBS];
    if (jb >= 0) {                                                       // compute_imitation_reward TA:1313-1418, compute_imitation_observations TA:1891-1927
        const float* r = c.irb[jb];                                      // prefetched (prefetch_limb)
        const float dp0 = r[0] - p.x, dp1 = r[1] - p.y, dp2 = r[2] - p.z, dv0 = r[3] - v.x, dv1 = r[4] - v.y, dv2 = r[5] - v.z;
        c.pos_acc += (dp0 * dp0 + dp1 * dp1 + dp2 * dp2) / 3.0f;
        c.vel_acc += (dv0 * dv0 + dv1 * dv1 + dv2 * dv2) / 3.0f;
        c.norm_acc += sqrtf(dp0 * dp0 + dp1 * dp1 + dp2 * dp2);
        const V3 t = heading_rotate(c.hinv, mk(dp0, dp1, dp2)), tv = heading_rotate(c.hinv, mk(dv0, dv1, dv2));
        o[121 + 3 * jb] = t.x * 10.f; o[122 + 3 * jb] = t.y * 10.f; o[123 + 3 * jb] = t.z * 10.f;
        o[121 + 3 * tatask::TA_NBAL + 3 * jb] = tv.x; o[122 + 3 * tatask::TA_NBAL + 3 * jb] = tv.y; o[123 + 3 * tatask::TA_NBAL + 3 * jb] = tv.z;
    }
    if (jo >= 0) {                                                       // compute_humanoid_observations TA:1849-1888
        const V3 lp = heading_rotate(c.hinv, p - c.rootp), lv = heading_rotate(c.hinv, v);
        o[3 * jo] = lp.x; o[3 * jo + 1] = lp.y; o[3 * jo + 2] = lp.z;
        o[30 + 3 * jo] = lv.x; o[30 + 3 * jo + 1] = lv.y; o[30 + 3 * jo + 2] = lv.z;
    }
    if (BODY == 39) put3(c.S.paddle, 0, c.e, p);
    if (c.a.rb_states) {                                                 // on request only: the [N,42,13] row (uncoalesced, 52 bytes per lane)
        float* row = c.a.rb_states + ((size_t)c.env * PPENV_NUM_BODIES + BODY) * 13;
        float qt[4];
        rot_to_quat(R, qt);
        row[0] = p.x; row[1] = p.y; row[2] = p.z; row[3] = qt[0]; row[4] = qt[1]; row[5] = qt[2]; row[6] = qt[3];
        row[7] = v.x; row[8] = v.y; row[9] = v.z; row[10] = wworld.x; row[11] = wworld.y; row[12] = wworld.z;
    }
}
// the rigid body of link LI and the bodies welded to it
template <int LI>
__device__ __forceinline__ void link_out(TaskCtx& c, const Frame& f) {
    constexpr LinkC L = T::link(LI);
    const V3p la = mul(f.Rw, pk(f.v, f.w));
    body_out<L.body>(c, f.Rw, f.pw, lo(la), hi(la));
    static_for<L.fcount>([&](auto kc) {
        constexpr FixedC F = T::fixed(L.ffirst + decltype(kc)::value);
        V3 p, vel;
        point_of(f, ld3(F.xyz), p, vel);
        body_out<F.body>(c, mul(f.Rw, ldm(F.rot)), p, vel, hi(la));
    });
}
// final-state kinematics of a limb and what its bodies contribute; returns the frame of its last link
template <int FIRST, int N>
__device__ __forceinline__ Frame limb_frames(const Limb<FIRST, N>& lb, Frame f, Frame (&out)[N]) {
    static_for<N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        float cs, sn;
        kin_step<FIRST + k>(lb.q[k], lb.qd[k], f, cs, sn);
        out[k] = f;
    });
    return f;
}
template <int FIRST, int N>
__device__ __forceinline__ void limb_out(TaskCtx& c, const Limb<FIRST, N>& lb, Frame f) {
    static_for<N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        float cs, sn;
        kin_step<FIRST + k>(lb.q[k], lb.qd[k], f, cs, sn);
        link_out<FIRST + k>(c, f);
    });
}

// collision geometry of the shapes riding on link LI -> rows of a hand-off slot (a va b vb per shape in the order of ArmGeom; paddle; bound)
template <int LI, int NROWS>
__device__ __forceinline__ void geo_capture(float (*slot)[kP], int e, const Frame& f, int& row) {
    constexpr LinkC L = T::link(LI);
    static_for<T::kShapes>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr ((L.geo_mask >> s) & 1) {
            V3 pa, va, pb, vb;
            point_of(f, ld3(T::shape_a(s).x), pa, va);
            point_of(f, ld3(T::shape_b(s).x), pb, vb);
            put3(slot, row, e, pa); put3(slot, row + 3, e, va); put3(slot, row + 6, e, pb); put3(slot, row + 9, e, vb);
            row += 12;
        }
    });
    if constexpr ((L.geo_mask >> 6) & 1) {
        V3 pc, vpc;
        point_of(f, ld3(T::paddle_center().x), pc, vpc);
        const V3 pn = mul(f.Rw, ld3(T::paddle_normal().x));
        const V3 pnd = cross(mul(f.Rw, f.w), pn);
        put3(slot, row, e, pc); put3(slot, row + 3, e, pn); put3(slot, row + 6, e, vpc); put3(slot, row + 9, e, pnd);
        row += 12;
    }
    if constexpr ((L.geo_mask >> 7) & 1) {
        V3 bc, d;
        point_of(f, ld3(T::bound_center().x), bc, d);
        put3(slot, row, e, bc);
        row += 3;
    }
}
// which rows of the two slots hold shape s (the order geo_capture wrote them in: links ascending, shapes ascending within a link)
struct GeoMap { int slot[T::kShapes], row[T::kShapes], paddle_row, bound_row, rows_w, rows_ra; };
constexpr GeoMap geo_map() {
    GeoMap m{};
    int rw = 0, rr = 0;
    const int wl[4] = {0, 13, 14, 15}, rl[5] = {23, 24, 25, 26, 27};
    for (int i = 0; i < 4; i++) {
        for (int s = 0; s < T::kShapes; s++)
            if ((T::link(wl[i]).geo_mask >> s) & 1) { m.slot[s] = 0; m.row[s] = rw; rw += 12; }
        if ((T::link(wl[i]).geo_mask >> 7) & 1) { m.bound_row = rw; rw += 3; }
    }
    for (int i = 0; i < 5; i++) {
        for (int s = 0; s < T::kShapes; s++)
            if ((T::link(rl[i]).geo_mask >> s) & 1) { m.slot[s] = 1; m.row[s] = rr; rr += 12; }
        if ((T::link(rl[i]).geo_mask >> 6) & 1) { m.paddle_row = rr; rr += 12; }
    }
    m.rows_w = rw; m.rows_ra = rr;
    return m;
}
static_assert(geo_map().rows_w == kGeoW && geo_map().rows_ra == kGeoRA, "collision geometry hand-off sizes");
constexpr bool geo_owners_ok() {   // every shape, the paddle and the bound centre ride on a link of the waist or the right-arm wave
    int mask = 0;
    const int ls[9] = {0, 13, 14, 15, 23, 24, 25, 26, 27};
    for (int i = 0; i < 9; i++) mask |= T::link(ls[i]).geo_mask;
    for (int i = 0; i < NL; i++) {
        bool mine = false;
        for (int j = 0; j < 9; j++) mine = mine || ls[j] == i;
        if (!mine && T::link(i).geo_mask != 0) return false;
    }
    return mask == ((1 << T::kShapes) - 1 | (1 << 6) | (1 << 7));
}
static_assert(geo_owners_ok(), "collision shapes must ride on the pelvis, the waist chain or the right arm");

// How the full-workgroup flush leaves the CU (profiling builds: -DTA_FLUSH_MODE=0 plain stores; 1 = non-temporal, the default)
#ifndef TA_FLUSH_MODE
#define TA_FLUSH_MODE 1
#endif
template <class V>
__device__ __forceinline__ void flush_store(V v, V* p) {
#if TA_FLUSH_MODE == 1
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// ---- the kernel -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ Frame torso_frame(const Shared& S, int e) {
    Frame f;
#pragma unroll
    for (int k = 0; k < 9; k++) f.Rw.m[k] = S.torso[k][e];
    f.pw = row3(S.torso, 9, e); f.w = row3(S.torso, 12, e); f.v = row3(S.torso, 15, e);
    return f;
}
__device__ __forceinline__ void put_torso(Shared& S, int e, const Frame& f) {
#pragma unroll
    for (int k = 0; k < 9; k++) S.torso[k][e] = f.Rw.m[k];
    put3(S.torso, 9, e, f.pw); put3(S.torso, 12, e, f.w); put3(S.torso, 15, e, f.v);
}

// DR: the table-reading instantiation (domain randomisation on); DR = false is the kernel every reference yaml runs (randomize: False)
template <bool DR>
__global__ __launch_bounds__(kWaves * 64) void ta_chain_kernel(const TAScal P, const TAChainArgs a) {
    __shared__ Shared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e0 = blockIdx.x * kEPW;
    const int n = a.p.num_envs;
    const int nvalid = min(kEPW, n - e0);
    const int e = lane;
    const int env = e0 + min(e, nvalid - 1);      // lanes past the workgroup's last env compute on a copy of it and store nothing (or the same values to its rows)
    const bool live = e < nvalid;
    bool dead = false;                            // a hand-off timed out: finish without storing (TA_AWAIT)
    const int substeps = P.substeps;

    CH_STAMP(0);
    if (tid < 12) (&S.f_torso)[tid] = 0;          // the eleven sequence numbers and `dead`
    // inputs -> LDS, coalesced: the workgroup's blocks of dof_states [N,27,2], actions [N,27], root_states [N,3,13].  A full
    // workgroup moves them as float4 with constant trip counts (every load of the three tensors is in flight before the first LDS
    // store); a ragged last one element by element.
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (nvalid == kE) {
        constexpr int kThreads = kWaves * 64;
        constexpr int nD = kE * 2 * NDOF / 4, nA = kE * NDOF / 4, nR = kE * 39 / 4;
        static_assert(kE * 2 * NDOF % 4 == 0 && kE * NDOF % 4 == 0 && kE * 39 % 4 == 0, "float4 staging");
        constexpr int tD = (nD + kThreads - 1) / kThreads, tA = (nA + kThreads - 1) / kThreads, tR = (nR + kThreads - 1) / kThreads;
        f4v vd[tD], va[tA], vr[tR];
        const f4v* sd = reinterpret_cast<const f4v*>(a.dof_states + (size_t)e0 * 2 * NDOF);
        const f4v* sa = reinterpret_cast<const f4v*>(a.actions + (size_t)e0 * NDOF);
        const f4v* sr = reinterpret_cast<const f4v*>(a.root_states + (size_t)e0 * 39);
#pragma unroll
        for (int it = 0; it < tD; it++) { const int i = it * kThreads + tid; vd[it] = i < nD ? sd[i] : f4v{0, 0, 0, 0}; }
#pragma unroll
        for (int it = 0; it < tA; it++) { const int i = it * kThreads + tid; va[it] = i < nA ? sa[i] : f4v{0, 0, 0, 0}; }
#pragma unroll
        for (int it = 0; it < tR; it++) { const int i = it * kThreads + tid; vr[it] = i < nR ? sr[i] : f4v{0, 0, 0, 0}; }
        // the LDS images have the tensors' own layout: flat 16-byte copies
#pragma unroll
        for (int it = 0; it < tD; it++) { const int i = it * kThreads + tid; if (i < nD) reinterpret_cast<f4v*>(S.dof)[i] = vd[it]; }
#pragma unroll
        for (int it = 0; it < tA; it++) { const int i = it * kThreads + tid; if (i < nA) reinterpret_cast<f4v*>(S.act)[i] = va[it]; }   // raw: Limb::load maps them to PD targets
#pragma unroll
        for (int it = 0; it < tR; it++) {
            const int i = it * kThreads + tid;
            if (i < nR) {
                int ee = (4 * i) / 39, c = 4 * i - ee * 39;
#pragma unroll
                for (int k = 0; k < 4; k++) { S.root[c][ee] = vr[it][k]; if (++c == 39) { c = 0; ee++; } }
            }
        }
    } else {
        for (int t = tid; t < nvalid * 2 * NDOF; t += kWaves * 64) S.dof[t] = a.dof_states[(size_t)e0 * 2 * NDOF + t];
        for (int t = tid; t < nvalid * NDOF; t += kWaves * 64) S.act[t] = a.actions[(size_t)e0 * NDOF + t];
        for (int t = tid; t < nvalid * 39; t += kWaves * 64) { const int ee = t / 39, c = t - ee * 39; S.root[c][ee] = a.root_states[(size_t)e0 * 39 + t]; }
    }
    __syncthreads();
    if (!live) {   // a ragged last workgroup: give the idle lanes a valid state to chew on (copies of its last env)
        const int src = nvalid - 1;
        if (wave == W_WAIST) {
            for (int d = 0; d < NDOF; d++) { S.dof[(e) * (2 * NDOF) + 2 * (d)] = S.dof[(src) * (2 * NDOF) + 2 * (d)]; S.dof[(e) * (2 * NDOF) + 2 * (d) + 1] = S.dof[(src) * (2 * NDOF) + 2 * (d) + 1]; S.act[(e) * NDOF + (d)] = S.act[(src) * NDOF + (d)]; }
            for (int k = 0; k < 39; k++) S.root[k][e] = S.root[k][src];
        }
    }
    __syncthreads();

    DrKeys dk{&a, env, n, (uint32_t)(a.p.env_id_offset + env), 0u, 0u, nullptr};
    if constexpr (DR) {
        __shared__ DrShared Sdr;                                       // (only this instantiation carries it: 120 KB + 21 KB of the CU's 160 KB)
        dk.tab = (DrLds)&Sdr;
        dk.ep = a.episode[env]; dk.prog = (uint32_t)a.progress[env];   // the noise keys: episode / progress at the step's start
    }

    CH_STAMP(1);
    // ---- physics: every wave runs its own role, start to finish (its limb's state lives in ITS registers only; the substep loop is
    // inside the role so that no other role's variables are live across it) -----------------------------------------------------
    auto no_geo = [](auto, const Frame&) {};
    auto leg_role = [&](auto limb_tag, const int leg) {
        typename decltype(limb_tag)::type L;
        L.load(S, e, P.clip_actions, dk);
        for (int sub = 0; sub < substeps; sub++) {
            if (sub > 0) TA_AWAIT(&S.f_base, sub);                  // the base state of this substep
            CH_STAMP(2 + 8 * (sub & 1));
            L.pass1(base_frame(S.root, e), no_geo);
            CH_STAMP(3 + 8 * (sub & 1));
            const ArtI up = L.pass2(P, art_zero());
            put_art(S.u.hub.art_leg[leg], e, up);
            publish(&S.f_leg[leg], sub + 1);
            CH_STAMP(4 + 8 * (sub & 1));
            TA_AWAIT(&S.f_accb, sub + 1);
            CH_STAMP(5 + 8 * (sub & 1));
            V3 aw = row3(S.u.hub.acc_base, 0, e), av = row3(S.u.hub.acc_base, 3, e);
            L.pass3(P, aw, av);
            CH_STAMP(6 + 8 * (sub & 1));
        }
        L.store(S, e);
    };
    auto arm_role = [&](auto limb_tag, const int arm, auto with_geo) {
        typename decltype(limb_tag)::type L;
        L.load(S, e, P.clip_actions, dk);
        for (int sub = 0; sub < substeps; sub++) {
            TA_AWAIT(&S.f_torso, sub + 1);
            CH_STAMP(2 + 8 * (sub & 1));
            const Frame ft = torso_frame(S, e);
            if constexpr (decltype(with_geo)::value) {              // the right arm carries collision shapes and the paddle
                if (sub >= 2) TA_AWAIT(&S.f_ball, sub - 1);         // the ball wave is done with this geometry slot
                int grow = 0;
                float (*gslot)[kP] = S.u.hub.geo_ra[sub & 1];
                L.pass1(ft, [&](auto lc, const Frame& f) { geo_capture<decltype(lc)::value, kGeoRA>(gslot, e, f, grow); });
                publish(&S.f_geo_ra, sub + 1);
            } else {
                L.pass1(ft, no_geo);
            }
            CH_STAMP(3 + 8 * (sub & 1));
            const ArtI up = L.pass2(P, art_zero());
            put_art(S.u.hub.art_arm[arm], e, up);
            publish(&S.f_arm[arm], sub + 1);
            CH_STAMP(4 + 8 * (sub & 1));
            TA_AWAIT(&S.f_acct, sub + 1);
            CH_STAMP(5 + 8 * (sub & 1));
            V3 aw = row3(S.u.hub.acc_torso, 0, e), av = row3(S.u.hub.acc_torso, 3, e);
            L.pass3(P, aw, av);
            CH_STAMP(6 + 8 * (sub & 1));
        }
        L.store(S, e);
    };
    auto waist_role = [&]() {   // pelvis + waist: the two hubs, the base solve, the base integration
        Limb<13, 3, DR> WA;
        WA.load(S, e, P.clip_actions, dk);
        float ms0 = 1.f;                                            // the pelvis' mass scale
        if constexpr (DR) ms0 = a.dr_ms ? a.dr_ms[env] : 1.f;
        BaseState base;
        base.p = row3(S.root, 0, e);
        for (int k = 0; k < 4; k++) base.quat[k] = S.root[3 + k][e];
        base.vw = row3(S.root, 7, e);
        base.ww = row3(S.root, 10, e);
        for (int sub = 0; sub < substeps; sub++) {
            if (sub >= 2) TA_AWAIT(&S.f_ball, sub - 1);             // the ball wave is done with this geometry slot
            CH_STAMP(2 + 8 * (sub & 1));
            const Frame f0 = base_frame(S.root, e);                 // (this wave's own write of the previous substep)
            WA.pass1(f0, no_geo);
            {   // the arms wait for this: the torso's pose leaves before anything else is done with the frames
                Frame ft;
                ft.Rw = WA.sv[2].Rw; ft.pw = WA.sv[2].pw; ft.w = WA.sv[2].w; ft.v = WA.sv[2].v;
                put_torso(S, e, ft);
            }
            publish(&S.f_torso, sub + 1);
            {   // collision geometry of the pelvis / torso shapes for the ball wave
                int grow = 0;
                float (*gslot)[kP] = S.u.hub.geo_w[sub & 1];
                geo_capture<0, kGeoW>(gslot, e, f0, grow);
                static_for<3>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    Frame f;
                    f.Rw = WA.sv[k].Rw; f.pw = WA.sv[k].pw; f.w = WA.sv[k].w; f.v = WA.sv[k].v;
                    geo_capture<13 + k, kGeoW>(gslot, e, f, grow);
                });
            }
            publish(&S.f_geo_w, sub + 1);
            constexpr CPTable cp = cp_table();
            const LinkC L0 = dr_link<DR>(T::link(0), ms0, 1.f, 1.f);
            CH_STAMP(3 + 8 * (sub & 1));
            ArtI I0 = link_dynamics(P, L0, cp.p, f0.Rw, f0.pw, f0.w, f0.v);
            pin_art(I0);                                            // the pelvis' own dynamics: before the wait for the arms
            ArtI own[3];
            WA.pass2_own(P, own);                                   // and the waist links' own
            CH_STAMP(7 + 8 * (sub & 1));
            TA_AWAIT(&S.f_arm[0], sub + 1);
            TA_AWAIT(&S.f_arm[1], sub + 1);
            CH_STAMP(4 + 8 * (sub & 1));
            ArtI arms = get_art(S.u.hub.art_arm[0], e);
            add_art(arms, get_art(S.u.hub.art_arm[1], e));
            const ArtI up = WA.template pass2<true>(P, arms, own);
            add_art(I0, up);
            pin_art(I0);                                            // done before the wait for the legs, not after it
            CH_STAMP(8 + 8 * (sub & 1));
            TA_AWAIT(&S.f_leg[0], sub + 1);
            TA_AWAIT(&S.f_leg[1], sub + 1);
            CH_STAMP(9 + 8 * (sub & 1));
            add_art(I0, get_art(S.u.hub.art_leg[0], e));
            add_art(I0, get_art(S.u.hub.art_leg[1], e));
            V3 alpha, acc;
            solve_base_art(I0, alpha, acc);
            put3(S.u.hub.acc_base, 0, e, alpha); put3(S.u.hub.acc_base, 3, e, acc);
            publish(&S.f_accb, sub + 1);
            CH_STAMP(5 + 8 * (sub & 1));
            V3 aw = alpha, av = acc;
            WA.pass3(P, aw, av);                                    // leaves the torso's acceleration in (aw, av)
            put3(S.u.hub.acc_torso, 0, e, aw); put3(S.u.hub.acc_torso, 3, e, av);
            publish(&S.f_acct, sub + 1);
            integrate_base_regs(P, f0.Rw, f0.w, f0.v, base, alpha, acc);
            put3(S.root, 0, e, base.p);
            for (int k = 0; k < 4; k++) S.root[3 + k][e] = base.quat[k];
            put3(S.root, 7, e, base.vw); put3(S.root, 10, e, base.ww);
            publish(&S.f_base, sub + 1);
            CH_STAMP(6 + 8 * (sub & 1));
        }
        WA.store(S, e);
    };
    auto ball_role = [&]() {   // the ball against the humanoid as it is at each substep's start
        Ball ball;
        ball.p = row3(S.root, 26, e);
        for (int k = 0; k < 4; k++) ball.quat[k] = S.root[29 + k][e];
        ball.v = row3(S.root, 33, e);
        ball.w = row3(S.root, 36, e);
        S.pre_vx[e] = ball.v.x;                                                              // TA:1143
        if (live && a.pre_vx) a.pre_vx[e0 + e] = ball.v.x;
        EnvDR bdr;
        if constexpr (DR) { bdr.es = a.dr_es ? a.dr_es[env] : 1.f; bdr.fs = a.dr_fs ? a.dr_fs[env] : 1.f; }
        for (int sub = 0; sub < substeps; sub++) {
            TA_AWAIT(&S.f_geo_w, sub + 1);
            TA_AWAIT(&S.f_geo_ra, sub + 1);
            CH_STAMP(2 + 8 * (sub & 1));
            constexpr GeoMap gm = geo_map();
            ArmGeom<ModelG1TA::kShapes> g[1];
            V3 bound[1];
            static_for<T::kShapes>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const float (*slot)[kP] = gm.slot[s] == 0 ? S.u.hub.geo_w[sub & 1] : S.u.hub.geo_ra[sub & 1];
                g[0].a[s] = row3(slot, gm.row[s], e); g[0].va[s] = row3(slot, gm.row[s] + 3, e);
                g[0].b[s] = row3(slot, gm.row[s] + 6, e); g[0].vb[s] = row3(slot, gm.row[s] + 9, e);
            });
            {
                const float (*slot)[kP] = S.u.hub.geo_ra[sub & 1];
                g[0].pc = row3(slot, gm.paddle_row, e); g[0].pn = row3(slot, gm.paddle_row + 3, e);
                g[0].vpc = row3(slot, gm.paddle_row + 6, e); g[0].pnd = row3(slot, gm.paddle_row + 9, e);
                bound[0] = row3(S.u.hub.geo_w[sub & 1], gm.bound_row, e);
            }
            ball_substep<ModelG1TA, 1, DR>(*a.K, ball, g, bound, &bdr);
            publish(&S.f_ball, sub + 1);
            CH_STAMP(6 + 8 * (sub & 1));
        }
        put3(S.root, 26, e, ball.p);
        for (int k = 0; k < 4; k++) S.root[29 + k][e] = ball.quat[k];
        put3(S.root, 33, e, ball.v); put3(S.root, 36, e, ball.w);
    };
    if (wave == W_LL) leg_role(std::common_type<Limb<1, 6, DR>>{}, 0);
    else if (wave == W_RL) leg_role(std::common_type<Limb<7, 6, DR>>{}, 1);
    else if (wave == W_WAIST) waist_role();
    else if (wave == W_LA) arm_role(std::common_type<Limb<16, 7, DR>>{}, 0, std::false_type{});
    else if (wave == W_RA) arm_role(std::common_type<Limb<23, 5, DR>>{}, 1, std::true_type{});
    else ball_role();
    CH_STAMP(20);
    __syncthreads();   // B1: the hubs are dead from here on (their memory becomes the observation tile)
    CH_STAMP(21);

    // ---- final phase 1: kinematics of the new state; every rigid body's share of the reward sums and of the observation row.
    // Meanwhile the ball wave does the part of the task arithmetic that only needs the dof tiles (ppenv_ta_task.h ta_task_env).
    float* const orow = &S.u.obs[e * PPENV_TA_NUM_OBS];
    const ppenv_ta_params& p = a.p;
    float s22 = 0.f, s5 = 0.f, sv = 0.f, power = 0.f;      // ball wave: carried over B2
    uint32_t f = 0, ep_in = 0, dr_prog0 = 0;
    long long prog = 0;
    if (wave != W_BALL) {
        TaskCtx c{a, S, e, env, {0, 0, 0, 0}, mk(0, 0, 0), 0.f, 0.f, 0.f, {}};
        if (wave == W_LL) prefetch_limb<1, 6>(c);
        else if (wave == W_RL) prefetch_limb<7, 6>(c);
        else if (wave == W_WAIST) { prefetch_limb<0, 1>(c); prefetch_limb<13, 3>(c); }
        else if (wave == W_LA) prefetch_limb<16, 7>(c);
        else prefetch_limb<23, 5>(c);
        {
            const float rq[4] = {S.root[3][e], S.root[4][e], S.root[5][e], S.root[6][e]};
            heading_quat_inv(rq, c.hinv);                        // calc_heading_quat_inv of the (pre-reset) pelvis, TA:1862
            c.rootp = row3(S.root, 0, e);
        }
        if (wave == W_LL) { Limb<1, 6> L; L.load_state(S, e); limb_out(c, L, base_frame(S.root, e)); }
        else if (wave == W_RL) { Limb<7, 6> L; L.load_state(S, e); limb_out(c, L, base_frame(S.root, e)); }
        else if (wave == W_WAIST) {
            Limb<13, 3> WA;
            WA.load_state(S, e);
            const Frame f0 = base_frame(S.root, e);
            Frame fr[3];
            const Frame ft = limb_frames(WA, f0, fr);
            put_torso(S, e, ft);
            publish(&S.f_torso, substeps + 1);                   // the arms can start
            link_out<0>(c, f0);
            link_out<13>(c, fr[0]); link_out<14>(c, fr[1]); link_out<15>(c, fr[2]);
        } else {
            TA_AWAIT(&S.f_torso, substeps + 1);
            const Frame ft = torso_frame(S, e);
            if (wave == W_LA) { Limb<16, 7> L; L.load_state(S, e); limb_out(c, L, ft); } else { Limb<23, 5> L; L.load_state(S, e); limb_out(c, L, ft); }
        }
        S.sums[wave][0][e] = c.pos_acc; S.sums[wave][1][e] = c.vel_acc; S.sums[wave][2][e] = c.norm_acc;
    } else {
        f = a.flags[env]; ep_in = a.episode[env]; prog = a.progress[env] + 1;                  // TA:1146 (loads in flight during the sums below)
        dr_prog0 = (uint32_t)(prog - 1);
#pragma unroll
        for (int d = 0; d < NDOF; d++) {
            const float qv = S.dof[(e) * (2 * NDOF) + 2 * (d)], qdv = S.dof[(e) * (2 * NDOF) + 2 * (d) + 1];
            const float epd = p.init_dof_pos[d] - qv, evd = p.init_dof_vel[d] - qdv;
            if (d < 22) { s22 += epd * epd; sv += evd * evd; } else s5 += epd * epd;
            power += fabsf(S.act[(e) * NDOF + (d)] * qdv);
            orow[60 + d] = qv; orow[60 + NDOF + d] = qdv * 0.1f;                               // TA:1881-1882 (overwritten below if the env resets)
            // the constant tail of the row (TA:1921-1927: the initial dof state)
            orow[121 + 6 * tatask::TA_NBAL + d] = p.init_dof_pos[d]; orow[121 + 6 * tatask::TA_NBAL + NDOF + d] = p.init_dof_vel[d];
        }
        if (a.rb_states && live) {   // on request: the table and ball rows of rigid_body_states
            float* rb = a.rb_states + (size_t)env * PPENV_NUM_BODIES * 13;
            for (int k = 0; k < 13; k++) { rb[40 * 13 + k] = S.root[13 + k][e]; rb[41 * 13 + k] = S.root[26 + k][e]; }
        }
    }
    CH_STAMP(22);
    __syncthreads();   // B2
    CH_STAMP(23);

    // ---- final phase 2a (the five limb waves, beside the ball wave's phase 2): the policy's input, where the row is already final ------
    // ppenv_ta_sim_set_policy_input: clamp((obs - mean) * inv_std, +-clip) as fp16, rows padded with zeros to ld — the arithmetic of
    // ppenv_mlp_prepare_input, which this replaces.  Thread t < 320 owns the column PAIR t % 160 of the rows t / 160 + 2 k: its four statistics are
    // loaded once, a wave-store is 256 contiguous bytes, and the LDS reads of a wave step through the row two floats apart (2-way, not the 8-way of
    // wider pieces).  Columns [60, 121) of a row — the dof entries a resetting env gets rewritten, the ball — are the ball wave's to finish, and
    // observation noise rewrites the whole tile after B3: those pairs wait for B3.  (Round 4 also sent the final part of the fp32 tile from here:
    // the predicate broke the store loop's pipelining, 3.6k + 4.4k cycles against 2.2k for the plain loop after B3 — the tile is not what the tail waits for.)
    typedef float f4v __attribute__((ext_vector_type(4)));
    bool noisy = false;
    if constexpr (DR) noisy = a.dr_obs_sigma > 0.f;
    const bool pin_fast = a.pin_out && nvalid == kE && a.pin_ld == 320;
    const int pcol = 2 * (tid % 160), prow0 = tid / 160;
    const bool pin_early = !noisy && (pcol + 1 < 60 || pcol >= 121);
    float pmu0 = 0.f, pmu1 = 0.f, pis0 = 0.f, pis1 = 0.f;
    auto pin_rows = [&]() {                                       // this thread's 32 rows of its column pair
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const int c0 = pcol < PPENV_TA_NUM_OBS ? pcol : PPENV_TA_NUM_OBS - 1, c1 = pcol + 1 < PPENV_TA_NUM_OBS ? pcol + 1 : PPENV_TA_NUM_OBS - 1;
        _Float16* dst = reinterpret_cast<_Float16*>(a.pin_out) + (size_t)e0 * 320 + pcol;
#pragma unroll 8
        for (int k = 0; k < kE / 2; k++) {
            const int ee = prow0 + 2 * k;
            float g0 = (S.u.obs[ee * PPENV_TA_NUM_OBS + c0] - pmu0) * pis0;
            float g1 = (S.u.obs[ee * PPENV_TA_NUM_OBS + c1] - pmu1) * pis1;
            g0 = fminf(fmaxf(g0, -a.pin_clip), a.pin_clip); g1 = fminf(fmaxf(g1, -a.pin_clip), a.pin_clip);
            const h2 v = {(_Float16)(pcol < PPENV_TA_NUM_OBS ? g0 : 0.f), (_Float16)(pcol + 1 < PPENV_TA_NUM_OBS ? g1 : 0.f)};
            *reinterpret_cast<h2*>(dst + (size_t)ee * 320) = v;
        }
    };
    if (wave != W_BALL && pin_fast && !S.dead) {
        const int c0 = pcol < PPENV_TA_NUM_OBS ? pcol : PPENV_TA_NUM_OBS - 1, c1 = pcol + 1 < PPENV_TA_NUM_OBS ? pcol + 1 : PPENV_TA_NUM_OBS - 1;
        pmu0 = a.pin_mean[c0]; pmu1 = a.pin_mean[c1]; pis0 = a.pin_inv_std[c0]; pis1 = a.pin_inv_std[c1];
        if (pin_early) pin_rows();
    }

    // ---- final phase 2 (ball wave): reward, reset, the rest of the row — ppenv_ta_task.h ta_task_env, one lane per env ---------
    bool any_here = false;
    if (wave == W_BALL) {
        const bool store = live && !S.dead;
        if constexpr (DR) {   // the observation noise's keys — episode and progress at the step's START — in two rows of the torso hand-off, which is dead after B2
            S.torso[0][e] = __uint_as_float(ep_in); S.torso[1][e] = __uint_as_float(dr_prog0);
        }
        float pos_acc = 0.f, vel_acc = 0.f, norm_acc = 0.f;
#pragma unroll
        for (int w = 0; w < 5; w++) { pos_acc += S.sums[w][0][e]; vel_acc += S.sums[w][1][e]; norm_acc += S.sums[w][2][e]; }
        const float r_body_pos = expf(-50.f * (pos_acc / (float)tatask::TA_NBAL));             // TA:1349-1351
        const float r_body_vel = expf(-4.0f * (vel_acc / (float)tatask::TA_NBAL));             // TA:1354-1356
        const float r22 = (0.2f * 50.0f) * expf(-(5.0f * 500.0f) * (s22 / 22.0f));            // TA:1372-1380
        const float r5 = 0.2f * expf(-5.0f * (s5 / 5.0f));                                     // TA:1383-1387
        const float r_dof_vel = expf(-0.05f * (sv / 22.0f));                                   // TA:1393,1401
        float ref_reward = r22 + r5 + 0.2f * r_dof_vel + 0.4f * r_body_pos + 0.2f * r_body_vel;   // TA:1403
        const bool has_fallen = (norm_acc / (float)tatask::TA_NBAL) > (p.is_train ? 0.32f : 1e6f);   // TA:1407-1415
        if (has_fallen) ref_reward = 1.0f * -50.0f;                                            // TA:1416-1417

        const bool paddle_cond = f & PPENV_TA_FLAG_PADDLE_COND, hit_table_calc = f & PPENV_TA_FLAG_HIT_TABLE_CALC;
        const bool die_pen_calc = f & PPENV_TA_FLAG_DIE_PENALTY_CALC, hum_die = f & PPENV_TA_FLAG_HUMANOID_DIE_CALC;
        const V3 paddle = row3(S.paddle, 0, e);
        const V3 ball_p = row3(S.root, 26, e), ball_v = row3(S.root, 33, e);   // the stepped ball (pre-reset)
        const float pre_vx = S.pre_vx[e];
        const float bx = ball_p.x, by = ball_p.y, bz = ball_p.z, vx = ball_v.x;
        const float root_x = S.root[0][e], pelvis_h = S.root[2][e];
        if (has_fallen) f |= PPENV_TA_COUNT_FALL_DOWN;                                         // TA:1525-1529
        const bool x_close = fabsf(bx - paddle.x) < 0.2f;                                      // TA:1544
        const bool first_close = x_close && !paddle_cond;
        const float dy = by - paddle.y, dz = bz - paddle.z;
        const float yz = sqrtf(dy * dy + dz * dz);                                             // TA:1548
        const bool in_circle = yz < 0.15f;
        float pos_reward = 0.f;
        if (first_close && !hum_die) pos_reward = in_circle ? p.hit_paddle_reward : p.miss_paddle_penalty_coefficient * yz;   // TA:1555-1563
        if (first_close && in_circle) f |= PPENV_TA_COUNT_CLOSER;                              // TA:1566-1570
        const bool hit_paddle = pre_vx < 0.f && vx > 1.5f;                                     // TA:1577
        if (hit_paddle) f |= PPENV_TA_COUNT_HIT_PADDLE;
        const float vel_reward = (hit_paddle && !paddle_cond && !hum_die) ? p.alpha_velocity_reward * fabsf(vx) : 0.f;   // TA:1586-1590
        if (x_close) f |= PPENV_TA_FLAG_PADDLE_COND;                                           // TA:1595
        const float time_penalty = (bx > root_x && vx < 0.f) ? -0.01f * (float)prog : 0.f;     // TA:1602-1607
        const bool z_in = bz >= 0.82f && bz <= 0.83f && vx > 0.f;                              // compute_gradient_penalty TA:1245-1301
        const float ddx = bx - 2.5f, ddy = by - 0.0f;
        const float dist = sqrtf(ddx * ddx + ddy * ddy);
        const bool in_range = bx >= 1.9f && bx <= 3.1f && by >= -0.6f && by <= 0.6f;
        if (z_in && in_range) f |= PPENV_TA_COUNT_HIT_TABLE;
        float hit_rp = 0.f;
        if (z_in && !hit_table_calc && !hum_die) hit_rp = in_range ? p.hit_table_reward : p.not_hit_table_penalty * dist;
        if (z_in) f |= PPENV_TA_FLAG_HIT_TABLE_CALC;
        const bool over_net = bx > 1.72f && bx < 1.78f && vx > 0.f;                            // TA:1619-1650
        const bool suitable = bz > 0.96f && bz < 1.25f;
        float over_h = 0.f;
        if (!suitable) over_h = bz > 1.25f ? bz - 1.25f : 0.96f - bz;
        float net_rp = 0.f;
        if (over_net && !hum_die) net_rp = suitable ? p.cross_net_reward : -400.f * over_h;
        if (net_rp > 0.f) f |= PPENV_TA_COUNT_CROSS_NET;                                       // TA:1652-1656
        const float power_reward = -p.power_coefficient * power;                               // TA:1664-1665
        const float die_penalty = (bz < 0.78f && !die_pen_calc && !hum_die) ? p.die_penalty : 0.f;   // TA:1677-1679
        if (bz < 0.78f) f |= PPENV_TA_FLAG_DIE_PENALTY_CALC;                                   // TA:1681
        if (pelvis_h < 0.97f) f |= PPENV_TA_FLAG_HUMANOID_DIE_CALC;                            // TA:1683
        const float reward = 0.f + (((((((pos_reward + power_reward) + vel_reward) + hit_rp) + net_rp) + die_penalty) + time_penalty) + ref_reward);   // TA:1686
        const long long rst = (prog >= (long long)p.max_episode_length - 1) ? 1 : 0;           // TA:1688: time-out only

        // the heading frame of the observation is the PRE-reset pelvis (TA:1150-1160: body states are refreshed before the reset)
        float hinv[4];
        {
            const float rq[4] = {S.root[3][e], S.root[4][e], S.root[5][e], S.root[6][e]};
            heading_quat_inv(rq, hinv);
        }
        const V3 rootp = row3(S.root, 0, e);
        V3 bp = ball_p, bv = ball_v;
        if (rst) {                                                                             // _reset_idx TA:965-1028
            const uint32_t ep = ep_in + 1u;
            float ov[5];
            if (a.reset_override) {
#pragma unroll
                for (int k = 0; k < 5; k++) ov[k] = a.reset_override[(size_t)env * 5 + k];
            } else {
                const uint32_t gid = (uint32_t)(p.env_id_offset + env);
                float u[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint64_t sd = mix64(p.seed + 0x9E3779B97F4A7C15ull * ((uint64_t)gid + 1));
                    uint64_t x = mix64(sd + 0x9E3779B97F4A7C15ull * ((uint64_t)ep * 8 + k + 1));
                    u[k] = (float)(x >> 40) * (1.0f / 16777216.0f);
                }
                ov[0] = p.ball_y_lo + (p.ball_y_hi - p.ball_y_lo) * u[0];                      // draw order TA:976-979: y, z, speed, tilt, tilt_z
                ov[1] = p.ball_z_lo + (p.ball_z_hi - p.ball_z_lo) * u[1];
                const float speed = p.serve_speed_lo + (p.serve_speed_hi - p.serve_speed_lo) * u[2];
                const float ta = p.serve_tilt_lo_deg + (p.serve_tilt_hi_deg - p.serve_tilt_lo_deg) * u[3];
                const float taz = p.serve_tilt_z_lo_deg + (p.serve_tilt_z_hi_deg - p.serve_tilt_z_lo_deg) * u[4];
                const V3 svl = serve_from_draws(PPENV_VARIANT_TN, speed, ta, taz);            // TA:370-375 is TN's form
                ov[2] = svl.x; ov[3] = svl.y; ov[4] = svl.z;
            }
#pragma unroll
            for (int act = 0; act < 3; act++) {
#pragma unroll
                for (int k = 0; k < 7; k++) S.root[act * 13 + k][e] = p.init_root[act][k];
#pragma unroll
                for (int k = 7; k < 13; k++) S.root[act * 13 + k][e] = 0.f;
            }
            S.root[26 + 1][e] = ov[0]; S.root[26 + 2][e] = ov[1]; S.root[26 + 7][e] = ov[2]; S.root[26 + 8][e] = ov[3]; S.root[26 + 9][e] = ov[4];
            bp = mk(p.init_root[2][0], ov[0], ov[1]); bv = mk(ov[2], ov[3], ov[4]);
#pragma unroll
            for (int d = 0; d < NDOF; d++) {
                S.dof[(e) * (2 * NDOF) + 2 * (d)] = p.init_dof_pos[d]; S.dof[(e) * (2 * NDOF) + 2 * (d) + 1] = p.init_dof_vel[d];
                orow[60 + d] = p.init_dof_pos[d]; orow[60 + NDOF + d] = p.init_dof_vel[d] * 0.1f;
            }
            prog = 0;
            f &= ~(PPENV_TA_FLAG_PADDLE_COND | PPENV_TA_FLAG_DIE_PENALTY_CALC | PPENV_TA_FLAG_HUMANOID_DIE_CALC | PPENV_TA_FLAG_HIT_TABLE_CALC);   // TA:1021-1024
            if (store) a.episode[env] = ep;
        }
        {   // compute_pingpong_observations TA:1811-1846 (the ball as it is after a possible reset)
            const V3 lb = heading_rotate(hinv, bp - rootp), lv = heading_rotate(hinv, bv);
            orow[114] = lb.x; orow[115] = lb.y; orow[116] = lb.z; orow[117] = lv.x; orow[118] = lv.y; orow[119] = lv.z;
            orow[120] = lb.y + (lv.y / (-lv.x + 1e-6f)) * lb.x;                                 // TA:1839
        }
        if (store) { a.progress[env] = prog; a.flags[env] = f; a.rew[env] = reward; a.reset[env] = rst; }
        any_here = __ballot(store && rst) != 0ull;
    }
    CH_STAMP(24);
    __syncthreads();   // B3
    CH_STAMP(25);

    // ---- final phase 3: the tiles leave as the workgroup's contiguous blocks ------------------------------------------------------
    if (!S.dead) {
        constexpr int kThreads = kWaves * 64;
        if constexpr (DR) {
            // observation noise (yaml: observations / gaussian / additive): on the finished tile, after the reward has been computed from the clean
            // state and before the row leaves (and before the policy's copy of it is made) — index 32 + k, clear of the 27 action draws
            if (a.dr_obs_sigma > 0.f) {
                // a work item is a PAIR of neighbouring values: indices 32 + 2 j and 32 + 2 j + 1 share their Box-Muller radius and angle (dr_gauss_pair)
                constexpr int kPairs = (PPENV_TA_NUM_OBS + 1) / 2;
                for (int t = tid; t < nvalid * kPairs; t += kThreads) {
                    const int ee = t / kPairs, j = t - ee * kPairs;
                    const uint32_t index = 32u + 2u * (uint32_t)j;                   // ta_dr_gauss's folding of the 512-wide index space, for an even index
                    float gc, gs;
                    dr_gauss_pair(a.p.seed, (uint32_t)(a.p.env_id_offset + e0 + ee), __float_as_uint(S.torso[0][ee]), 2u * __float_as_uint(S.torso[1][ee]) + (index >> 8),
                                  (index & 255u) >> 1, gc, gs);
                    float* o = &S.u.obs[ee * PPENV_TA_NUM_OBS + 2 * j];
                    o[0] += a.dr_obs_sigma * gc;
                    if (2 * j + 1 < PPENV_TA_NUM_OBS) o[1] += a.dr_obs_sigma * gs;
                }
            }
            __syncthreads();       // (S.dead is the same for every thread of the workgroup after B3)
        }
        float* dobs = a.obs + (size_t)e0 * PPENV_TA_NUM_OBS;                                           // 16-byte aligned: e0 is a multiple of 64
        if (nvalid == kE) {
            // constant trip counts: several LDS reads are in flight before the first store
            constexpr int nO = kE * PPENV_TA_NUM_OBS / 4, nD = kE * 2 * NDOF / 4, nF = kE * NDOF / 4, nR = kE * 39 / 4;
            constexpr int tO = (nO + kThreads - 1) / kThreads;
#pragma unroll 7
            for (int it = 0; it < tO; it++) {
                const int i = it * kThreads + tid;
                if (i < nO) flush_store(reinterpret_cast<const f4v*>(S.u.obs)[i], reinterpret_cast<f4v*>(dobs) + i);
            }
            CH_STAMP(27);
            f4v* dd = reinterpret_cast<f4v*>(a.dof_states + (size_t)e0 * 2 * NDOF);
#pragma unroll
            for (int it = 0; it < (nD + kThreads - 1) / kThreads; it++) {
                const int i = it * kThreads + tid;
                if (i < nD) flush_store(reinterpret_cast<const f4v*>(S.dof)[i], dd + i);
            }
            CH_STAMP(28);
            f4v* df = reinterpret_cast<f4v*>(a.dof_force + (size_t)e0 * NDOF);
#pragma unroll
            for (int it = 0; it < (nF + kThreads - 1) / kThreads; it++) {
                const int i = it * kThreads + tid;
                if (i < nF) flush_store(reinterpret_cast<const f4v*>(S.act)[i], df + i);
            }
            CH_STAMP(29);
            f4v* dr = reinterpret_cast<f4v*>(a.root_states + (size_t)e0 * 39);
#pragma unroll
            for (int it = 0; it < (nR + kThreads - 1) / kThreads; it++) {
                const int i = it * kThreads + tid;
                if (i < nR) {
                    f4v v;
                    int ee = (4 * i) / 39, c = 4 * i - ee * 39;
#pragma unroll
                    for (int k = 0; k < 4; k++) { v[k] = S.root[c][ee]; if (++c == 39) { c = 0; ee++; } }
                    flush_store(v, dr + i);
                }
            }
            CH_STAMP(30);
            if (pin_fast) {
                if (wave != W_BALL && !pin_early) pin_rows();                      // the column pairs phase 2a could not send (all of them with observation noise on)
            } else if (a.pin_out) {
                // The policy's input straight from the tile: clamp((obs - mean) * inv_std, +-clip) as fp16, rows
                // padded with zeros to pin_ld — the same arithmetic as ppenv_mlp_prepare_input, which this replaces.  Two columns per
                // lane: 256 contiguous bytes per wave-store; the statistics are L1 hits after the first row.
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const int ppr = a.pin_ld >> 1, total = kE * ppr;
                _Float16* dst = reinterpret_cast<_Float16*>(a.pin_out) + (size_t)e0 * a.pin_ld;
                for (int i = tid; i < total; i += kThreads) {
                    const int ee = i / ppr, c = 2 * (i - ee * ppr);
                    const int c0 = c < PPENV_TA_NUM_OBS ? c : PPENV_TA_NUM_OBS - 1, c1 = c + 1 < PPENV_TA_NUM_OBS ? c + 1 : PPENV_TA_NUM_OBS - 1;
                    float g0 = (S.u.obs[ee * PPENV_TA_NUM_OBS + c0] - a.pin_mean[c0]) * a.pin_inv_std[c0];
                    float g1 = (S.u.obs[ee * PPENV_TA_NUM_OBS + c1] - a.pin_mean[c1]) * a.pin_inv_std[c1];
                    g0 = fminf(fmaxf(g0, -a.pin_clip), a.pin_clip); g1 = fminf(fmaxf(g1, -a.pin_clip), a.pin_clip);
                    const h2 v = {(_Float16)(c < PPENV_TA_NUM_OBS ? g0 : 0.f), (_Float16)(c + 1 < PPENV_TA_NUM_OBS ? g1 : 0.f)};
                    *reinterpret_cast<h2*>(dst + (size_t)ee * a.pin_ld + c) = v;
                }
            }
        } else {
            if (a.pin_out) {   // ragged last block: the same, row by row
                const int ppr = a.pin_ld >> 1;
                _Float16* dst = reinterpret_cast<_Float16*>(a.pin_out) + (size_t)e0 * a.pin_ld;
                for (int i = tid; i < nvalid * ppr; i += kThreads) {
                    const int ee = i / ppr, c = 2 * (i - ee * ppr);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        float g = 0.f;
                        if (c + k < PPENV_TA_NUM_OBS) g = fminf(fmaxf((S.u.obs[ee * PPENV_TA_NUM_OBS + c + k] - a.pin_mean[c + k]) * a.pin_inv_std[c + k], -a.pin_clip), a.pin_clip);
                        dst[(size_t)ee * a.pin_ld + c + k] = (_Float16)g;
                    }
                }
            }
            const int nvec = nvalid * PPENV_TA_NUM_OBS / 4, rem = nvalid * PPENV_TA_NUM_OBS - 4 * nvec;   // a ragged block need not be a multiple of 4
            for (int t = tid; t < nvec; t += kThreads)
                __builtin_nontemporal_store(reinterpret_cast<const f4v*>(S.u.obs)[t], reinterpret_cast<f4v*>(dobs) + t);
            if (tid < rem) dobs[4 * nvec + tid] = S.u.obs[4 * nvec + tid];
            for (int t = tid; t < nvalid * 2 * NDOF; t += kThreads) __builtin_nontemporal_store(S.dof[t], &a.dof_states[(size_t)e0 * 2 * NDOF + t]);
            for (int t = tid; t < nvalid * NDOF; t += kThreads) __builtin_nontemporal_store(S.act[t], &a.dof_force[(size_t)e0 * NDOF + t]);
            for (int t = tid; t < nvalid * 39; t += kThreads) { const int ee = t / 39, c = t - ee * 39; __builtin_nontemporal_store(S.root[c][ee], &a.root_states[(size_t)e0 * 39 + t]); }
        }
    }
    CH_STAMP(26);
    // TA:1162-1166: whenever ANY env resets, the diagnostic count flags of ALL envs are cleared.  The workgroups take tickets in
    // the scratch word (bit 0: somebody reset; the rest: workgroups done); the one that draws the last ticket has seen every other
    // workgroup's flag stores (release before the ticket, acquire after it) and does the clearing.  Last in the kernel: the round
    // trip of the atomic overlaps the other waves' stores.  (A workgroup that died still takes its ticket.)
    if (wave == W_BALL) {
        __threadfence();
        uint32_t old = 0;
        if (lane == 0) {
            if (any_here) __hip_atomic_fetch_or(a.scratch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __hip_atomic_fetch_add(a.scratch, 2u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __shfl(old, 0);
        if ((old >> 1) == gridDim.x - 1) {
            __threadfence();
            if ((old & 1u) || any_here)
                for (int i = lane; i < n; i += 64) __hip_atomic_fetch_and(&a.flags[i], ~PPENV_TA_COUNT_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store(a.scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
        }
    }
}
}  // namespace

#if defined(TA_STAMP)
extern "C" int ppenv_ta_chain_debug_read_stamps(unsigned long long* dst, size_t count) {
    if (hipDeviceSynchronize() != hipSuccess) return PPENV_EHIP;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ta_chain_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : PPENV_EHIP;
}
#endif

namespace pp {
namespace ta {

bool ta_chain_model_matches(const TAConsts& C, char* why, size_t nwhy) {
    auto fail = [&](const char* what, int i) { if (why) snprintf(why, nwhy, "%s %d differs from the compiled model", what, i); return false; };
    for (int i = 0; i < NL; i++) {
        const LinkC L = T::link(i);
        const LinkC& R = C.link[i];
        if (L.parent != R.parent || L.axis != R.axis || L.body != R.body) return fail("link topology", i);
        if (L.cfirst != R.cfirst || L.ccount != R.ccount || L.ffirst != R.ffirst || L.fcount != R.fcount) return fail("link contact / welded-body range", i);
        if (L.geo_mask != R.geo_mask) return fail("link geometry mask", i);
        if (memcmp(L.r, R.r, 12) != 0 || memcmp(L.R0, R.R0, 36) != 0) return fail("link frame", i);
        if (L.mass != R.mass || memcmp(L.mc, R.mc, 12) != 0 || memcmp(L.Io, R.Io, 24) != 0) return fail("link inertia", i);
        if (memcmp(&L.lo, &R.lo, 7 * 4) != 0) return fail("link drive", i);
    }
    if (C.sc.num_contacts != T::kContacts || C.sc.num_shapes != T::kShapes || C.sc.paddle_link != T::kPaddleLink || C.sc.bound_link != T::kBoundLink)
        return fail("contact / shape counts or the paddle / bound link", 0);
    for (int k = 0; k < T::kContacts; k++)
        if (memcmp(T::cpoint(k).x, C.cpoint[k], 12) != 0) return fail("contact point", k);
    for (int f = 0; f < PPENV_TA_NUM_FIXED; f++) {
        const FixedC F = T::fixed(f);
        if (F.body != C.fixed[f].body || F.link != C.fixed[f].link || memcmp(F.xyz, C.fixed[f].xyz, 12) != 0 || memcmp(F.rot, C.fixed[f].rot, 36) != 0) return fail("welded body", f);
    }
    for (int s = 0; s < T::kShapes; s++)
        if (C.sc.shape_link[s] != T::shape_link(s) || memcmp(C.sc.shape_a[s], T::shape_a(s).x, 12) != 0 || memcmp(C.sc.shape_b[s], T::shape_b(s).x, 12) != 0) return fail("collision shape", s);
    if (memcmp(C.sc.paddle_center, T::paddle_center().x, 12) != 0 || memcmp(C.sc.paddle_normal, T::paddle_normal().x, 12) != 0 ||
        memcmp(C.sc.bound_center, T::bound_center().x, 12) != 0)
        return fail("paddle / bound centre", 0);
    return true;
}

int ta_chain_launch(const TAScal& P, const TAChainArgs& a, void* stream) {
    const int n = a.p.num_envs;
    if (a.dr_on()) hipLaunchKernelGGL(ta_chain_kernel<true>, dim3((n + kEPW - 1) / kEPW), dim3(kWaves * 64), 0, (hipStream_t)stream, P, a);
    else hipLaunchKernelGGL(ta_chain_kernel<false>, dim3((n + kEPW - 1) / kEPW), dim3(kWaves * 64), 0, (hipStream_t)stream, P, a);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching the chain-wave 27-dof step failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

}  // namespace ta
}  // namespace pp
