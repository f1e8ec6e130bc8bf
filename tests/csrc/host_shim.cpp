// host_shim.cpp — TEST INFRASTRUCTURE ONLY.
// Runs the per-env arithmetic of isaacgym_amd/csrc/ppenv_device.h (the code the HIP
// kernels execute) on the CPU, one env at a time, so the `-m "not gpu"` tests can
// compare it with the oracle without a GPU.  Never loaded by the product.
#include <cstring>

#include "../../isaacgym_amd/csrc/ppenv_device.h"

using namespace pp;
using T = ModelG1;

namespace {
struct RowStore {
    float* row;
    inline void operator()(int k, float v) { row[k] = v; }
};
template <int A>
void load_state(int n, int i, const float* dof_pos, const float* dof_vel, const float* dof_force, const float* ball,
                const long long* progress, const uint32_t* flags, const uint32_t* episode, EnvStateT<A>& st) {
    for (int d = 0; d < A * ND; d++) { st.q[d] = dof_pos[(size_t)d * n + i]; st.qd[d] = dof_vel[(size_t)d * n + i]; st.dof_force[d] = dof_force[(size_t)d * n + i]; }
    st.ball.p = mk(ball[0 * (size_t)n + i], ball[1 * (size_t)n + i], ball[2 * (size_t)n + i]);
    for (int k = 0; k < 4; k++) st.ball.quat[k] = ball[(size_t)(3 + k) * n + i];
    st.ball.v = mk(ball[7 * (size_t)n + i], ball[8 * (size_t)n + i], ball[9 * (size_t)n + i]);
    st.ball.w = mk(ball[10 * (size_t)n + i], ball[11 * (size_t)n + i], ball[12 * (size_t)n + i]);
    st.progress = progress[(size_t)i * A];
    for (int a = 0; a < A; a++) st.flags[a] = flags[(size_t)a * n + i];
    st.episode = episode[i];
}
template <int A>
void store_state(int n, int i, float* dof_pos, float* dof_vel, float* dof_force, float* ball, long long* progress,
                 uint32_t* flags, uint32_t* episode, const EnvStateT<A>& st) {
    for (int d = 0; d < A * ND; d++) { dof_pos[(size_t)d * n + i] = st.q[d]; dof_vel[(size_t)d * n + i] = st.qd[d]; dof_force[(size_t)d * n + i] = st.dof_force[d]; }
    const float b[13] = {st.ball.p.x, st.ball.p.y, st.ball.p.z, st.ball.quat[0], st.ball.quat[1], st.ball.quat[2], st.ball.quat[3],
                         st.ball.v.x, st.ball.v.y, st.ball.v.z, st.ball.w.x, st.ball.w.y, st.ball.w.z};
    for (int k = 0; k < 13; k++) ball[(size_t)k * n + i] = b[k];
    for (int a = 0; a < A; a++) { progress[(size_t)i * A + a] = st.progress; flags[(size_t)a * n + i] = st.flags[a]; }
    episode[i] = st.episode;
}

// one fused step over SoA arrays laid out like ppenv_buffers; A agents per env (rows A*e + a of actions / obs / rew / reset)
template <int A>
void step_all(const ppenv_config* cfg, const float* actions, float* dof_pos, float* dof_vel, float* dof_force,
              float* ball, long long* progress, uint32_t* flags, uint32_t* episode, const float* serve_override,
              float* obs, float* rew, long long* reset, float* bodies_out) {
    const int n = cfg->num_envs;
    const StepConsts K = make_step_consts(*cfg);
    for (int i = 0; i < n; i++) {
        EnvStateT<A> st;
        load_state<A>(n, i, dof_pos, dof_vel, dof_force, ball, progress, flags, episode, st);
        BodyState bodies[A * NB];
        float pre_vx;
        simulate_env<T, A>(K, &actions[(size_t)i * A * ND], st, bodies, pre_vx);
        if (bodies_out)
            for (int j = 0; j < A * NB; j++) {
                float* o = &bodies_out[((size_t)i * A * NB + j) * 13];
                o[0] = bodies[j].pos.x; o[1] = bodies[j].pos.y; o[2] = bodies[j].pos.z;
                o[3] = o[4] = o[5] = 0.f; o[6] = 1.f;   // the fused step does not carry orientations / angular velocities
                o[7] = bodies[j].lin.x; o[8] = bodies[j].lin.y; o[9] = bodies[j].lin.z;
                o[10] = o[11] = o[12] = 0.f;
            }
        V3 ov;
        const V3* ovp = nullptr;
        if (serve_override) { ov = mk(serve_override[i], serve_override[(size_t)n + i], serve_override[2 * (size_t)n + i]); ovp = &ov; }
        RowStore rs[A];
        for (int a = 0; a < A; a++) rs[a].row = &obs[((size_t)i * A + a) * PPENV_NUM_OBS];
        float r[A];
        long long rst;
        post_physics_env<A>(K, (uint32_t)(cfg->env_id_offset + i), st, bodies, pre_vx, ovp, r, rst, rs);
        for (int a = 0; a < A; a++) { rew[(size_t)i * A + a] = r[a]; reset[(size_t)i * A + a] = rst; }
        store_state<A>(n, i, dof_pos, dof_vel, dof_force, ball, progress, flags, episode, st);
    }
}
}  // namespace

extern "C" {

int shim_model_matches(const ppenv_config* cfg) { return model_matches<T>(*cfg) ? 1 : 0; }

void shim_step(const ppenv_config* cfg, const float* actions /*[A*N,7]*/, float* dof_pos, float* dof_vel, float* dof_force,
               float* ball, long long* progress, uint32_t* flags, uint32_t* episode, const float* serve_override /*[3][N] or NULL*/,
               float* obs /*[A*N,80]*/, float* rew, long long* reset, float* bodies_out /*[N,A*10,13] or NULL*/) {
    if (cfg->num_humanoids == 2)
        step_all<2>(cfg, actions, dof_pos, dof_vel, dof_force, ball, progress, flags, episode, serve_override, obs, rew, reset, bodies_out);
    else
        step_all<1>(cfg, actions, dof_pos, dof_vel, dof_force, ball, progress, flags, episode, serve_override, obs, rew, reset, bodies_out);
}

// joint accelerations of the kernel's ABA for one env (KAT against the oracle's RNEA + solve)
void shim_arm_qdd(const ppenv_config* cfg, const float* q, const float* qd, const float* tau, const float* arm_eff, float* qdd) {
    const StepConsts K = make_step_consts(*cfg);
    JointSave js[ND];
    ArmGeom<T::kShapes> g;
    GeomVisitor<T> gv(g);
    fk_sweep<T>(K.site[0], q, qd, js, gv);
    aba_solve<T>(K.site[0], js, qd, tau, arm_eff, qdd);
}

void shim_serve_from_draws(int form, int m, const float* draws /*[m,3]*/, float* out /*[m,3]*/) {
    for (int i = 0; i < m; i++) {
        V3 v = serve_from_draws(form, draws[3 * i], draws[3 * i + 1], draws[3 * i + 2]);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}
void shim_pd_targets(int m, int nd, const float* actions /*[m,nd]*/, const float* lo, const float* hi /*[nd]*/, float clip, float* out) {
    for (int i = 0; i < m; i++)
        for (int d = 0; d < nd; d++) out[(size_t)i * nd + d] = pd_target(actions[(size_t)i * nd + d], lo[d], hi[d], clip);
}

void shim_serve_velocity(const ppenv_config* cfg, uint32_t gid, uint32_t episode, float* out) {
    V3 v = serve_velocity(make_step_consts(*cfg), gid, episode);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
}

// ---- 27-DoF variant: the rigid-body step through the kernels' arithmetic (ppenv_ta_device.h)
#include "../../isaacgym_amd/csrc/ppenv_ta_device.h"
namespace {
struct ArrayStore {
    float* buf;
    inline float& operator()(int slot) { return buf[slot]; }
};
}  // namespace
extern "C" int shim_ta_simulate(const ppenv_config* scene, const ppenv_ta_model* model, int n, const float* actions, float* root_states,
                                float* dof_states, float* rb_states, float* dof_force, float* pre_vx) {
    using namespace pp::ta;
    static TAConsts C;
    const char* why;
    if (!make_ta_consts(*scene, *model, C, &why)) return -1;
    const StepConsts K = make_step_consts(*scene);
    static float buf[NUM_SLOTS];
    for (int e = 0; e < n; e++) {
        ArrayStore st{buf};
        float* root = &root_states[(size_t)e * 39];
        float* dofs = &dof_states[(size_t)e * 2 * NDOF];
        for (int d = 0; d < NDOF; d++) {
            const LinkC& L = C.link[d + 1];
            st(DOF_BASE + d * DOF_STRIDE + G_TARGET) = pd_target(actions[(size_t)e * NDOF + d], L.lo, L.hi, C.sc.clip_actions);
            st(DOF_BASE + d * DOF_STRIDE + G_Q) = dofs[2 * d];
            st(DOF_BASE + d * DOF_STRIDE + G_QD) = dofs[2 * d + 1];
            st(DOF_BASE + d * DOF_STRIDE + G_FORCE) = 0.f;
        }
        BaseState b;
        b.p = ld3(&root[0]); for (int k = 0; k < 4; k++) b.quat[k] = root[3 + k];
        b.vw = ld3(&root[7]); b.ww = ld3(&root[10]);
        float* bl = &root[26];
        Ball ball;
        ball.p = ld3(&bl[0]); for (int k = 0; k < 4; k++) ball.quat[k] = bl[3 + k];
        ball.v = ld3(&bl[7]); ball.w = ld3(&bl[10]);
        pre_vx[e] = ball.v.x;
        simulate_env_ta(C, K, st, b, ball);
        root[0] = b.p.x; root[1] = b.p.y; root[2] = b.p.z; for (int k = 0; k < 4; k++) root[3 + k] = b.quat[k];
        root[7] = b.vw.x; root[8] = b.vw.y; root[9] = b.vw.z; root[10] = b.ww.x; root[11] = b.ww.y; root[12] = b.ww.z;
        bl[0] = ball.p.x; bl[1] = ball.p.y; bl[2] = ball.p.z; for (int k = 0; k < 4; k++) bl[3 + k] = ball.quat[k];
        bl[7] = ball.v.x; bl[8] = ball.v.y; bl[9] = ball.v.z; bl[10] = ball.w.x; bl[11] = ball.w.y; bl[12] = ball.w.z;
        for (int d = 0; d < NDOF; d++) {
            dofs[2 * d] = st(DOF_BASE + d * DOF_STRIDE + G_Q);
            dofs[2 * d + 1] = st(DOF_BASE + d * DOF_STRIDE + G_QD);
            dof_force[(size_t)e * NDOF + d] = st(DOF_BASE + d * DOF_STRIDE + G_FORCE);
        }
        pass_kinematics<false>(C, st, b);
        float* rb = &rb_states[(size_t)e * 42 * 13];
        write_body_states(C, st, rb);
        memcpy(&rb[40 * 13], &root[13], 13 * sizeof(float));
        memcpy(&rb[41 * 13], &root[26], 13 * sizeof(float));
    }
    return 0;
}
