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

void shim_serve_velocity(const ppenv_config* cfg, uint32_t gid, uint32_t episode, float* out) {
    V3 v = serve_velocity(make_step_consts(*cfg), gid, episode);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
}
