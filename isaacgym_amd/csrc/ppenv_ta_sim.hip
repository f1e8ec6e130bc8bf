// ppenv_ta_sim.hip — the 27-DoF variant's rigid-body step on gfx950: kernel + C ABI (include/ppenv.h, ppenv_ta_simulate).
//
// Mapping: one lane per env, 16 envs per workgroup.  The 28-link tree does not fit a lane's registers, so each lane keeps its
// link / dof records in its own column of an LDS array [slot][16] (1424 slots -> 89 KB per workgroup; consecutive lanes hit
// consecutive banks).  At BASELINE config 5's size (4096 envs per GPU) that is 256 workgroups, one per CU; the link loops are
// uniform across the wave, so the model tables in global memory are read with scalar loads.  The simulation state is the
// caller's Isaac-Gym-layout tensors (AoS, ~0.7 KB per env in and 2.9 KB out): the step is bound by its ~1e5 dependent fp32
// operations per env, not by these bytes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "ppenv_ta_device.h"

using namespace pp;
using namespace pp::ta;

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
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
            float a = fminf(fmaxf(actions[(size_t)e * NDOF + d], -C.clip_actions), C.clip_actions);
            st(DOF_BASE + d * DOF_STRIDE + G_TARGET) = 0.5f * (L.hi + L.lo) + 0.5f * (L.hi - L.lo) * a;
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
}  // namespace

struct ppenv_ta_sim {
    TAConsts host;
    TAConsts* dev;
    StepConsts K;
    int device;
};

extern "C" {

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
    if (hipGetDevice(&s->device) != hipSuccess || hipMalloc((void**)&s->dev, sizeof(TAConsts)) != hipSuccess) {
        ppenv_set_error("ppenv_ta_sim_create: hipMalloc of the model constants failed");
        delete s;
        return PPENV_EHIP;
    }
    // pageable host source: the copy is staged before the call returns, the struct may be reused by the caller
    if (hipMemcpyAsync(s->dev, &s->host, sizeof(TAConsts), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        (void)hipFree(s->dev);
        delete s;
        ppenv_set_error("ppenv_ta_sim_create: uploading the model constants failed");
        return PPENV_EHIP;
    }
    *out = s;
    return PPENV_OK;
}

void ppenv_ta_sim_destroy(ppenv_ta_sim* s) {
    if (!s) return;
    if (s->dev) (void)hipFree(s->dev);
    delete s;
}

int ppenv_ta_simulate(ppenv_ta_sim* s, int32_t n, const float* actions_dev, float* root_states_dev, float* dof_states_dev, float* rb_states_dev,
                      float* dof_force_dev, float* pre_ball_vx_dev, void* stream) {
    if (!s || n <= 0 || !actions_dev || !root_states_dev || !dof_states_dev || !rb_states_dev || !dof_force_dev || !pre_ball_vx_dev) {
        ppenv_set_error("ppenv_ta_simulate: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    hipLaunchKernelGGL(ta_sim_kernel<true>, dim3((n + kTaLanes - 1) / kTaLanes), dim3(kTaLanes), 0, (hipStream_t)stream, s->dev, s->K, n, actions_dev,
                       root_states_dev, dof_states_dev, rb_states_dev, dof_force_dev, pre_ball_vx_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_sim_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

int ppenv_ta_forward_kinematics(ppenv_ta_sim* s, int32_t n, const float* root_states_dev, const float* dof_states_dev, float* rb_states_dev,
                                void* stream) {
    if (!s || n <= 0 || !root_states_dev || !dof_states_dev || !rb_states_dev) {
        ppenv_set_error("ppenv_ta_forward_kinematics: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    hipLaunchKernelGGL(ta_sim_kernel<false>, dim3((n + kTaLanes - 1) / kTaLanes), dim3(kTaLanes), 0, (hipStream_t)stream, s->dev, s->K, n,
                       (const float*)nullptr, const_cast<float*>(root_states_dev), const_cast<float*>(dof_states_dev), rb_states_dev,
                       (float*)nullptr, (float*)nullptr);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_sim_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

}  // extern "C"
