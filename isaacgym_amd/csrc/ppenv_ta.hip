// ppenv_ta.hip — 27-DoF variant, tensor-API mode: post_physics_step of
// tasks/humanoid_pingpong_3_actor_all_dof.py ("TA") TA:1145-1192 on caller-supplied simulator tensors.
// One lane per env; every op is fp32 in the reference's operation order (IEEE division / sqrt / expf).
// The per-env arithmetic is ppenv_ta_task.h (shared with the fused step of ppenv_ta_sim.hip).
#undef PP_STAMP   // the phase stamps of diagnostic builds belong to ppenv.hip's kernels
#include <hip/hip_runtime.h>

#include <cstdio>

#include "ppenv_ta_task.h"

using namespace pp;

namespace {

constexpr int kTaBlock = 16;   // one lane per env with strided AoS rows: small workgroups spread the envs over all CUs (4096 envs -> 256 workgroups)
__global__ __launch_bounds__(kTaBlock) void ta_post_physics_kernel(const ppenv_ta_params p, const float* __restrict__ rb_states,
                                                                    const float* __restrict__ initial_rb_states, float* root_states,
                                                                    float* dof_states, const float* __restrict__ dof_force,
                                                                    const float* __restrict__ pre_ball_vx,
                                                                    const float* __restrict__ reset_override, uint32_t* flags,
                                                                    uint32_t* episode, long long* progress, float* obs, float* rew,
                                                                    long long* reset_out, uint32_t* any_reset) {
    const int n = p.num_envs;
    const int i = blockIdx.x * kTaBlock + threadIdx.x;
    if (i >= n) return;
    tatask::ta_task_env<1>(p, i, rb_states + (size_t)i * PPENV_NUM_BODIES * 13, initial_rb_states + (size_t)(p.initial_rb_shared ? 0 : i) * PPENV_NUM_BODIES * 13,
                        root_states + (size_t)i * PPENV_NUM_ACTORS * 13, dof_states + (size_t)i * tatask::TA_ND * 2, dof_force + (size_t)i * tatask::TA_ND,
                        pre_ball_vx[i], reset_override ? reset_override + (size_t)i * 5 : nullptr, &flags[i], &episode[i], &progress[i],
                        obs + (size_t)i * PPENV_TA_NUM_OBS, &rew[i], &reset_out[i], any_reset, 0, true);
}

// TA:1162-1166: whenever ANY env resets, the diagnostic count flags of ALL envs are cleared
// One workgroup: every thread reads the word before the barrier, then it is zeroed for the next step — the step needs no memset
// launch (5.5 us of a 133 us step) as long as the word starts at zero.
__global__ __launch_bounds__(1024) void ta_clear_counts_kernel(int n, uint32_t* flags, uint32_t* any_reset) {
    const bool any = *any_reset != 0u;
    __syncthreads();
    if (threadIdx.x == 0) *any_reset = 0u;
    if (any)
        for (int i = threadIdx.x; i < n; i += 1024) flags[i] &= ~PPENV_TA_COUNT_MASK;
}

// ---- 4-actor variant: compute_humanoid1_pingpong_reward (== TT's, T4:1113-1278) and its mirror
// compute_humanoid2_pingpong_reward T4:1280-1439, both sides in one launch
__global__ __launch_bounds__(kTaBlock) void t4_rewards_kernel(const ppenv_t4_params p, const float* __restrict__ rb_states,
                                                               const float* __restrict__ root_states, const float* __restrict__ dof_states,
                                                               const float* __restrict__ dof_force, const float* __restrict__ pre_ball_vx,
                                                               const long long* __restrict__ progress, const uint32_t* __restrict__ flags1_in,
                                                               const uint32_t* __restrict__ flags2_in, uint32_t* flags1, uint32_t* flags2,
                                                               float* rew1, float* rew2, long long* reset1, long long* reset2) {
    const int i = blockIdx.x * kTaBlock + threadIdx.x;
    if (i >= p.num_envs) return;
    const float* rb = rb_states + (size_t)i * PPENV_T4_NUM_BODIES * 13;
    const float* root = root_states + (size_t)i * PPENV_T4_NUM_ACTORS * 13;
    const float* ball = root + 3 * 13;
    float power = 0.f;
#pragma unroll
    for (int d = 0; d < PPENV_T4_NUM_DOF; d++)   // the class hands the whole 14-dof tensors to the reward (T4:746-747)
        power += fabsf(dof_force[(size_t)i * PPENV_T4_NUM_DOF + d] * dof_states[((size_t)i * PPENV_T4_NUM_DOF + d) * 2 + 1]);
    const V3 bp = mk(ball[0], ball[1], ball[2]);
    // side 1 = TT's function verbatim (T4:1113-1278 == TT:1105-1270): the shared compute_reward with TT semantics
    RewardConsts c;
    c.variant = PPENV_VARIANT_TT;
    c.max_episode_length = p.max_episode_length;
    c.alpha = p.alpha_velocity_reward; c.power_coefficient = p.power_coefficient; c.penalty = p.penalty;
    c.hit_table_reward = p.hit_table_reward; c.not_hit_table_penalty = p.not_hit_table_penalty;
    RewardIn in;
    in.humanoid_x = root[0];
    in.paddle = mk(rb[39 * 13], rb[39 * 13 + 1], rb[39 * 13 + 2]);
    in.pre_vx = pre_ball_vx[i];
    in.bp = bp;
    in.vx = ball[7];
    in.power = power;
    in.progress = progress[i];
    uint32_t f1 = flags1_in[i], f2 = flags2_in[i];
    long long r1, r2;
    rew1[i] = compute_reward(c, in, f1, r1);
    RewardIn in2 = in;
    in2.humanoid_x = root[13];
    in2.paddle = mk(rb[79 * 13], rb[79 * 13 + 1], rb[79 * 13 + 2]);
    rew2[i] = compute_reward_side2(c, in2, f2, r2);
    flags1[i] = f1; flags2[i] = f2; reset1[i] = r1; reset2[i] = r2;
}

}  // namespace

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
// The stateless entries have no handle to remember a device: they launch on the device that owns their output tensor, whatever the
// caller's current device is (a rank that never called hipSetDevice would otherwise launch on device 0 with another device's pointers).
int use_device_of(const void* dev_ptr) {
    hipPointerAttribute_t at;
    int cur = -1;
    if (hipPointerGetAttributes(&at, dev_ptr) != hipSuccess || hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        ppenv_set_error("could not determine the device of the output tensor (is it a device pointer?)");
        return PPENV_EINVAL;
    }
    if (at.device != cur && hipSetDevice(at.device) != hipSuccess) { ppenv_set_error("hipSetDevice to the output tensor's device failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
}  // namespace

// TA:1162-1166 as a launch of its own (also used by the fused step of ppenv_ta_sim.hip); not part of the public ABI
int ppenv_ta_clear_counts(int n, uint32_t* flags_dev, uint32_t* any_reset_dev, void* stream) {
    hipLaunchKernelGGL(ta_clear_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, flags_dev, any_reset_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching ta_clear_counts_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_t4_rewards(const ppenv_t4_params* params, const float* rb_states_dev, const float* root_states_dev,
                                const float* dof_states_dev, const float* dof_force_dev, const float* pre_ball_vx_dev,
                                const int64_t* progress_dev, const uint32_t* flags1_in_dev, const uint32_t* flags2_in_dev,
                                uint32_t* flags1_dev, uint32_t* flags2_dev, float* rew1_dev, float* rew2_dev, int64_t* reset1_dev,
                                int64_t* reset2_dev, void* stream) {
    if (!params || params->num_envs <= 0 || !rb_states_dev || !root_states_dev || !dof_states_dev || !dof_force_dev || !pre_ball_vx_dev ||
        !progress_dev || !flags1_in_dev || !flags2_in_dev || !flags1_dev || !flags2_dev || !rew1_dev || !rew2_dev || !reset1_dev || !reset2_dev) {
        ppenv_set_error("ppenv_t4_rewards: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    const int n = params->num_envs;
    if (int rc = use_device_of(rew1_dev)) return rc;
    hipLaunchKernelGGL(t4_rewards_kernel, dim3((n + kTaBlock - 1) / kTaBlock), dim3(kTaBlock), 0, (hipStream_t)stream, *params, rb_states_dev,
                       root_states_dev, dof_states_dev, dof_force_dev, pre_ball_vx_dev, (const long long*)progress_dev, flags1_in_dev, flags2_in_dev, flags1_dev, flags2_dev,
                       rew1_dev, rew2_dev, (long long*)reset1_dev, (long long*)reset2_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching t4_rewards_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_ta_post_physics_step(const ppenv_ta_params* params, const float* rb_states_dev, const float* initial_rb_states_dev,
                                          float* root_states_dev, float* dof_states_dev, const float* dof_force_dev,
                                          const float* pre_ball_vx_dev, const float* reset_override_dev, uint32_t* flags_dev,
                                          uint32_t* episode_dev, int64_t* progress_dev, float* obs_dev, float* rew_dev, int64_t* reset_dev,
                                          uint32_t* scratch_any_reset_dev, void* stream) {
    if (!params || params->num_envs <= 0 || !rb_states_dev || !initial_rb_states_dev || !root_states_dev || !dof_states_dev ||
        !dof_force_dev || !pre_ball_vx_dev || !flags_dev || !episode_dev || !progress_dev || !obs_dev || !rew_dev || !reset_dev ||
        !scratch_any_reset_dev) {
        ppenv_set_error("ppenv_ta_post_physics_step: NULL argument or num_envs <= 0");
        return PPENV_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const int n = params->num_envs;
    if (int rc = use_device_of(obs_dev)) return rc;
    hipLaunchKernelGGL(ta_post_physics_kernel, dim3((n + kTaBlock - 1) / kTaBlock), dim3(kTaBlock), 0, s, *params, rb_states_dev,
                       initial_rb_states_dev, root_states_dev, dof_states_dev, dof_force_dev, pre_ball_vx_dev, reset_override_dev, flags_dev,
                       episode_dev, (long long*)progress_dev, obs_dev, rew_dev, (long long*)reset_dev, scratch_any_reset_dev);
    hipLaunchKernelGGL(ta_clear_counts_kernel, dim3(1), dim3(1024), 0, s, n, flags_dev, scratch_any_reset_dev);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching the TA post-physics kernels failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
