/*
 * ppenv_policy.h — C ABI of the policy forward that sits next to the env step in the rollout loop.
 *
 * The reference trains with rl_games' a2c_continuous on separate actor and critic MLPs, units [2048, 1536, 1024, 1024, 512, 512],
 * ELU, mixed_precision, normalize_input (cfg/train/HumanoidPingpongTiltG1PPO.yaml:11-31,50-51).  In a rollout that forward runs
 * once per env step, on the observation rows the step kernel has just written.  This entry is one dense layer of it on the
 * matrix cores (v_mfma_f32_32x32x16_f16: fp16 operands, fp32 accumulation — what autocast does to nn.Linear), with the work that
 * surrounds a layer fused in:
 *   - layer 1 can read obs_buf [M, K] fp32 IN PLACE and apply rl_games' RunningMeanStd in eval mode while staging the tile:
 *     x = clamp((obs - mean) * inv_std, -clip, clip), cast to fp16 (in_f32 = 1; no normalised copy of the observations is written) —
 *     or take that as a separate small launch (ppenv_mlp_prepare_input, 2 M K bytes) and run on the faster LDS-DMA tile kernels;
 *   - bias add and ELU run on the accumulators; the activations leave as fp16.
 * Plain C, device pointers, caller's HIP stream, no synchronisation; returns 0 or a negative PPENV_E* code (ppenv.h) with the
 * message in ppenv_last_error().
 */
#ifndef PPENV_POLICY_H
#define PPENV_POLICY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppenv_mlp_layer {
    int32_t m, n, k;          /* out[m, n] = act(in[m, k] * w[n, k]^T + bias[n]) */
    int32_t batch;            /* independent problems in one launch (actor | critic): problem b uses the pointers + b * stride */
    /* input: fp16 activations [m, lda] (in_f32 = 0), or fp32 observations [m, lda] normalised on the fly (in_f32 = 1) */
    const void* in;           int64_t in_stride;  int32_t lda;  int32_t in_f32;
    const float* mean;        /* [k]  (in_f32 only; NULL = no normalisation) */
    const float* inv_std;     /* [k]  1 / sqrt(var + eps) */
    float clip;               /* clamp of the normalised observation (rl_games: 5.0) */
    const uint16_t* w;        int64_t w_stride;   int32_t ldw;   /* fp16 weights, torch.nn.Linear layout [n, k] */
    const uint16_t* bias;     int64_t bias_stride;               /* fp16 [n] */
    int32_t elu;              /* 1: ELU(alpha = 1) on the result, 0: linear (the mu / value heads) */
    void* out;                int64_t out_stride; int32_t ldo;   int32_t out_f32;   /* fp16 (or fp32 for the heads) [m, ldo] */
} ppenv_mlp_layer;

/* One layer (x batch) in one launch.  Tile choice: fp16 inputs with k % 64 == 0 and 16-byte aligned rows (lda, ldw, strides multiples
 * of 8, base pointers 16-byte aligned) and n >= 128 take the LDS-DMA kernels (256 x 256, 128 x 256 or 128 x 128 of `out` per
 * workgroup, the largest that still gives three quarters of the CUs a workgroup); everything else the register-staged ones. */
int ppenv_mlp_layer_forward(const ppenv_mlp_layer* layer, void* stream);

/* The first layer's input as its own small launch, so that layer 1 runs on the LDS-DMA kernels too:
 * out[m, ld_out] (fp16) = clamp((obs[m, k] - mean) * inv_std, -clip, clip) in columns < k, zero in columns k .. ld_out - 1
 * (ld_out: k rounded up to a multiple of 64; layer 1's weight rows are zero-padded to the same length).  mean / inv_std NULL: cast
 * only.  rl_games RunningMeanStd in eval mode, as the in_f32 path of ppenv_mlp_layer_forward applies while staging. */
int ppenv_mlp_prepare_input(const float* obs, int32_t m, int32_t k, int32_t ld_obs, const float* mean, const float* inv_std, float clip,
                            void* out, int32_t ld_out, void* stream);

/* The step between the heads and env.step in a rollout (rl_games a2c_continuous, fixed sigma: action ~ Normal(mu, sigma), its
 * negative log-probability for PPO; the clamp is VecTask.step's clip_actions): one launch instead of five elementwise ones.
 *   raw = mu[i, j] + sigma[j] * g,  g = the counter RNG's standard normal at (seed, counter, i, j)   (ppenv_device.h dr_gauss)
 *   actions[i, j] = clamp(raw, lo, hi)            (lo >= hi: no clamp)        [m, a] contiguous
 *   neglogp[i]    = sum_j (0.5 g^2 + log sigma[j]) + 0.5 a log(2 pi)          (NULL: not wanted)
 * The same (seed, counter) gives the same draws; the caller advances `counter` every step.  a <= 256. */
int ppenv_mlp_sample_actions(const float* mu, int32_t m, int32_t a, int32_t ld_mu, const float* sigma, uint64_t seed, uint64_t counter,
                             float lo, float hi, float* actions, float* neglogp, void* stream);

/* The heads layer and the draw in ONE launch: `heads` must be a layer the skinny heads kernel takes (fp16 input, fp32 output [m, n <= 32],
 * batch 1, k % 16 == 0, 16-byte aligned rows); its first num_actions output columns are mu.  Writes `out` as ppenv_mlp_layer_forward
 * does, and actions / neglogp exactly as ppenv_mlp_sample_actions(out, ...) with the same (seed, counter) would. */
int ppenv_mlp_heads_sample(const ppenv_mlp_layer* heads, int32_t num_actions, const float* sigma, uint64_t seed, uint64_t counter,
                           float lo, float hi, float* actions, float* neglogp, void* stream);

/* Generalised advantage estimation over a horizon-major rollout (rl_games' discount_values, a2c_common.py; `gamma`, `tau` of
 * cfg/train/HumanoidPingpongTiltG1PPO.yaml:58-59): for t = H-1 .. 0
 *   delta = scale * rewards[t] + gamma * values[t+1] * (1 - done[t]) - values[t];   adv[t] = delta + gamma * tau * (1 - done[t]) * adv[t+1]
 * returns[t] = adv[t] + values[t].  rewards [H, N] fp32, values [H+1, N] fp32 with row stride ld_values (values[H] = the bootstrap
 * value of the last observation), dones [H, N] int64 (VecTask's reset_buf), advantages / returns [H, N].  One thread per env. */
int ppenv_gae(const float* rewards, const float* values, int32_t ld_values, int64_t values_step, const int64_t* dones, int32_t horizon, int32_t n,
              float gamma, float tau, float reward_scale, float* advantages, float* returns, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPENV_POLICY_H */
