/*
 * ppenv_policy.h — C ABI of the policy forward that sits next to the env step in the rollout loop.
 *
 * The reference trains with rl_games' a2c_continuous on separate actor and critic MLPs, units [2048, 1536, 1024, 1024, 512, 512],
 * ELU, mixed_precision, normalize_input (cfg/train/HumanoidPingpongTiltG1PPO.yaml:11-31,50-51).  In a rollout that forward runs
 * once per env step, on the observation rows the step kernel has just written.  This entry is one dense layer of it on the
 * matrix cores (v_mfma_f32_32x32x16_f16: fp16 operands, fp32 accumulation — what autocast does to nn.Linear), with the work that
 * surrounds a layer fused in:
 *   - layer 1 reads obs_buf [M, K] fp32 IN PLACE and applies rl_games' RunningMeanStd in eval mode while staging the tile:
 *     x = clamp((obs - mean) * inv_std, -clip, clip), cast to fp16 (no normalised copy of the observations is written);
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

/* One layer (x batch) in one launch. */
int ppenv_mlp_layer_forward(const ppenv_mlp_layer* layer, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPENV_POLICY_H */
