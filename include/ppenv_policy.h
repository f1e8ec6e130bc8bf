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

/* The same launch with its grid sized for `cus` of the 256 CUs (0 = the whole chip, what ppenv_mlp_layer_forward does): the tile is
 * chosen so that the layer is about one workgroup per CU of that share.  For env GROUPS stepped on separate streams (the reference's
 * rollout is policy -> env -> policy per env, tasks/humanoid_pingpong_3_actor_all_dof.py:965-1028 under rl_games' play_steps; groups of
 * envs are independent): with cus = 128 two groups' layers run side by side — one group's loads and stores under the other's MFMAs,
 * and the 64-CU env step of one under the other's forward — instead of each launch claiming every CU in turn.  Same results as
 * ppenv_mlp_layer_forward bit for bit (the tile choice does not change the summation order along k). */
int ppenv_mlp_layer_forward_share(const ppenv_mlp_layer* layer, int32_t cus, void* stream);

/* 2 .. 4 CONSECUTIVE hidden layers in one launch: layers[i + 1] reads exactly what layers[i] writes (in == out, k == n, lda == ldo,
 * in_stride == out_stride), all with the same m and batch, all qualifying for the LDS-DMA tiles (see ppenv_mlp_layer_forward).  The
 * layers' 128-row tiles are drawn by ticket by one persistent workgroup per CU; a tile starts when the tiles of the layer below that cover
 * its 128 rows are done (device-scope release / acquire on a counter per layer, problem and row panel) — no launch boundary, no
 * device-wide barrier, no assumption about workgroup order or co-residency; every wait is bounded (a timeout sets the word
 * ppenv_mlp_chain_status reads and the launch still ends).  Same results as the per-layer launches bit for bit.
 * workspace: ppenv_mlp_chain_workspace_bytes(m, batch, count) bytes of device memory, zeroed ONCE by the caller before the first use (each
 * launch leaves its counters zeroed again) and not shared between launches that may run concurrently.  The reference's rollout has
 * eight dependent launches per policy forward (cfg/train/HumanoidPingpongTiltG1PPO.yaml:25-31: six hidden layers, heads, input
 * normalisation); at 4096 rows each launch boundary costs about as much as a narrow layer's arithmetic. */
size_t ppenv_mlp_chain_workspace_bytes(int32_t m, int32_t batch, int32_t count);
int ppenv_mlp_chain_forward(const ppenv_mlp_layer* layers, int32_t count, void* workspace, void* stream);
/* 0, or 1 after a wait inside ppenv_mlp_chain_forward timed out (sticky; synchronises with the device) */
int ppenv_mlp_chain_status(const void* workspace);

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

/* ---------------------------------------------------------------------------------------------------------------------------------
 * The learner's side of the same network (second half): rl_games' a2c agent runs mini_epochs x minibatches of
 * forward + backward on the horizon's observations (cfg/train/HumanoidPingpongTiltG1PPO.yaml:73-76: horizon 32, minibatch 32768,
 * 5 mini-epochs; mixed_precision :50, normalize_input :51).  For a layer y = ELU(z), z = x w^T + b with gradient dz = dy * ELU'(z):
 *     dx    = dz . w          ppenv_mlp_layer_backward_input  (the forward kernels on the transposed weight image)
 *     dw    = dz^T . x        ppenv_mlp_layer_backward_weight (transposed LDS reads, ds_read_b64_tr_b16; split over m)
 *     db    = column sums of dz, produced by the launch that produces dz: per-64-row-block partials + ppenv_mlp_reduce_rows
 * ELU' needs no saved pre-activation: ELU'(z) = 1 for y > 0, y + 1 otherwise — the launch that computes the dx of layer l + 1 multiplies
 * it by ELU'(y_l) on the way out (y_l = that layer's input, saved by the forward), so what it writes IS dz_l.
 * Gradients are fp16 where autocast's are (dz, dx) and fp32 where the parameters are (dw, db); a loss scale is the caller's (linear). */

/* dx in forward form: g->in = dz [m, g->k] (g->k = the forward layer's n), g->w = the TRANSPOSED weights [g->n, g->k] (g->n = the forward
 * layer's k; ppenv_mlp_cast_weights writes that image), g->out = dx fp16 [m, g->n]; g->bias NULL, g->elu 0, g->out_f32 0.  g->in_f32 = 1
 * takes the fp32 head gradients [m, num_actions + 1] directly.
 *   elu_out (fp16 [m, g->n], may be NULL): dx *= ELU'(elu_out) elementwise — elu_out is the ELU output the gradient flows back through.
 *   colsum_partial (fp32 [ceil(m / 64), ld_colsum], may be NULL): row i receives the column sums of the written dx over rows 64 i ..
 *   64 i + 63; ppenv_mlp_reduce_rows over those rows is the bias gradient of the layer below.
 * Strides per batch entry as in ppenv_mlp_layer. */
int ppenv_mlp_layer_backward_input(const ppenv_mlp_layer* g, const uint16_t* elu_out, int64_t elu_out_stride, int32_t ld_elu_out,
                                   float* colsum_partial, int64_t colsum_stride, int32_t ld_colsum, void* stream);

typedef struct ppenv_mlp_dw {
    int32_t m, n, k, batch;   /* of the forward layer: dw[n, k] = sum over the m rows of dz[m, n] x[m, k] */
    const uint16_t* dz;       int64_t dz_stride;  int32_t lddz;   /* fp16 [m, lddz]  */
    const uint16_t* x;        int64_t x_stride;   int32_t ldx;    /* fp16 [m, ldx]: the layer's input as the forward read it */
    float* dw;                int64_t dw_stride;  int32_t lddw;   /* fp32 [n, lddw] */
    int32_t accumulate;       /* 1: dw += (gradient accumulation), 0: dw = */
    int32_t splits;           /* workgroups along m per output tile: 0 = chosen by the library; else a power of two <= m / 64 */
    void* workspace;          size_t workspace_bytes;             /* >= ppenv_mlp_dw_workspace_bytes(d) (0 when splits == 1) */
} ppenv_mlp_dw;

/* Needs m % 64 == 0, n % 8 == 0, k % 8 == 0, 16-byte aligned rows (lddz, ldx, strides multiples of 8; base pointers 16-byte aligned).
 * With splits > 1 the partial tiles are summed in a fixed order by a second small launch: results are deterministic. */
size_t ppenv_mlp_dw_workspace_bytes(const ppenv_mlp_dw* d);
int ppenv_mlp_layer_backward_weight(const ppenv_mlp_dw* d, void* stream);

/* out[i] (+)= sum over r < rows of partial[r * row_stride + i], i < n, in row order (the bias gradient from backward_input's partials). */
int ppenv_mlp_reduce_rows(const float* partial, int32_t rows, int64_t row_stride, int64_t n, float* out, int32_t accumulate, void* stream);

/* Column sums of an fp32 [m, n] matrix (the heads' bias gradient from the loss's d mu | d value): workspace of
 * ppenv_mlp_bias_grad_workspace_bytes(m, n). */
size_t ppenv_mlp_bias_grad_workspace_bytes(int32_t m, int32_t n);
int ppenv_mlp_bias_grad_f32(const float* dz, int32_t m, int32_t n, int32_t ld, void* workspace, float* out, int32_t accumulate, void* stream);

/* The per-optimizer-step cast of mixed precision, both operand images at once: fp32 master weights w32 [n, k] -> w16 [n, ldw16] (rows
 * zero-padded to ldw16; NULL: not wanted) and wt16 [wt_rows, ldwt16] = the transpose, zero beyond k rows / n columns (NULL: not wanted). */
int ppenv_mlp_cast_weights(const float* w32, int32_t n, int32_t k, int32_t ldw32, uint16_t* w16, int32_t ldw16, uint16_t* wt16, int32_t ldwt16,
                           int32_t wt_rows, void* stream);

/* The same for up to 32 matrices in ONE launch (a whole network: its weight matrices, and its bias vectors as 1-row matrices). */
typedef struct ppenv_mlp_cast {
    const float* w32;   int32_t n, k, ldw32;
    uint16_t* w16;      int32_t ldw16;            /* NULL: not wanted */
    uint16_t* wt16;     int32_t ldwt16, wt_rows;  /* NULL: not wanted */
} ppenv_mlp_cast;
int ppenv_mlp_cast_weights_batch(const ppenv_mlp_cast* items, int32_t count, void* stream);

/* rl_games' RunningMeanStd in training mode (normalize_input, yaml:51) on one batch obs [m, k] fp32: batch mean and unbiased batch
 * variance per column merged into the running float64 (mean, var, count) by the parallel-moments rule; also writes the fp32 mean and
 * 1 / sqrt(var + eps) that ppenv_mlp_prepare_input reads (either may be NULL).  One launch, one pass over obs.  workspace:
 * ppenv_running_mean_std_workspace_bytes(m, k) bytes, 8-byte aligned, its first 1024 bytes zeroed once by the caller (tickets the kernel
 * re-arms itself); k <= 16256.  rl_games is not part of the reference: restated from its published running_mean_std.py — parity unpinned. */
size_t ppenv_running_mean_std_workspace_bytes(int32_t m, int32_t k);
int ppenv_running_mean_std_update(const float* obs, int32_t m, int32_t k, int32_t ld, double* mean, double* var, double* count, float* mean_f32,
                                  float* inv_std_f32, float eps, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPENV_POLICY_H */
