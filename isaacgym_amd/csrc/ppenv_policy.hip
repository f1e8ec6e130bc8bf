// ppenv_policy.hip — one dense layer of the policy MLP on the matrix cores (include/ppenv_policy.h).
//
// out[M, N] = act(in[M, K] * W[N, K]^T + bias): both operands are K-contiguous (activations row-major, torch.nn.Linear weights
// [out, in]), which is exactly what v_mfma_f32_32x32x16_f16 wants: lane l (r = l & 31, h = l >> 5) feeds A[row r][k = 8h .. 8h+7] and
// B[k = 8h .. 8h+7][col r], i.e. eight consecutive fp16 of one row of `in` and of one row of `W` — one 16-byte LDS read each.
//
// Tiling: a 256-thread workgroup (4 waves, 2 x 2) owns a 128 x 128 tile of `out`; each wave a 64 x 64 quadrant = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator registers).  K advances in steps of 64 through a double-buffered LDS image (rows padded to 72 fp16: the
// eight lanes of a 128-byte LDS phase then hit 32 distinct banks), so a step is 16 MFMAs per wave between two barriers, with the next
// step's global loads in flight meanwhile.  The first layer stages fp32 observations: (x - mean) * inv_std, clamp, cast on the way
// into LDS.  Epilogue on the accumulators: + bias, ELU, fp16; a lane holds one column of 16 rows, so the 32 lanes of a half-wave
// store 64 contiguous bytes of one output row.
//
// Grid: x = N tiles (fastest: consecutive workgroups share the same rows of `in`), y = M tiles, z = batch.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdio>

#include "../../include/ppenv.h"
#include "../../include/ppenv_policy.h"

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
constexpr int BM = 128, BN = 128, BK = 64, LDS_LD = BK + 8;   // fp16 elements per LDS row
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct Args {
    int m, n, k, lda, ldw, ldo, elu, out_f32;
    const void* in; long long in_stride;
    const float* mean; const float* inv_std; float clip;
    const _Float16* w; long long w_stride;
    const _Float16* bias; long long bias_stride;
    void* out; long long out_stride;
};

// global -> registers: this thread's share of a 128 x 64 fp16 tile (rows row0.., k from k0), zero outside [rows, kmax).
// 128 rows x 8 chunks of 8 fp16 = 1024 chunks, 4 per thread: chunk c = tid + 256 i -> row c >> 3, k-chunk c & 7.
__device__ __forceinline__ void load_tile_h(const _Float16* __restrict__ base, int ld, int rows, int kmax, int row0, int k0, int tid, h8 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + 256 * i, r = row0 + (c >> 3), kk = k0 + (c & 7) * 8;
        h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < rows) {
            const _Float16* p = base + (size_t)r * ld + kk;
            if (kk + 8 <= kmax && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) x = *reinterpret_cast<const h8*>(p);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) if (kk + j < kmax) x[j] = p[j];
            }
        }
        v[i] = x;
    }
}
// the first layer: fp32 observations, normalised and clamped on the way (rl_games RunningMeanStd, eval mode)
__device__ __forceinline__ void load_tile_obs(const float* __restrict__ base, int ld, int rows, int kmax, int row0, int k0, int tid,
                                              const float* __restrict__ mean, const float* __restrict__ inv_std, float clip, h8 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + 256 * i, r = row0 + (c >> 3), kk = k0 + (c & 7) * 8;
        h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < rows) {
            const float* p = base + (size_t)r * ld + kk;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (kk + j < kmax) {
                    float f = p[j];
                    if (mean) f = fminf(fmaxf((f - mean[kk + j]) * inv_std[kk + j], -clip), clip);
                    x[j] = (_Float16)f;
                }
        }
        v[i] = x;
    }
}
__device__ __forceinline__ void store_tile(_Float16* __restrict__ s, int tid, const h8 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + 256 * i;
        *reinterpret_cast<h8*>(&s[(c >> 3) * LDS_LD + (c & 7) * 8]) = v[i];
    }
}

template <bool OBS>
__global__ __launch_bounds__(256) void mlp_layer_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, b = blockIdx.z;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* inh = OBS ? nullptr : reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    const float* inf = OBS ? reinterpret_cast<const float*>(a.in) + (size_t)b * a.in_stride : nullptr;

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    h8 ra[4], rb[4];
    const int ksteps = (a.k + BK - 1) / BK;
    if (OBS) load_tile_obs(inf, a.lda, a.m, a.k, m0, 0, tid, a.mean, a.inv_std, a.clip, ra); else load_tile_h(inh, a.lda, a.m, a.k, m0, 0, tid, ra);
    load_tile_h(W, a.ldw, a.n, a.k, n0, 0, tid, rb);
    store_tile(sA[0], tid, ra);
    store_tile(sB[0], tid, rb);
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    for (int ks = 0; ks < ksteps; ks++) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) {   // next step's global loads in flight during this step's MFMAs
            if (OBS) load_tile_obs(inf, a.lda, a.m, a.k, m0, (ks + 1) * BK, tid, a.mean, a.inv_std, a.clip, ra); else load_tile_h(inh, a.lda, a.m, a.k, m0, (ks + 1) * BK, tid, ra);
            load_tile_h(W, a.ldw, a.n, a.k, n0, (ks + 1) * BK, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; kk++) {
            h8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; i++) fa[i] = *reinterpret_cast<const h8*>(&sA[cur][(wm * 64 + i * 32 + r) * LDS_LD + kk * 16 + h * 8]);
#pragma unroll
            for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const h8*>(&sB[cur][(wn * 64 + j * 32 + r) * LDS_LD + kk * 16 + h * 8]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < ksteps) {
            store_tile(sA[cur ^ 1], tid, ra);   // the other buffer: last read before the barrier that ended step ks - 1
            store_tile(sB[cur ^ 1], tid, rb);
        }
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 MFMA — col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const _Float16* bias = a.bias ? a.bias + (size_t)b * a.bias_stride : nullptr;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = n0 + wn * 64 + j * 32 + r;
        if (col >= a.n) continue;
        const float bv = bias ? (float)bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (row >= a.m) continue;
                float x = acc[i][j][reg] + bv;
                if (a.elu) x = x > 0.f ? x : __expf(x) - 1.0f;
                if (a.out_f32) (reinterpret_cast<float*>(a.out) + (size_t)b * a.out_stride)[(size_t)row * a.ldo + col] = x;
                else (reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.out_stride)[(size_t)row * a.ldo + col] = (_Float16)x;
            }
        }
    }
}
}  // namespace

extern "C" int ppenv_mlp_layer_forward(const ppenv_mlp_layer* L, void* stream) {
    if (!L || !L->in || !L->w || !L->out || L->m <= 0 || L->n <= 0 || L->k <= 0 || L->batch <= 0 || L->lda < L->k || L->ldw < L->k || L->ldo < L->n) {
        ppenv_set_error("ppenv_mlp_layer_forward: NULL pointer or inconsistent sizes (need lda >= k, ldw >= k, ldo >= n)");
        return PPENV_EINVAL;
    }
    if (L->in_f32 && ((L->mean == nullptr) != (L->inv_std == nullptr))) { ppenv_set_error("ppenv_mlp_layer_forward: mean and inv_std go together"); return PPENV_EINVAL; }
    Args a{L->m, L->n, L->k, L->lda, L->ldw, L->ldo, L->elu, L->out_f32, L->in, (long long)L->in_stride, L->mean, L->inv_std, L->clip,
           reinterpret_cast<const _Float16*>(L->w), (long long)L->w_stride, reinterpret_cast<const _Float16*>(L->bias), (long long)L->bias_stride,
           L->out, (long long)L->out_stride};
    const dim3 grid((L->n + BN - 1) / BN, (L->m + BM - 1) / BM, L->batch), block(256);
    if (L->in_f32) hipLaunchKernelGGL(mlp_layer_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_layer_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_layer_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
