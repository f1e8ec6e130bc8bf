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
#include <cstdlib>

#include "../../include/ppenv.h"
#include "../../include/ppenv_policy.h"

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
constexpr int PATCH_LD = 72;   // row stride (fp16) of the epilogue patch; the operand tiles use BK + 8 (BK = K step, a template parameter)
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

// global -> registers: this thread's share of a ROWS x BK fp16 tile (rows row0.., k from k0), zero outside [rows, kmax).
// ROWS rows x BK / 8 chunks of 8 fp16, C = ROWS * (BK / 8) / T per thread: chunk c = tid + T i -> row c / (BK / 8), k-chunk c % (BK / 8).
template <int T, int C, int BK>
__device__ __forceinline__ void load_tile_h(const _Float16* __restrict__ base, int ld, int rows, int kmax, int row0, int k0, int tid, h8 (&v)[C]) {
    constexpr int CPR = BK / 8;
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int c = tid + T * i, r = row0 + c / CPR, kk = k0 + (c % CPR) * 8;
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
template <int T, int C, int BK>
__device__ __forceinline__ void load_tile_obs(const float* __restrict__ base, int ld, int rows, int kmax, int row0, int k0, int tid,
                                              const float* __restrict__ mean, const float* __restrict__ inv_std, float clip, h8 (&v)[C]) {
    constexpr int CPR = BK / 8;
    static_assert(T % CPR == 0, "a thread's k offset inside the tile must not depend on the chunk");
    const int kk = k0 + (tid % CPR) * 8;          // the same for all of this thread's chunks: its eight statistics are fetched once
    float mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const bool in = kk + j < kmax; mu[j] = (mean && in) ? mean[kk + j] : 0.f; is[j] = (mean && in) ? inv_std[kk + j] : 1.f; }
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && kk + 8 <= kmax;   // 16-byte aligned rows: two float4 per chunk
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int c = tid + T * i, r = row0 + c / CPR;
        h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < rows) {
            const float* p = base + (size_t)r * ld + kk;
            float f[8];
            if (vec) {
                const f4v lo = *reinterpret_cast<const f4v*>(p), hi = *reinterpret_cast<const f4v*>(p + 4);
#pragma unroll
                for (int j = 0; j < 4; j++) { f[j] = lo[j]; f[4 + j] = hi[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) f[j] = kk + j < kmax ? p[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float g = (f[j] - mu[j]) * is[j];
                if (mean) g = fminf(fmaxf(g, -clip), clip);
                x[j] = kk + j < kmax ? (_Float16)g : (_Float16)0.f;
            }
        }
        v[i] = x;
    }
}
template <int T, int C, int BK>
__device__ __forceinline__ void store_tile(_Float16* __restrict__ s, int tid, const h8 (&v)[C]) {
    constexpr int CPR = BK / 8, LDS_LD = BK + 8;
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int c = tid + T * i;
        *reinterpret_cast<h8*>(&s[(c / CPR) * LDS_LD + (c % CPR) * 8]) = v[i];
    }
}

// WM x WN waves per workgroup, (32 TI) x (32 TJ) of `out` per wave (TI x TJ MFMA tiles, 16 accumulator registers each): the
// workgroup's tile is BM = 32 TI WM rows by BN = 32 TJ WN columns.  A K sub-step of 16 costs a wave TI + TJ fragment reads (16 bytes
// per lane each) for TI TJ MFMAs: 1 read per MFMA at 2 x 2, 0.75 at 4 x 2, 0.5 at 4 x 4 — the LDS read traffic, not the global
// traffic, is what the small wave tile pays for.
template <bool OBS, int WM, int WN, int TI, int TJ, int BK>
__global__ __launch_bounds__(64 * WM * WN) void mlp_layer_kernel(const Args a) {
    constexpr int T = 64 * WM * WN, BM = 32 * TI * WM, BN = 32 * TJ * WN, CPR = BK / 8, CA = BM * CPR / T, CB = BN * CPR / T, LDS_LD = BK + 8;
    static_assert(BM * CPR % T == 0 && BN * CPR % T == 0, "staging shares");
    static_assert(TI % 2 == 0 && TJ % 2 == 0, "the epilogue works on 64 x 64 blocks");
    constexpr int kOperand = 2 * (BM + BN) * LDS_LD, kPatch = WM * WN * 64 * PATCH_LD;
    __shared__ __attribute__((aligned(16))) _Float16 smem[kOperand > kPatch ? kOperand : kPatch];   // [A buf 0 | A buf 1 | B buf 0 | B buf 1]; afterwards the epilogue patches
    _Float16* const sA[2] = {smem, smem + BM * LDS_LD};
    _Float16* const sB[2] = {smem + 2 * BM * LDS_LD, smem + 2 * BM * LDS_LD + BN * LDS_LD};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, b = blockIdx.z;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* inh = OBS ? nullptr : reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    const float* inf = OBS ? reinterpret_cast<const float*>(a.in) + (size_t)b * a.in_stride : nullptr;

    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // Operand pipeline: tile ks is in LDS, tiles ks + 1 and ks + 2 are on their way in two register sets.  A global load has two
    // K-steps to land before its registers are written to LDS.
    h8 ra[2][CA], rb[2][CB];
    const int ksteps = (a.k + BK - 1) / BK;
    auto gload = [&](int set, int ks) {
        if (ks >= ksteps) return;
        if (OBS) load_tile_obs<T, CA, BK>(inf, a.lda, a.m, a.k, m0, ks * BK, tid, a.mean, a.inv_std, a.clip, ra[set]); else load_tile_h<T, CA, BK>(inh, a.lda, a.m, a.k, m0, ks * BK, tid, ra[set]);
        load_tile_h<T, CB, BK>(W, a.ldw, a.n, a.k, n0, ks * BK, tid, rb[set]);
    };
    gload(0, 0);
    store_tile<T, CA, BK>(sA[0], tid, ra[0]);
    store_tile<T, CB, BK>(sB[0], tid, rb[0]);
    gload(1, 1);
    gload(0, 2);
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    auto kstep = [&](int ks, int set_next) {   // set_next: the register set that holds tile ks + 1 (and is refilled with tile ks + 3)
        const int cur = ks & 1;
        // fragments of sub-step kk + 1 are read while the MFMAs of sub-step kk run (two fragment sets)
        h8 fa[2][TI], fb[2][TJ];
        auto frags = [&](int set, int kk) {
#pragma unroll
            for (int i = 0; i < TI; i++) fa[set][i] = *reinterpret_cast<const h8*>(&sA[cur][(wm * 32 * TI + i * 32 + r) * LDS_LD + kk * 16 + h * 8]);
#pragma unroll
            for (int j = 0; j < TJ; j++) fb[set][j] = *reinterpret_cast<const h8*>(&sB[cur][(wn * 32 * TJ + j * 32 + r) * LDS_LD + kk * 16 + h * 8]);
        };
        frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; kk++) {
            if (kk + 1 < BK / 16) frags((kk + 1) & 1, kk + 1);
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < ksteps) {
            store_tile<T, CA, BK>(sA[cur ^ 1], tid, ra[set_next]);   // the other buffer: last read before the barrier that ended step ks - 1
            store_tile<T, CB, BK>(sB[cur ^ 1], tid, rb[set_next]);
        }
        gload(set_next, ks + 3);
        __syncthreads();
    };
    for (int ks = 0; ks < ksteps; ks += 2) {
        kstep(ks, 1);
        if (ks + 1 < ksteps) kstep(ks + 1, 0);
    }
    // epilogue: C/D layout of the 32x32 MFMA — col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane holds ONE column,
    // so stored straight from the accumulators every store instruction would move 2 bytes per lane.  fp16 results therefore go through
    // the wave's own 64 x 64 patch of LDS (the operand buffers are free after the last barrier), one 64 x 64 block of the wave's tile at
    // a time, and leave as 16 bytes per lane, 128 contiguous bytes per output row.  The fp32 heads (a few columns) are stored directly.
    const _Float16* bias = a.bias ? a.bias + (size_t)b * a.bias_stride : nullptr;
    const int wrow0 = m0 + wm * 32 * TI, wcol0 = n0 + wn * 32 * TJ;
    if (!a.out_f32) {
        __syncthreads();                                   // every wave is done with the operand buffers: the patches overlay them
        _Float16* patch = smem + wave * (64 * PATCH_LD);
        _Float16* out = reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.out_stride;
        const int rl = lane >> 3, ch = lane & 7;
#pragma unroll
        for (int ib = 0; ib < TI; ib += 2)
#pragma unroll
            for (int jb = 0; jb < TJ; jb += 2) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = wcol0 + (jb + j) * 32 + r;
                    const float bv = (bias && col < a.n) ? (float)bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) {
                            float x = acc[ib + i][jb + j][reg] + bv;
                            if (a.elu) x = x > 0.f ? x : __expf(x) - 1.0f;
                            patch[(i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * PATCH_LD + j * 32 + r] = (_Float16)x;
                        }
                }
                __builtin_amdgcn_wave_barrier();   // the patch is this wave's own: DS operations of a wave execute in order
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int prow = it * 8 + rl, row = wrow0 + ib * 32 + prow, col = wcol0 + jb * 32 + ch * 8;
                    if (row >= a.m || col >= a.n) continue;
                    const h8 v = *reinterpret_cast<const h8*>(&patch[prow * PATCH_LD + ch * 8]);
                    _Float16* dst = out + (size_t)row * a.ldo + col;
                    if (col + 8 <= a.n && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) *reinterpret_cast<h8*>(dst) = v;
                    else {
#pragma unroll
                        for (int q = 0; q < 8; q++) if (col + q < a.n) dst[q] = v[q];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
        const int col = wcol0 + j * 32 + r;
        if (col >= a.n) continue;
        const float bv = bias ? (float)bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; i++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = wrow0 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (row >= a.m) continue;
                float x = acc[i][j][reg] + bv;
                if (a.elu) x = x > 0.f ? x : __expf(x) - 1.0f;
                (reinterpret_cast<float*>(a.out) + (size_t)b * a.out_stride)[(size_t)row * a.ldo + col] = x;
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
    // Tile choice.  Measured on the reference's layers (tools/gpu_mlp_layers.py, M = 16384, TFLOP/s on the 2048 -> 1536 layer):
    //   128 x 128, 4 waves of 64 x 64, BK 64 (two workgroups per CU)            600
    //   256 x 128, 4 waves of 128 x 64 (one workgroup of four waves per CU)      507   too few waves to hide anything
    //   256 x 256, 4 waves of 128 x 128                                          157   512 registers and still spilling
    //   256 x 256, 8 waves of 128 x 64, BK 64 (one workgroup of 8 waves per CU)  727
    // so: the big tile whenever it still gives most CUs a workgroup, else the small one.  PPENV_MLP_TILE = 128 | 384 | 385 (BK 32) forces one.
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PPENV_MLP_TILE"); forced = e ? atoi(e) : 0; }
    auto wgs = [&](int bm, int bn) { return (long long)((L->n + bn - 1) / bn) * ((L->m + bm - 1) / bm) * L->batch; };
    int cfg = forced;
    if (cfg == 0) {
        // the first layer (fp32 observations, K = 80 or 313) converts its obs tile once per column tile: wide tiles and a K step of 32
        // (less zero padding of K) — 54 us against 86 (256 x 256 / BK 32 vs 128 x 128 / BK 64, M = 4096, K = 313)
        if (L->in_f32) cfg = (wgs(256, 256) >= 128 && L->n >= 256) ? 385 : 129;
        else cfg = (wgs(256, 256) >= 192 && L->n >= 256) ? 384 : 128;
    }
#define PP_LAUNCH(WM_, WN_, TI_, TJ_, BK_)                                                                                                    \
    do {                                                                                                                                      \
        const dim3 grid((L->n + 32 * TJ_ * WN_ - 1) / (32 * TJ_ * WN_), (L->m + 32 * TI_ * WM_ - 1) / (32 * TI_ * WM_), L->batch), block(64 * WM_ * WN_); \
        if (L->in_f32) hipLaunchKernelGGL((mlp_layer_kernel<true, WM_, WN_, TI_, TJ_, BK_>), grid, block, 0, (hipStream_t)stream, a);        \
        else hipLaunchKernelGGL((mlp_layer_kernel<false, WM_, WN_, TI_, TJ_, BK_>), grid, block, 0, (hipStream_t)stream, a);                  \
    } while (0)
    if (cfg == 384) PP_LAUNCH(2, 4, 4, 2, 64);
    else if (cfg == 385) PP_LAUNCH(2, 4, 4, 2, 32);
    else if (cfg == 129) PP_LAUNCH(2, 2, 2, 2, 32);
    else PP_LAUNCH(2, 2, 2, 2, 64);
#undef PP_LAUNCH
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_layer_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
