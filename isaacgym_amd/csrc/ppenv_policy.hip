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
#include <type_traits>

#include "../../include/ppenv.h"
#include "../../include/ppenv_policy.h"
#include "ppenv_device.h"   // dr_gauss: the counter RNG's standard normal

void ppenv_set_error(const char* msg);   // ppenv.hip

#ifndef PP_OUT_NT
#define PP_OUT_NT 0       // 1: the activations leave by non-temporal stores (experiment, tools/gpu_mlp_exp1.py)
#endif
#if PP_OUT_NT
#define PP_STORE_H8(dst, v) __builtin_nontemporal_store((v), reinterpret_cast<h8*>(dst))
#else
#define PP_STORE_H8(dst, v) (*reinterpret_cast<h8*>(dst) = (v))
#endif
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
    // backward-input mode (ppenv_mlp_layer_backward_input): out = (in . w^T) * ELU'(aux), aux = the saved ELU output this gradient flows
    // back through; colsum = per-64-row-block column sums of `out` (the bias gradient of the layer below, summed by ppenv_mlp_reduce_rows)
    const _Float16* aux; long long aux_stride; int ldaux;
    float* colsum; long long colsum_stride; int ldcs;
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

// Epilogue on the accumulators.  C/D layout of the 32x32 MFMA — col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane
// holds ONE column, so stored straight from the accumulators every store instruction would move 2 bytes per lane.  fp16 results
// therefore go through the wave's own 64 x 64 patch of LDS (the operand buffers are free after the barrier in here), one 64 x 64 block
// of the wave's tile at a time, and leave as 16 bytes per lane, 128 contiguous bytes per output row.  The fp32 heads (a few columns)
// are stored directly.  wrow0 / wcol0: the wave's first row / column of `out`.
template <int TI, int TJ>
__device__ __forceinline__ void epilogue(const Args& a, f16v (&acc)[TI][TJ], _Float16* smem, int wrow0, int wcol0, int b) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const _Float16* bias = a.bias ? a.bias + (size_t)b * a.bias_stride : nullptr;
    if (!a.out_f32) {
        __syncthreads();                                   // every wave is done with the operand buffers: the patches overlay them
        _Float16* patch = smem + wave * (64 * PATCH_LD);
        _Float16* out = reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.out_stride;
        constexpr int JW = TJ >= 2 ? 2 : 1, CH = 4 * JW;        // the block is 64 x 32 JW: CH 16-byte chunks per row, 64 / CH rows per store pass
        const int rl = lane / CH, ch = lane % CH;
#pragma unroll
        for (int ib = 0; ib < TI; ib += 2)
#pragma unroll
            for (int jb = 0; jb < TJ; jb += JW) {
                // backward-input mode: the block's ELU outputs are fetched BEFORE the accumulators go through the patch, so the loads'
                // latency hides behind the conversions and LDS writes instead of sitting in front of every store
                h8 yv[CH];
                if (a.aux) {
#pragma unroll
                    for (int it = 0; it < CH; it++) {
                        const int row = wrow0 + ib * 32 + it * (64 / CH) + rl, col = wcol0 + jb * 32 + ch * 8;
                        h8 y = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (row < a.m && col < a.n) {
                            const _Float16* ap = a.aux + (size_t)b * a.aux_stride + (size_t)row * a.ldaux + col;
                            if (col + 8 <= a.n && ((reinterpret_cast<uintptr_t>(ap) & 15) == 0)) y = *reinterpret_cast<const h8*>(ap);
                            else {
#pragma unroll
                                for (int q = 0; q < 8; q++) if (col + q < a.n) y[q] = ap[q];
                            }
                        }
                        yv[it] = y;
                    }
                }
#pragma unroll
                for (int j = 0; j < JW; j++) {
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
                float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // backward-input mode: this lane's eight columns summed over the block's rows
#pragma unroll
                for (int it = 0; it < CH; it++) {
                    const int prow = it * (64 / CH) + rl, row = wrow0 + ib * 32 + prow, col = wcol0 + jb * 32 + ch * 8;
                    if (row >= a.m || col >= a.n) continue;
                    h8 v = *reinterpret_cast<const h8*>(&patch[prow * PATCH_LD + ch * 8]);
                    if (a.aux) {                                   // gradient through the ELU whose OUTPUT is aux: ELU'(z) = y > 0 ? 1 : y + 1
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const float yf = (float)yv[it][q];
                            v[q] = (_Float16)((float)v[q] * (yf > 0.f ? 1.0f : yf + 1.0f));
                        }
                    }
                    if (a.colsum) {
#pragma unroll
                        for (int q = 0; q < 8; q++) cs[q] += col + q < a.n ? (float)v[q] : 0.f;
                    }
                    _Float16* dst = out + (size_t)row * a.ldo + col;
                    if (col + 8 <= a.n && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) PP_STORE_H8(dst, v);
                    else {
#pragma unroll
                        for (int q = 0; q < 8; q++) if (col + q < a.n) dst[q] = v[q];
                    }
                }
                if (a.colsum) {                                    // lanes that share ch differ in rl: butterfly over the rl bits, lane rl = 0 writes
#pragma unroll
                    for (int d = CH; d < 64; d <<= 1)
#pragma unroll
                        for (int q = 0; q < 8; q++) cs[q] += __shfl_xor(cs[q], d);
                    const int col = wcol0 + jb * 32 + ch * 8, brow = wrow0 + ib * 32;
                    if (rl == 0 && brow < a.m) {
                        float* dst = a.colsum + (size_t)b * a.colsum_stride + (size_t)(brow >> 6) * a.ldcs + col;
#pragma unroll
                        for (int q = 0; q < 8; q++) if (col + q < a.n) dst[q] = cs[q];
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

// The same for accumulators of v_mfma_f32_16x16x32_f16 issued with the operands swapped (A := the W fragment, B := the `in` fragment),
// i.e. holding out^T: lane l has out[row l & 15][columns 4 (l >> 4) .. + 3] of each 16 x 16 tile — four consecutive columns, one
// 8-byte patch store per tile.  MT x NT tiles per wave (MT, NT multiples of 4: 64 x 64 blocks as above).
template <int MT, int NT>
__device__ __forceinline__ void epilogue16(const Args& a, f4v (&acc)[MT][NT], _Float16* smem, int wrow0, int wcol0, int b) {
    static_assert(MT % 4 == 0 && NT <= 4, "blocks of 64 rows x 16 NT <= 64 columns");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const _Float16* bias = a.bias ? a.bias + (size_t)b * a.bias_stride : nullptr;
    if (!a.out_f32) __syncthreads();                       // every wave is done with the operand buffers: the patches overlay them
    _Float16* patch = smem + wave * (64 * PATCH_LD);
    float bv[NT][4];
#pragma unroll
    for (int jn = 0; jn < NT; jn++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const int col = wcol0 + jn * 16 + 4 * q + e; bv[jn][e] = (bias && col < a.n) ? (float)bias[col] : 0.f; }
#pragma unroll
    for (int mb = 0; mb < MT; mb += 4) {
#pragma unroll
        for (int jn = 0; jn < NT; jn++) {
            const int col = wcol0 + jn * 16 + 4 * q;
#pragma unroll
            for (int im = 0; im < 4; im++) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    x[e] = acc[mb + im][jn][e] + bv[jn][e];
                    if (a.elu) x[e] = x[e] > 0.f ? x[e] : __expf(x[e]) - 1.0f;
                }
                if (!a.out_f32) {
                    const h4 v = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
                    *reinterpret_cast<h4*>(&patch[(im * 16 + r) * PATCH_LD + jn * 16 + 4 * q]) = v;
                } else {
                    const int row = wrow0 + (mb + im) * 16 + r;
                    float* dst = reinterpret_cast<float*>(a.out) + (size_t)b * a.out_stride + (size_t)row * a.ldo + col;
#pragma unroll
                    for (int e = 0; e < 4; e++) if (row < a.m && col + e < a.n) dst[e] = x[e];
                }
            }
        }
        if (a.out_f32) continue;
        _Float16* out = reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.out_stride;
        __builtin_amdgcn_wave_barrier();
        constexpr int CPR = 2 * NT;                        // 16-byte chunks per patch row
        // backward-input mode (NT = 4 only: 64 lanes = 8 rows x 8 chunks, so a lane keeps its eight columns over the passes — as in epilogue())
        const bool bwd = NT == 4 && (a.aux || a.colsum);
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < CPR; it++) {                 // 64 rows x CPR chunks = 64 CPR chunks, 64 per pass
            const int c = it * 64 + lane, prow = c / CPR, ch = c % CPR;
            const int row = wrow0 + mb * 16 + prow, col = wcol0 + ch * 8;
            if (row >= a.m || col >= a.n) continue;
            h8 v = *reinterpret_cast<const h8*>(&patch[prow * PATCH_LD + ch * 8]);
            if (bwd) {
                if (a.aux) {                               // (the loads sit in the store pass here: this kernel's blocks come two to a wave, the next block's
                    const _Float16* ap = a.aux + (size_t)b * a.aux_stride + (size_t)row * a.ldaux + col;   // conversions hide them)
                    h8 y = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (col + 8 <= a.n && ((reinterpret_cast<uintptr_t>(ap) & 15) == 0)) y = *reinterpret_cast<const h8*>(ap);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; e++) if (col + e < a.n) y[e] = ap[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float yf = (float)y[e];
                        v[e] = (_Float16)((float)v[e] * (yf > 0.f ? 1.0f : yf + 1.0f));
                    }
                }
                if (a.colsum) {
#pragma unroll
                    for (int e = 0; e < 8; e++) cs[e] += col + e < a.n ? (float)v[e] : 0.f;
                }
            }
            _Float16* dst = out + (size_t)row * a.ldo + col;
            if (col + 8 <= a.n && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) PP_STORE_H8(dst, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; e++) if (col + e < a.n) dst[e] = v[e];
            }
        }
        if (bwd && a.colsum) {                             // NT = 4: lane = 8 rl + ch; butterfly over rl, lane rl = 0 writes the block's sums
#pragma unroll
            for (int d = 8; d < 64; d <<= 1)
#pragma unroll
                for (int e = 0; e < 8; e++) cs[e] += __shfl_xor(cs[e], d);
            const int col = wcol0 + (lane & 7) * 8, brow = wrow0 + mb * 16;
            if ((lane >> 3) == 0 && brow < a.m) {
                float* dst = a.colsum + (size_t)b * a.colsum_stride + (size_t)(brow >> 6) * a.ldcs + col;
#pragma unroll
                for (int e = 0; e < 8; e++) if (col + e < a.n) dst[e] = cs[e];
            }
        }
        __builtin_amdgcn_wave_barrier();
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
    static_assert(TI % 2 == 0 && (TJ % 2 == 0 || TJ == 1), "the epilogue works on 64 x 64 (or 64 x 32) blocks");
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
    epilogue<TI, TJ>(a, acc, smem, m0 + wm * 32 * TI, n0 + wn * 32 * TJ, b);
}

// ---- the 256 x 256 tile for the wide layers: LDS-DMA staging and two wave groups taking turns on the matrix cores ----
//
// 512 threads = 8 waves as 2 (M) x 4 (N), 128 x 64 of `out` per wave (4 x 2 MFMA tiles, 128 accumulator registers), K in tiles of 64.
// The two waves of a SIMD belong to different groups (wm = wave >> 2), and group 1 runs one barrier behind group 0: while one group
// issues its eight MFMAs of a phase the other reads its next fragments from LDS, so a SIMD's matrix core is fed by one wave while the
// other waits for the LDS — neither needs a second fragment set.  A K tile is two phases, 64 rows of the wave's tile each (the B
// fragments are kept for the second):
//
//      phase   LDS reads (16 B per lane)           global -> LDS, next K tile (1 KiB pieces per wave)          MFMAs
//      A       A rows 0-63 (8), B cols 0-63 (8)    A rows 0-63 of both wave rows (2), B (4)                     16
//      B       A rows 64-127 (8)                   A rows 64-127 (2); all but these two have landed after it    16
//
// (Four phases of eight MFMAs — one 64 x 32 quadrant each — measured 3940 cycles per K tile against 2375 for the 64 MFMAs of a SIMD's
// two waves alone: 600 of the difference were the eight barriers.)
//
// Operands go global -> LDS by global_load_lds_dwordx4 (no registers, no ds_write pass).  One such instruction writes the wave's
// 64 x 16 bytes contiguously (eight 128-byte tile rows), so the LDS image cannot be padded; the bank conflicts of 128-byte rows are
// avoided by an XOR swizzle instead, applied on the SOURCE address: 16-byte slot s of row r holds k-chunk s ^ ((r >> 1) & 7), and a
// fragment read of k-chunk c of row r looks in slot c ^ ((r >> 1) & 7).  A ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32), each group wanting 16 distinct 16-byte slots of the 256-byte bank
// row = two tile rows: row parity picks the half, and (r >> 1) & 7 takes eight distinct values over the eight row pairs of a group.
//
// Ordering.  Tile t + 1 is issued during tile t into the other buffer, whose last readers (phase 3 of tile t - 1, the lagging group)
// have waited for their reads (lgkmcnt) and passed two barriers before the first piece is issued.  The vector-memory counter retires in
// issue order, so a counted wait says which pieces have landed: vmcnt(2) before the barrier that ends tile t for group 0 leaves only
// the two pieces nobody reads before phase B in flight, and vmcnt(6) before the barrier that ends group 0's phase A (six younger pieces
// issued by then) retires those.  A piece is read only after such a wait by its issuing waves AND a barrier the reader
// has passed.  Barriers are raw s_barrier: a __syncthreads() would drain the DMAs at every phase.
//
// Needs fp16 input, K a multiple of 64 and 16-byte aligned rows; rows beyond M / N are clamped on load and never stored.
// Diagnostic builds only (-DPPM_STAMP, tools/gpu_mlp_stamps.py): shader-clock stamps per workgroup and wave.
#if defined(PPM_STAMP)
__device__ unsigned long long mlp_stamp_buf[256 * 8 * 32];
#define PPM_STAMP_AT(k)                                                                                \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (lane == 0 && blockIdx.x < 256 && blockIdx.y == 0) mlp_stamp_buf[(blockIdx.x * 8 + wave) * 32 + (k)] = t_; \
    } while (0)
// the constant 100 MHz clock (s_memrealtime): comparable across workgroups and CUs, unlike the shader clock
#define PPM_STAMP_RT(k)                                                                               \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (lane == 0 && blockIdx.x < 256 && blockIdx.y == 0) mlp_stamp_buf[(blockIdx.x * 8 + wave) * 32 + (k)] = t_; \
    } while (0)
#else
#define PPM_STAMP_AT(k) do { } while (0)
#define PPM_STAMP_RT(k) do { } while (0)
#endif
#ifndef PP_EXP
#define PP_EXP 0          // timing experiments (diagnostic builds): 1 no fragment reads, 2 no DMA, 4 no barriers in the K loop, 8 no MFMAs, 16 no epilogue (8, 16: the ring kernel only) — results are wrong
#endif
#define PPM_TILE_STAMP(j) do { if (t == 8) PPM_STAMP_AT(j); else if (t == 9) PPM_STAMP_AT(9 + (j)); } while (0)

// M16: the same tile on v_mfma_f32_16x16x32_f16 (8 x 4 tiles of 16 x 16 per wave, operands swapped so the accumulators hold out^T, see
// epilogue16) instead of v_mfma_f32_32x32x16_f16 (4 x 2 tiles of 32 x 32): same fragment reads, same flops, half the K depth per
// instruction pair.
// NT (M16 only): 16-column tiles per wave — 4: 256 columns per workgroup; 3: 192, for the layer whose 256 x 256 grid would leave a
// quarter of the CUs without a workgroup (2048 -> 1536 at M = 4096: 192 tiles of 256 x 256, 256 of 256 x 192).
template <bool M16, int NT = 4>
__global__ __launch_bounds__(512) void mlp_layer_pp_kernel(const Args a, const int tiles_n, const int tiles_m) {
    static_assert(M16 ? (NT == 3 || NT == 4) : NT == 4, "wave tile 128 x 16 NT");
    constexpr int BM = 256, BN = 64 * NT, BK = 64, TI = 4, TJ = 2, NB = BN / 64;   // NB: 1 KiB-per-wave pieces of a B tile
    constexpr int kOperand = 2 * (BM + BN) * BK, kPatch = 8 * 64 * PATCH_LD;
    __shared__ __attribute__((aligned(16))) _Float16 smem[kOperand > kPatch ? kOperand : kPatch];   // [A 0 | B 0 | A 1 | B 1]; afterwards the epilogue patches
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;   // wave: scalar
    // consecutive workgroup ids go round the eight XCDs: give each XCD a contiguous run of tiles (neighbours share rows of `in`)
    const int nwg = tiles_n * tiles_m, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int n0 = (id % tiles_n) * BN, m0 = (id / tiles_n) * BM, b = blockIdx.y;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* in = reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    PPM_STAMP_RT(27); PPM_STAMP_AT(26);

    // staging: thread -> (row tid >> 3 (+ 64 per issue), slot tid & 7) of the tile; its source k-chunk is slot ^ ((row >> 1) & 7)
    const int srow = tid >> 3, kc = (lane & 7) ^ ((srow >> 1) & 7);
    const _Float16* pa[4];
    const _Float16* pb[NB];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int ra = m0 + i * 64 + srow; pa[i] = in + (size_t)(ra < a.m ? ra : a.m - 1) * a.lda + kc * 8; }
#pragma unroll
    for (int i = 0; i < NB; i++) { const int rb = n0 + i * 64 + srow; pb[i] = W + (size_t)(rb < a.n ? rb : a.n - 1) * a.ldw + kc * 8; }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // piece i of a tile = its rows 64 i .. 64 i + 63 (A: rows 0-63 / 64-127 of wave row i >> 1; B: 64 columns)
    auto stage_a = [&](int buf, int k0, int i) {
        __builtin_amdgcn_global_load_lds((glb_ptr)(pa[i] + k0), (lds_ptr)(smem + buf * (BM + BN) * BK + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };
    auto stage_b = [&](int buf, int k0) {                          // the whole B tile
#pragma unroll
        for (int i = 0; i < NB; i++)
            __builtin_amdgcn_global_load_lds((glb_ptr)(pb[i] + k0), (lds_ptr)(smem + buf * (BM + BN) * BK + BM * BK + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };

    // accumulators: 4 x 2 tiles of 32 x 32 (16 registers each), or 8 x 4 tiles of 16 x 16 (4 each): 128 registers either way
    f16v acc[M16 ? 1 : TI][M16 ? 1 : TJ];
    f4v acc16[M16 ? 8 : 1][M16 ? NT : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc16[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    }

    // fragment reads: lane -> row r of its MFMA tile, 16-byte k-chunk KH kk + h of the K tile (32 x 32 x 16: 32 rows, 2 chunks per
    // instruction, 4 instructions deep; 16 x 16 x 32: 16 rows, 4 chunks, 2 deep).  Tile origins are multiples of 16 rows, so a lane's
    // swizzle term (row >> 1) & 7 is that of r.
    constexpr int TR = M16 ? 16 : 32, KH = M16 ? 4 : 2, KD = M16 ? 2 : 4, TPH = 64 / TR;   // rows per tile; chunks per instruction; depth; tiles per 64 rows
    const int r = lane & (TR - 1), h = lane / TR;
    int swz[KD];
#pragma unroll
    for (int kk = 0; kk < KD; kk++) swz[kk] = ((KH * kk + h) ^ ((r >> 1) & 7)) * 8;
    constexpr int TB = M16 ? NT : 2;                               // B tiles per wave (of TR columns)
    const int arow = (wm * 128 + r) * BK, brow = BM * BK + (wn * TB * TR + r) * BK;
    int tcur = 0;
    h8 fa[TPH][KD], fb[TB][KD];                                    // A: the current 64-row half, [tile][kk]; B: [tile][kk] — 8 + 2 TB (8 or 6) reads
    auto read_a = [&](const _Float16* t, int half) {
        if ((PP_EXP & 1) && tcur > 0) return;
#pragma unroll
        for (int i = 0; i < TPH; i++)
#pragma unroll
            for (int kk = 0; kk < KD; kk++) fa[i][kk] = *reinterpret_cast<const h8*>(&t[arow + (half * TPH + i) * TR * BK + swz[kk]]);
    };
    auto read_b = [&](const _Float16* t) {
        if ((PP_EXP & 1) && tcur > 0) return;
#pragma unroll
        for (int j = 0; j < TB; j++)
#pragma unroll
            for (int kk = 0; kk < KD; kk++) fb[j][kk] = *reinterpret_cast<const h8*>(&t[brow + j * TR * BK + swz[kk]]);
    };
    auto mfmas = [&](int half) {                                   // 64 x 64 x 64: every accumulator of the half in turn
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KD; kk++)
#pragma unroll
            for (int i = 0; i < TPH; i++)
#pragma unroll
                for (int j = 0; j < TB; j++) {
                    if constexpr (M16) acc16[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][kk], fa[i][kk], acc16[half * 4 + i][j], 0, 0, 0);
                    else acc[half * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][kk], fb[j][kk], acc[half * 2 + i][j], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); if (!(PP_EXP & 4)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

    const int ktiles = a.k / BK;
    PPM_STAMP_AT(30);
    stage_a(0, 0, 0); stage_a(0, 0, 2);                            // the order of every tile: what phase A reads first, A rows 64-127 last
    stage_b(0, 0);
    stage_a(0, 0, 1); stage_a(0, 0, 3);
    PP_VM(2);
    PP_BARRIER();
    if (wm == 1) PP_BARRIER();                                     // group 1 runs one barrier behind
    PPM_STAMP_AT(25);
    for (int t = 0; t < ktiles; t++) {
        const _Float16* tile = smem + (t & 1) * (BM + BN) * BK;
        const bool more = t + 1 < ktiles && !(PP_EXP & 2);
        tcur = t;
        const int nbuf = (t + 1) & 1, nk0 = (t + 1) * BK;
        PPM_TILE_STAMP(0);
        // phase A: rows 0-63 of the wave's tile
        read_a(tile, 0);
        read_b(tile);
        if (more) { stage_a(nbuf, nk0, 0); stage_a(nbuf, nk0, 2); stage_b(nbuf, nk0); }
        // rows 64-127 of THIS tile's A (issued last, a tile ago) are read in phase B: all but the 2 + NB pieces issued since must have
        // landed before the barrier that ends group 0's MFMAs / group 1's reads of phase A
        if (wm == 1) { if (more) PP_VM(2 + NB); else PP_VM(0); }
        PP_BARRIER();
        PPM_TILE_STAMP(1);
        PP_LGKM0();
        mfmas(0);
        if (wm == 0) { if (more) PP_VM(2 + NB); else PP_VM(0); }
        PP_BARRIER();
        PPM_TILE_STAMP(2);
        // phase B: rows 64-127.  The barrier after group 0's MFMAs is the one after group 1's reads: before it both wait for all of
        // the next tile but its last two pieces.
        read_a(tile, 1);
        if (more) { stage_a(nbuf, nk0, 1); stage_a(nbuf, nk0, 3); }
        if (wm == 1) PP_VM(2);
        PP_BARRIER();
        PPM_TILE_STAMP(3);
        PP_LGKM0();
        mfmas(1);
        if (wm == 0) PP_VM(2);
        PP_BARRIER();
        PPM_TILE_STAMP(4);
    }
    if (wm == 0) PP_BARRIER();                                     // as many barriers as group 1
#undef PP_BARRIER
#undef PP_LGKM0
#undef PP_VM
    PPM_STAMP_AT(31);
    if constexpr (M16) epilogue16<8, NT>(a, acc16, smem, m0 + wm * 128, n0 + wn * 16 * NT, b);
    else epilogue<TI, TJ>(a, acc, smem, m0 + wm * 128, n0 + wn * 64, b);
    PPM_STAMP_AT(29); PPM_STAMP_RT(28);
}

// ---- the same scheme on 128-row tiles, for the layers whose 256 x 256 grid would leave CUs idle (M = 4096: SURVEY.md §8(f)) ----
//
// 128 x (128 TJ) of `out` per workgroup, 64 x (32 TJ) per wave: the whole K tile is ONE phase (8 + 4 TJ fragment reads, then 8 TJ MFMAs),
// two barriers per K tile.  With a single phase the lagging group issues tile t + 1's DMAs in the last interval before the leading
// group reads them, so the ring is three tiles deep and a phase issues tile t + 2: vmcnt(pieces of one tile) before the barrier that
// ends tile t leaves exactly tile t + 2 in flight.  The buffer written in phase t is the one read in phase t - 1, one barrier
// earlier for the other group: every wave therefore waits for its fragment reads (lgkmcnt(0)) BEFORE its barrier, not after.
//
// S: the depth of the ring (a tile issued in phase t is read in phase t + S - 1).
// (Round 4 tried S = 5 on the 128 x 128 tile on that theory — 5 x 32 KB, all of the CU's 160 KB: 15.8 / 10.5 us against 14.7 / 10.1 on the
// 1024 -> 512 / 512 -> 512 layers at M = 4096.  The phase is NOT bound by the stream's latency; see mlp_layer_pp2_kernel for what it is bound by.)
template <int TJ, int S = 3>
__global__ __launch_bounds__(512) void mlp_layer_pp1_kernel(const Args a, const int tiles_n, const int tiles_m) {
    constexpr int BM = 128, BN = 128 * TJ, BK = 64, TI = 2, NA = 2, NB = 2 * TJ, P = NA + NB, TILE = (BM + BN) * BK;
    constexpr int kOperand = S * TILE, kPatch = 8 * 64 * PATCH_LD;
    static_assert(S >= 3 && S <= 5 && (S - 2) * P <= 63 && kOperand * 2 <= 160 * 1024, "ring depth: vmcnt is a 6-bit count, the CU has 160 KB of LDS");
    __shared__ __attribute__((aligned(16))) _Float16 smem[kOperand > kPatch ? kOperand : kPatch];   // S [A | B] tiles; afterwards the epilogue patches
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_n * tiles_m, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int n0 = (id % tiles_n) * BN, m0 = (id / tiles_n) * BM, b = blockIdx.y;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* in = reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    PPM_STAMP_RT(27); PPM_STAMP_AT(26);

    const int srow = tid >> 3, kc = (lane & 7) ^ ((srow >> 1) & 7);
    const _Float16* pa[NA];
    const _Float16* pb[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) { const int ra = m0 + i * 64 + srow; pa[i] = in + (size_t)(ra < a.m ? ra : a.m - 1) * a.lda + kc * 8; }
#pragma unroll
    for (int i = 0; i < NB; i++) { const int rb = n0 + i * 64 + srow; pb[i] = W + (size_t)(rb < a.n ? rb : a.n - 1) * a.ldw + kc * 8; }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++) __builtin_amdgcn_global_load_lds((glb_ptr)(pa[i] + k0), (lds_ptr)(smem + buf * TILE + (i * 512 + wave * 64) * 8), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; i++) __builtin_amdgcn_global_load_lds((glb_ptr)(pb[i] + k0), (lds_ptr)(smem + buf * TILE + BM * BK + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };

    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    int swz[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) swz[kk] = ((2 * kk + h) ^ ((r >> 1) & 7)) * 8;
    const int arow = (wm * 64 + r) * BK, brow = BM * BK + (wn * 32 * TJ + r) * BK;
    h8 fa[TI][4], fb[TJ][4];
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); if (!(PP_EXP & 4)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
// wait until all but the `ahead` youngest TILES (P pieces each) of this wave's DMAs have landed; ahead is wave-uniform
#define PP_WAIT_TILES(ahead)                                                                                        \
    do {                                                                                                             \
        const int ah_ = (PP_EXP & 2) ? 0 : (ahead);                                                                                     \
        if (S >= 5 && ah_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * P) : "memory");                      \
        else if (S >= 4 && ah_ == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * P) : "memory");                 \
        else if (ah_ >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    } while (0)
    const int ktiles = a.k / BK;
#pragma unroll
    for (int s = 0; s < S - 1; s++) if (s < ktiles) stage(s, s * BK);
    {   // tile 0 must have landed: tiles 1 .. min(S - 2, ktiles - 1) may stay in flight
        const int ahead0 = ktiles - 1 < S - 2 ? ktiles - 1 : S - 2;
        PP_WAIT_TILES(ahead0);
    }
    PP_BARRIER();
    if (wm == 1) PP_BARRIER();                                     // group 1 runs one barrier behind
    int buf = 0;
    PPM_STAMP_AT(30); PPM_STAMP_AT(25);
    for (int t = 0; t < ktiles; t++) {
        const _Float16* tile = smem + buf * TILE;
        // after this phase's issue the wave has tiles t + 1 .. min(t + S - 1, ktiles - 1) outstanding; tile t + 1 must land before the barrier
        // that ends the phase: all but the youngest min(S - 2, ktiles - t - 2) tiles
        const int rem = ktiles - t - 2, ahead = rem < 0 ? 0 : (rem < S - 2 ? rem : S - 2);
        PPM_TILE_STAMP(0);
        if (!((PP_EXP & 1) && t > 0)) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fa[i][kk] = *reinterpret_cast<const h8*>(&tile[arow + i * 32 * BK + swz[kk]]);
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fb[j][kk] = *reinterpret_cast<const h8*>(&tile[brow + j * 32 * BK + swz[kk]]);
        }
        const int wbuf = buf == 0 ? S - 1 : buf - 1;               // (t + S - 1) % S: the buffer read in phase t - 1
        PPM_TILE_STAMP(1);
        if (t + S - 1 < ktiles && !(PP_EXP & 2)) stage(wbuf, (t + S - 1) * BK);
        PPM_TILE_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PPM_TILE_STAMP(3);
        if (wm == 1) PP_WAIT_TILES(ahead);
        PPM_TILE_STAMP(4);
        PP_BARRIER();
        PPM_TILE_STAMP(5);
        if (!(PP_EXP & 32)) __builtin_amdgcn_s_setprio(1);
        if (!((PP_EXP & 8) && t > 0)) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
        if (!(PP_EXP & 32)) __builtin_amdgcn_s_setprio(0);
        PPM_TILE_STAMP(6);
        if (wm == 0) PP_WAIT_TILES(ahead);
        PPM_TILE_STAMP(7);
        PP_BARRIER();
        PPM_TILE_STAMP(8);
        buf = buf == S - 1 ? 0 : buf + 1;
    }
    if (wm == 0) PP_BARRIER();
#undef PP_BARRIER
#undef PP_WAIT_TILES
    PPM_STAMP_AT(31);
    if (PP_EXP & 16) { if (acc[0][0][0] == 123.f) reinterpret_cast<_Float16*>(a.out)[0] = 1; return; }     // experiment: no epilogue
    epilogue<TI, TJ>(a, acc, smem, m0 + wm * 64, n0 + wn * 32 * TJ, b);
    PPM_STAMP_AT(29); PPM_STAMP_RT(28);
}

// ---- the ring kernel with the DMA issue moved out of the read phase (round 4) ----
//
// tools/gpu_mlp_exp1.py (builds with one ingredient of the K loop removed, profiles/r04_b_mlp_exp1.txt): at M = 4096 the K loop of
// mlp_layer_pp1_kernel runs at half the matrix cores' rate — a barrier interval lasts as long as the READING group's phase (fragment reads,
// its share of the DMA issue at 100-185 cycles a piece, the waits), 800-1000 cycles against 256 TJ cycles of MFMAs in the other group.
// Here the LEADING group (wm = 0) issues every piece of tile t + 2 between the MFMAs of tile t, where a piece costs ~60 cycles and the
// matrix cores keep running; the read phases of both groups are fragment reads only.  Only the leading group has DMAs outstanding, so only
// it waits (vmcnt(2P): all but tile t + 2) before the barrier that releases tile t + 1 — to itself at once, to the lagging group one barrier
// later.  Tile t + 2 overwrites the buffer of tile t - 1, which the lagging group finished reading (lgkmcnt(0)) before its first barrier of
// iteration t - 1 = the leading group's second barrier of t - 1; the leading group issues after its first barrier of t.
template <int TJ> struct PP3 {
    static constexpr int BM = 128, BN = 128 * TJ, BK = 64, TI = 2, NA = 2, NB = 2 * TJ, P2 = 2 * (NA + NB), TILE = (BM + BN) * BK;   // P2: pieces per LEADING wave and tile
    static constexpr int kOperand = 3 * TILE, kPatch = 8 * 64 * PATCH_LD, kSmem = kOperand > kPatch ? kOperand : kPatch;            // three [A | B] tiles; afterwards the epilogue patches
    static_assert(P2 % 4 == 0 && P2 <= 63, "a quarter of the pieces after each K sub-step");
};
// one 128 x BN tile of `out` (problem b, origin (m0, n0)) by the calling workgroup; smem: PP3<TJ>::kSmem fp16, free on entry (every wave past a barrier after its last use)
template <int TJ>
__device__ __forceinline__ void pp3_tile(const Args& a, _Float16* smem, const int m0, const int n0, const int b) {
    constexpr int BM = PP3<TJ>::BM, BK = PP3<TJ>::BK, TI = PP3<TJ>::TI, NA = PP3<TJ>::NA, NB = PP3<TJ>::NB, P2 = PP3<TJ>::P2, TILE = PP3<TJ>::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* in = reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    PPM_STAMP_RT(27); PPM_STAMP_AT(26);

    // staging by the four leading waves: piece (i, hf) of an operand = rows 64 i + 32 hf + 8 wn .. + 7 (one 1 KiB piece per wave), thread ->
    // row srow = (tid & 255) >> 3 of the 32, 16-byte slot tid & 7 holding k-chunk slot ^ ((row >> 1) & 7) (32 hf and 64 i do not change that term)
    const int srow = (tid & 255) >> 3, kc = (lane & 7) ^ ((srow >> 1) & 7);
    unsigned offa[NA][2], offb[NB][2];                             // element offsets from `in` / W: 32-bit, the bases stay scalar
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++) { const int ra = m0 + i * 64 + hf * 32 + srow; offa[i][hf] = (unsigned)(ra < a.m ? ra : a.m - 1) * (unsigned)a.lda + kc * 8; }
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++) { const int rb = n0 + i * 64 + hf * 32 + srow; offb[i][hf] = (unsigned)(rb < a.n ? rb : a.n - 1) * (unsigned)a.ldw + kc * 8; }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // piece q of a tile, q = 0 .. P2 - 1: A pieces first (what a phase reads first), then B
    auto piece = [&](int buf, int k0, int q) {
        if (q < 2 * NA) __builtin_amdgcn_global_load_lds((glb_ptr)(in + offa[q >> 1][q & 1] + k0), (lds_ptr)(smem + buf * TILE + ((q >> 1) * 64 + (q & 1) * 32 + wn * 8) * BK), 16, 0, 0);
        else { const int qb = q - 2 * NA; __builtin_amdgcn_global_load_lds((glb_ptr)(W + offb[qb >> 1][qb & 1] + k0), (lds_ptr)(smem + buf * TILE + BM * BK + ((qb >> 1) * 64 + (qb & 1) * 32 + wn * 8) * BK), 16, 0, 0); }
    };

    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    int swz[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) swz[kk] = ((2 * kk + h) ^ ((r >> 1) & 7)) * 8;
    const int arow = (wm * 64 + r) * BK, brow = BM * BK + (wn * 32 * TJ + r) * BK;
    h8 fa[TI][4], fb[TJ][4];
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    const int ktiles = a.k / BK;
    if (wm == 0) {
#pragma unroll
        for (int q = 0; q < P2; q++) piece(0, 0, q);
        if (ktiles > 1) {
#pragma unroll
            for (int q = 0; q < P2; q++) piece(1, BK, q);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P2) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
    if (wm == 1) PP_BARRIER();                                     // group 1 runs one barrier behind
    int buf = 0;
    PPM_STAMP_AT(30); PPM_STAMP_AT(25);
    for (int t = 0; t < ktiles; t++) {
        const _Float16* tile = smem + buf * TILE;
        const bool more = t + 2 < ktiles;
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fa[i][kk] = *reinterpret_cast<const h8*>(&tile[arow + i * 32 * BK + swz[kk]]);
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fb[j][kk] = *reinterpret_cast<const h8*>(&tile[brow + j * 32 * BK + swz[kk]]);
        const int wbuf = buf == 0 ? 2 : buf - 1;                   // (t + 2) % 3: the buffer of tile t - 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
            if (wm == 0 && more) {
#pragma unroll
                for (int q = 0; q < P2 / 4; q++) piece(wbuf, (t + 2) * BK, kk * (P2 / 4) + q);
            }
        }
        if (wm == 0) { if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P2) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        PP_BARRIER();
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (wm == 0) PP_BARRIER();
#undef PP_BARRIER
    PPM_STAMP_AT(31);
    epilogue<TI, TJ>(a, acc, smem, m0 + wm * 64, n0 + wn * 32 * TJ, b);
    PPM_STAMP_AT(29); PPM_STAMP_RT(28);
}
template <int TJ>
__global__ __launch_bounds__(512) void mlp_layer_pp3_kernel(const Args a, const int tiles_n, const int tiles_m) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[PP3<TJ>::kSmem];
    const int nwg = tiles_n * tiles_m, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;      // consecutive workgroup ids go round the XCDs: a contiguous run of tiles each
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    pp3_tile<TJ>(a, smem, (id / tiles_n) * PP3<TJ>::BM, (id % tiles_n) * PP3<TJ>::BN, blockIdx.y);
}

// ---- consecutive layers in ONE launch (round 4): persistent workgroups take tiles by ticket, a tile waits for its rows of the layer below ----
//
// A layer launch at M = 4096 is one round of workgroups that load, multiply and store in step: per launch ~3 us of dispatch and first round trip
// and 2.5-6 us of dirty-line write-back sit outside the K loops (tools/gpu_mlp_phases.py), four to seven times per forward.  Here up to four
// consecutive hidden layers (all on the 128-row ring tiles) are the tiles of one launch: gridDim.x workgroups (one per CU) draw tickets from a
// counter; ticket order is layer by layer, and within a layer row panel by row panel, so the tile a workgroup draws depends only on tiles with
// LOWER tickets — every one of which some running workgroup has already drawn and will finish without waiting on anything higher: no ordering
// or co-residency of workgroups is assumed.  A tile of layer l > 0 for the 128-row panel p of problem b waits until all tiles_n[l - 1] tiles of
// (l - 1, b, p) have been counted; a finished tile is counted after every thread's agent-scope fence behind its stores and a barrier (release),
// the waiting lane reads the counter with acquire semantics (which also drops the CU's stale L1 / L2 lines of the activation buffer, rewritten
// every forward) and the barrier behind it releases the other waves.  Every wait is bounded: a timeout sets word 2 of the workspace and goes on.
// The last workgroup to finish zeroes the counters for the next launch.
struct ChainArgs {
    Args a[4];
    int count, batch, tiles_m, debug;    // debug (PPENV_CHAIN_DEBUG, diagnostic): 1 no tiles (tickets and counters only), 2 no waits
    int tj[4], tiles_n[4], first[5];      // per layer: 128-column units per tile (1 | 2), tiles along N, first ticket (first[count] = all tickets)
    unsigned* sync;                      // [0] ticket, [1] workgroups finished, [2] error flag (sticky), [3 + (l * batch + b) * tiles_m + p] tiles done
};
__global__ __launch_bounds__(512) void mlp_chain_pp3_kernel(const ChainArgs c) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[PP3<2>::kSmem];
    __shared__ unsigned s_word;
    const int tid = threadIdx.x;
    // one value from thread 0 to the whole workgroup, as a SCALAR (the loop below and every branch in it are then uniform for the compiler too: with the
    // ticket in a vector register it nested the barrier inside per-lane exit masks).  Also the barrier that frees smem between two tiles.
    auto broadcast = [&](unsigned v) -> int {
        if (tid == 0) s_word = v;
        __syncthreads();
        const int r = __builtin_amdgcn_readfirstlane((int)s_word);
        __syncthreads();                                            // s_word may be rewritten after this
        return r;
    };
    auto ticket = [&]() -> int {
        unsigned v = 0;
        if (tid == 0) v = __hip_atomic_fetch_add(&c.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return broadcast(v);
    };
    const int total = c.first[c.count];
    for (int t = ticket(); t < total; t = ticket()) {
        int l = 0;
        while (l + 1 < c.count && t >= c.first[l + 1]) l++;
        const int idx = t - c.first[l], tn = c.tiles_n[l];
        const int j = idx % tn, pb = idx / tn, b = pb % c.batch, p = pb / c.batch;      // row panel by row panel, both problems of a panel side by side
        if (l > 0 && !(c.debug & 2)) {
            if (tid == 0) {
                const unsigned* cnt = &c.sync[3 + ((l - 1) * c.batch + b) * c.tiles_m + p];
                const unsigned need = (unsigned)c.tiles_n[l - 1];
                // relaxed polls (an acquire per poll would invalidate the XCD's L2 under the working tiles), ONE acquire fence once the count is there.
                // Bounded: ~0.1 s, and after the first timeout anywhere (word 2) nobody waits any more — the launch ends, its results are void.
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(32);
                    if ((++spins & 63u) == 0 && (spins > (1u << 16) || __hip_atomic_load(&c.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(&c.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        if (c.debug & 1) { }
        else if (c.tj[l] == 2) pp3_tile<2>(c.a[l], smem, p * 128, j * 256, b);
        else pp3_tile<1>(c.a[l], smem, p * 128, j * 128, b);
        if (l + 1 < c.count) {
            __syncthreads();                                       // every wave's stores of the tile are performed (workgroup-scope release: vmcnt(0)) ...
            if (tid == 0) __hip_atomic_fetch_add(&c.sync[3 + (l * c.batch + b) * c.tiles_m + p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... and ONE write-back of the L2 publishes them with the count
        }
    }
    unsigned last = 0;
    if (tid == 0) last = __hip_atomic_fetch_add(&c.sync[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    if (broadcast(last)) {                                         // nobody else touches the workspace any more
        const int words = 3 + (c.count - 1) * c.batch * c.tiles_m;
        for (int i = tid; i < words; i += 512) if (i != 2) __hip_atomic_store(&c.sync[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- K tiles taken in turn by the two wave groups (round 4): the narrow layers at the rollout's M = 4096 ----
//
// In mlp_layer_pp1_kernel both groups work on every K tile, each on its half of the workgroup's rows: per K tile a wave reads 8 + 4 TJ
// fragments for 8 TJ MFMAs and meets two barriers, and a barrier interval lasts as long as the READING group needs (fragment reads, its
// share of the DMA issue, the wait) — 800-1000 cycles against 256 TJ of MFMAs: the K tile of the 128 x 128 kernel takes 1600-2100 cycles for
// 512 cycles of matrix work per SIMD (tools/gpu_mlp_phases.py; a deeper ring changed nothing — it is not the stream's latency).  Here
// group t & 1 owns K tile t entirely: 4 waves as 2 x 2 over the 128 x BN tile, 64 x (32 TJW) per wave, twice the MFMAs per fragment read
// (8 + 4 TJW reads for 8 TJW MFMAs) and ONE barrier per K tile — while one group multiplies tile t the other reads tile t + 1.  Each group
// ends with a partial sum over its K tiles; they swap halves through LDS (group 0 finishes the left half of every wave tile, group 1 the
// right half: sum = group 0's part + group 1's part on both sides) and all eight waves run the epilogue.
//
// Ring of R buffers, tile t in buffer t % R.  During interval t tile t is in its group's registers (read in interval t - 1, behind a barrier),
// so its buffer takes tile t + R; tile t + 1 is being read, tiles t + 2 .. t + R are in flight or landed, and before the barrier that ends
// the interval every wave waits for its pieces of tile t + 2: all but the youngest R - 2 tiles.
template <int TJW, int R>
__global__ __launch_bounds__(512) void mlp_layer_pp2_kernel(const Args a, const int tiles_n, const int tiles_m) {
    constexpr int BM = 128, BN = 64 * TJW, BK = 64, TI = 2, TJH = TJW / 2, NA = 2, NB = BN / 64, P = NA + NB, TILE = (BM + BN) * BK;
    constexpr int kOperand = R * TILE, kPatch = 8 * 64 * PATCH_LD, kSwap = 8 * TI * TJH * 16 * 64 * 2;   // in fp16 units (the swap area holds floats)
    static_assert(TJW == 2 || TJW == 4, "wave tile 64 x 64 or 64 x 128");
    static_assert(R >= 3 && R <= 5 && (R - 2) * P <= 63 && kOperand * 2 <= 160 * 1024 && kSwap <= kOperand, "ring depth / LDS");
    __shared__ __attribute__((aligned(16))) _Float16 smem[kOperand > kPatch ? kOperand : kPatch];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int nwg = tiles_n * tiles_m, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int n0 = (id % tiles_n) * BN, m0 = (id / tiles_n) * BM, b = blockIdx.y;
    const _Float16* W = a.w + (size_t)b * a.w_stride;
    const _Float16* in = reinterpret_cast<const _Float16*>(a.in) + (size_t)b * a.in_stride;
    PPM_STAMP_RT(27); PPM_STAMP_AT(26);

    const int srow = tid >> 3, kc = (lane & 7) ^ ((srow >> 1) & 7);
    const _Float16* pa[NA];
    const _Float16* pb[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) { const int ra = m0 + i * 64 + srow; pa[i] = in + (size_t)(ra < a.m ? ra : a.m - 1) * a.lda + kc * 8; }
#pragma unroll
    for (int i = 0; i < NB; i++) { const int rb = n0 + i * 64 + srow; pb[i] = W + (size_t)(rb < a.n ? rb : a.n - 1) * a.ldw + kc * 8; }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++) __builtin_amdgcn_global_load_lds((glb_ptr)(pa[i] + k0), (lds_ptr)(smem + buf * TILE + (i * 512 + wave * 64) * 8), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; i++) __builtin_amdgcn_global_load_lds((glb_ptr)(pb[i] + k0), (lds_ptr)(smem + buf * TILE + BM * BK + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };

    f16v acc[TI][TJW];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJW; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    int swz[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) swz[kk] = ((2 * kk + h) ^ ((r >> 1) & 7)) * 8;
    const int arow = (wm * 64 + r) * BK, brow = BM * BK + (wn * 32 * TJW + r) * BK;
    h8 fa[TI][4], fb[TJW][4];
    auto read_frags = [&](const _Float16* tile) {
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fa[i][kk] = *reinterpret_cast<const h8*>(&tile[arow + i * 32 * BK + swz[kk]]);
#pragma unroll
        for (int j = 0; j < TJW; j++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) fb[j][kk] = *reinterpret_cast<const h8*>(&tile[brow + j * 32 * BK + swz[kk]]);
    };
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
// wait until all but the `ahead` youngest TILES (P pieces each) of this wave's DMAs have landed; ahead is wave-uniform, 0 .. R - 2
#define PP_WAIT_TILES(ahead)                                                                                        \
    do {                                                                                                             \
        const int ah_ = (ahead);                                                                                     \
        if (R >= 5 && ah_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * P) : "memory");                      \
        else if (R >= 4 && ah_ == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * P) : "memory");                 \
        else if (ah_ >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    } while (0)
    const int ktiles = a.k / BK;
#pragma unroll
    for (int s = 0; s < R; s++) if (s < ktiles) stage(s, s * BK);
    {   // tiles 0 and 1 must have landed (group 0 reads tile 0 now, group 1 reads tile 1 in interval 0): tiles 2 .. min(R, ktiles) - 1 may fly on
        const int staged = ktiles < R ? ktiles : R, ahead0 = staged > 2 ? staged - 2 : 0;
        PP_WAIT_TILES(ahead0);
    }
    PP_BARRIER();
    if (grp == 0) { read_frags(smem); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    PP_BARRIER();                                                  // buffer 0 is free from here on
    PPM_STAMP_AT(30); PPM_STAMP_AT(25);
    int buf = 0;                                                   // t % R
    for (int t = 0; t < ktiles; t++) {
        const int nbuf = buf == R - 1 ? 0 : buf + 1;               // (t + 1) % R
        if (t + R < ktiles) stage(buf, (t + R) * BK);
        if (grp == (t & 1)) {
            if (!(PP_EXP & 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int i = 0; i < TI; i++)
#pragma unroll
                    for (int j = 0; j < TJW; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
            if (!(PP_EXP & 32)) __builtin_amdgcn_s_setprio(0);
        } else if (t + 1 < ktiles) {
            read_frags(smem + nbuf * TILE);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // before the barrier: the next interval's DMA overwrites this buffer
        }
        // tile t + 2 is read in the next interval: of this wave's tiles t + 2 .. min(t + R, ktiles - 1) all but the first may stay in flight
        const int rem = ktiles - t - 3, ahead = rem < 0 ? 0 : (rem < R - 2 ? rem : R - 2);
        PP_WAIT_TILES(ahead);
        PP_BARRIER();
        buf = nbuf;
    }
#undef PP_WAIT_TILES
    PPM_STAMP_AT(31);
    // swap halves: a wave keeps columns [32 TJH g, 32 TJH (g + 1)) of its 64 x (32 TJW) tile and sends the other half to wave ^ 4 (same tile, other group)
    f4v* swap = reinterpret_cast<f4v*>(smem);
    auto send = [&](auto G) {                                      // G: this wave's group as a compile-time constant (register arrays need constant indices)
        constexpr int g = decltype(G)::value;
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJH; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f16v& src = acc[i][(1 - g) * TJH + j];
                    swap[(((wave * TI + i) * TJH + j) * 4 + q) * 64 + lane] = f4v{src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]};
                }
    };
    if (grp == 0) send(std::integral_constant<int, 0>{}); else send(std::integral_constant<int, 1>{});
    PP_BARRIER();
#undef PP_BARRIER
    f16v fin[TI][TJH];
    auto take = [&](auto G) {
        constexpr int g = decltype(G)::value;
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJH; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f4v got = swap[((((wave ^ 4) * TI + i) * TJH + j) * 4 + q) * 64 + lane];
                    const f16v& own = acc[i][g * TJH + j];
#pragma unroll
                    for (int e = 0; e < 4; e++) fin[i][j][4 * q + e] = g == 0 ? own[4 * q + e] + got[e] : got[e] + own[4 * q + e];   // group 0's part first on both sides
                }
    };
    if (grp == 0) take(std::integral_constant<int, 0>{}); else take(std::integral_constant<int, 1>{});
    epilogue<TI, TJH>(a, fin, smem, m0 + wm * 64, n0 + wn * 32 * TJW + grp * 32 * TJH, b);
    PPM_STAMP_AT(29); PPM_STAMP_RT(28);
}

// obs [m, k] fp32 -> out [m, ld_out] fp16, normalised and clamped, zero beyond k.  One thread per two output columns: every load is
// unconditional (clamped index), so a thread has one round trip to memory, not one per element.
__global__ __launch_bounds__(256) void prepare_input_kernel(const float* __restrict__ obs, int m, int k, int ld_obs, const float* __restrict__ mean,
                                                            const float* __restrict__ inv_std, float clip, _Float16* __restrict__ out, int ld_out) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int ppr = ld_out / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)m * ppr) return;
    const int row = (int)(idx / ppr), c0 = (int)(idx % ppr) * 2;
    const int i0 = c0 < k ? c0 : k - 1, i1 = c0 + 1 < k ? c0 + 1 : k - 1;
    float g0 = obs[(size_t)row * ld_obs + i0], g1 = obs[(size_t)row * ld_obs + i1];
    if (mean) {
        g0 = fminf(fmaxf((g0 - mean[i0]) * inv_std[i0], -clip), clip);
        g1 = fminf(fmaxf((g1 - mean[i1]) * inv_std[i1], -clip), clip);
    }
    const h2 v = {(_Float16)(c0 < k ? g0 : 0.f), (_Float16)(c0 + 1 < k ? g1 : 0.f)};
    *reinterpret_cast<h2*>(out + (size_t)row * ld_out + c0) = v;
}

// 32 lanes per row (lane j draws actions j, j + 32, ...), eight rows per workgroup: the draws of a row are independent; its negative
// log-probability is a 32-lane butterfly sum
__global__ __launch_bounds__(256) void sample_actions_kernel(const float* __restrict__ mu, int m, int a, int ld_mu, const float* __restrict__ sigma,
                                                             unsigned long long seed, unsigned long long counter, float lo, float hi,
                                                             float* __restrict__ actions, float* __restrict__ neglogp) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), j0 = threadIdx.x & 31;
    const bool live = row < m;
    float nl = 0.f;
    if (live)
        for (int j = j0; j < a; j += 32) {
            const float sg = sigma[j];
            const float g = pp::dr_gauss(seed, (uint32_t)row, (uint32_t)(counter >> 24), (uint32_t)(counter & 0xFFFFFFu), (uint32_t)j);
            float x = __fmaf_rn(sg, g, mu[(size_t)row * ld_mu + j]);
            nl += __fmaf_rn(0.5f * g, g, logf(sg));   // spelled out: both samplers must round alike
            if (lo < hi) x = fminf(fmaxf(x, lo), hi);
            actions[(size_t)row * a + j] = x;
        }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) nl += __shfl_xor(nl, d, 32);
    if (live && j0 == 0 && neglogp) neglogp[row] = __fmaf_rn(0.9189385332f, (float)a, nl);     // + 0.5 log(2 pi) per action
}

// GAE backwards over the horizon, one thread per env (values: element (t, i) at values[t * values_step + i * ld_values])
__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values, int ld_values, long long values_step,
                                                  const long long* __restrict__ dones, int horizon, int n, float gamma, float tau, float scale,
                                                  float* __restrict__ adv, float* __restrict__ ret) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float next_v = values[(size_t)horizon * values_step + (size_t)i * ld_values], run = 0.f;
    for (int t = horizon - 1; t >= 0; t--) {
        const float nd = dones[(size_t)t * n + i] ? 0.f : 1.f;
        const float v = values[(size_t)t * values_step + (size_t)i * ld_values];
        const float delta = scale * rewards[(size_t)t * n + i] + gamma * next_v * nd - v;
        run = delta + gamma * tau * nd * run;
        adv[(size_t)t * n + i] = run;
        ret[(size_t)t * n + i] = run + v;
        next_v = v;
    }
}

// The heads: out[m, n <= 32] (fp32) = in[m, k] * W[n, k]^T + bias — a skinny layer that is all input traffic (8 MB of features for
// 0.2 GFLOP).  A workgroup of four waves owns 32 rows; wave w takes the K steps w, w + 4, ... with both MFMA operands loaded straight
// from global memory as fragments (16 bytes per lane; W is 56 KB and stays in L2), and the four partial 32 x 32 tiles are summed
// through LDS.  Rows of W beyond n are clamped on load and never stored.
// sampling fused into the heads (ppenv_mlp_heads_sample): what sample_actions_kernel does, on the row the 32-lane group has just finished
struct SampleArgs {
    int num_actions;                 // the first num_actions columns of `out` are mu
    const float* sigma;
    unsigned long long seed, counter;
    float lo, hi;
    float* actions;                  // [m, num_actions]; NULL: no sampling (plain ppenv_mlp_layer_forward)
    float* neglogp;                  // [m] or NULL
};
__global__ __launch_bounds__(256) void mlp_heads_kernel(const Args a, const SampleArgs sa) {
    __shared__ float part[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32;
    const int row = m0 + r < a.m ? m0 + r : a.m - 1, wr = r < a.n ? r : a.n - 1;
    const _Float16* pa = reinterpret_cast<const _Float16*>(a.in) + (size_t)row * a.lda + h * 8;
    const _Float16* pw = a.w + (size_t)wr * a.ldw + h * 8;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const int ksteps = a.k / 16;
    int ks = wave;
    // sixteen K steps of this wave per trip (k = 1024, the stacked features of the reference's network: the whole row at once) — the launch is
    // one round trip to memory per trip, and with four steps per trip it was four of them (8.8 us at M = 4096 for 8 MB of features)
    for (; ks + 60 < ksteps; ks += 64) {
        h8 fa[16], fb[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { fa[u] = *reinterpret_cast<const h8*>(pa + (ks + 4 * u) * 16); fb[u] = *reinterpret_cast<const h8*>(pw + (ks + 4 * u) * 16); }
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u], fb[u], acc, 0, 0, 0);
    }
    for (; ks + 12 < ksteps; ks += 16) {                         // four K steps of this wave per trip: eight loads in flight
        h8 fa[4], fb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { fa[u] = *reinterpret_cast<const h8*>(pa + (ks + 4 * u) * 16); fb[u] = *reinterpret_cast<const h8*>(pw + (ks + 4 * u) * 16); }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u], fb[u], acc, 0, 0, 0);
    }
    for (; ks < ksteps; ks += 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(pa + ks * 16), *reinterpret_cast<const h8*>(pw + ks * 16), acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; reg++) part[wave][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] = acc[reg];
    __syncthreads();
    float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int e = 0; e < 4; e++) {                                 // a 32-lane group holds one row of the tile
        const int idx = tid + 256 * e, orow = idx >> 5, col = idx & 31, row = m0 + orow;
        const bool live = row < a.m && col < a.n;
        float x = 0.f;
        if (live) {
            x = part[0][orow][col] + part[1][orow][col] + part[2][orow][col] + part[3][orow][col] + (a.bias ? (float)a.bias[col] : 0.f);
            if (a.elu) x = x > 0.f ? x : __expf(x) - 1.0f;
            out[(size_t)row * a.ldo + col] = x;
        }
        if (sa.actions) {                                         // same draws, clamp and log-probability as sample_actions_kernel
            float nl = 0.f;
            if (live && col < sa.num_actions) {
                const float sg = sa.sigma[col];
                const float g = pp::dr_gauss(sa.seed, (uint32_t)row, (uint32_t)(sa.counter >> 24), (uint32_t)(sa.counter & 0xFFFFFFu), (uint32_t)col);
                float v = __fmaf_rn(sg, g, x);
                nl = __fmaf_rn(0.5f * g, g, logf(sg));    // as sample_actions_kernel
                if (sa.lo < sa.hi) v = fminf(fmaxf(v, sa.lo), sa.hi);
                sa.actions[(size_t)row * sa.num_actions + col] = v;
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) nl += __shfl_xor(nl, d, 32);
            if (row < a.m && col == 0 && sa.neglogp) sa.neglogp[row] = __fmaf_rn(0.9189385332f, (float)sa.num_actions, nl);
        }
    }
}
}  // namespace

#if defined(PPM_STAMP)
extern "C" int ppenv_mlp_debug_read_stamps(unsigned long long* dst, size_t count) {
    if (hipDeviceSynchronize() != hipSuccess) return PPENV_EHIP;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mlp_stamp_buf), count * sizeof(unsigned long long)) == hipSuccess ? 0 : PPENV_EHIP;
}
#endif

extern "C" int ppenv_mlp_prepare_input(const float* obs, int32_t m, int32_t k, int32_t ld_obs, const float* mean, const float* inv_std, float clip,
                                       void* out, int32_t ld_out, void* stream) {
    if (!obs || !out || m <= 0 || k <= 0 || ld_obs < k || ld_out < k || (ld_out & 7) || (reinterpret_cast<uintptr_t>(out) & 15) || ((mean == nullptr) != (inv_std == nullptr))) {
        ppenv_set_error("ppenv_mlp_prepare_input: NULL pointer or inconsistent sizes (need ld_obs >= k, ld_out >= k and a multiple of 8, out 16-byte aligned, mean and inv_std together)");
        return PPENV_EINVAL;
    }
    const long long chunks = (long long)m * (ld_out / 2);
    hipLaunchKernelGGL(prepare_input_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, m, k, ld_obs, mean, inv_std, clip,
                       reinterpret_cast<_Float16*>(out), ld_out);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching prepare_input_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_mlp_sample_actions(const float* mu, int32_t m, int32_t a, int32_t ld_mu, const float* sigma, uint64_t seed, uint64_t counter,
                                        float lo, float hi, float* actions, float* neglogp, void* stream) {
    if (!mu || !sigma || !actions || m <= 0 || a <= 0 || a > 256 || ld_mu < a) {
        ppenv_set_error("ppenv_mlp_sample_actions: NULL pointer or inconsistent sizes (need 0 < a <= 256, ld_mu >= a)");
        return PPENV_EINVAL;
    }
    hipLaunchKernelGGL(sample_actions_kernel, dim3((m + 7) / 8), dim3(256), 0, (hipStream_t)stream, mu, m, a, ld_mu, sigma,
                       (unsigned long long)seed, (unsigned long long)counter, lo, hi, actions, neglogp);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching sample_actions_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_gae(const float* rewards, const float* values, int32_t ld_values, int64_t values_step, const int64_t* dones, int32_t horizon, int32_t n,
                         float gamma, float tau, float reward_scale, float* advantages, float* returns, void* stream) {
    if (!rewards || !values || !dones || !advantages || !returns || horizon <= 0 || n <= 0 || ld_values <= 0 || values_step <= 0) {
        ppenv_set_error("ppenv_gae: NULL pointer or non-positive size");
        return PPENV_EINVAL;
    }
    hipLaunchKernelGGL(gae_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, values, ld_values, (long long)values_step,
                       reinterpret_cast<const long long*>(dones), horizon, n, gamma, tau, reward_scale, advantages, returns);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching gae_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

// the heads layer and the action draw in one launch
extern "C" int ppenv_mlp_heads_sample(const ppenv_mlp_layer* L, int32_t num_actions, const float* sigma, uint64_t seed, uint64_t counter, float lo, float hi,
                                      float* actions, float* neglogp, void* stream) {
    const bool ok = L && L->in && L->w && L->out && actions && sigma && L->m > 0 && L->k > 0 && L->n > 0 && L->n <= 32 && num_actions > 0 && num_actions <= L->n &&
                    L->batch == 1 && L->out_f32 && !L->in_f32 && L->k % 16 == 0 && L->lda >= L->k && L->ldw >= L->k && L->ldo >= L->n && L->lda % 8 == 0 && L->ldw % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(L->in) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->w) & 15) == 0;
    if (!ok) {
        ppenv_set_error("ppenv_mlp_heads_sample: needs a heads layer the skinny kernel takes (fp16 input, fp32 output, batch 1, n <= 32, k % 16 == 0, "
                        "16-byte aligned rows) and 0 < num_actions <= n");
        return PPENV_EINVAL;
    }
    Args a{L->m, L->n, L->k, L->lda, L->ldw, L->ldo, L->elu, L->out_f32, L->in, (long long)L->in_stride, L->mean, L->inv_std, L->clip,
           reinterpret_cast<const _Float16*>(L->w), (long long)L->w_stride, reinterpret_cast<const _Float16*>(L->bias), (long long)L->bias_stride,
           L->out, (long long)L->out_stride};
    hipLaunchKernelGGL(mlp_heads_kernel, dim3((L->m + 31) / 32), dim3(256), 0, (hipStream_t)stream, a,
                       SampleArgs{num_actions, sigma, (unsigned long long)seed, (unsigned long long)counter, lo, hi, actions, neglogp});
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_heads_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

namespace {
struct BwdInput {                  // what ppenv_mlp_layer_backward_input adds to a layer launch (Args::aux / colsum)
    const _Float16* aux; long long aux_stride; int ldaux;
    float* colsum; long long colsum_stride; int ldcs;
};
int launch_layer(const ppenv_mlp_layer* L, const BwdInput* bw, void* stream, int cus = 0) {
    if (cus <= 0) cus = 256;                     // the CUs this launch should fill: 256 = the chip; fewer when another launch sequence runs beside it (ppenv_mlp_layer_forward_share)
    if (!L || !L->in || !L->w || !L->out || L->m <= 0 || L->n <= 0 || L->k <= 0 || L->batch <= 0 || L->lda < L->k || L->ldw < L->k || L->ldo < L->n) {
        ppenv_set_error("ppenv_mlp_layer_forward: NULL pointer or inconsistent sizes (need lda >= k, ldw >= k, ldo >= n)");
        return PPENV_EINVAL;
    }
    if (L->in_f32 && ((L->mean == nullptr) != (L->inv_std == nullptr))) { ppenv_set_error("ppenv_mlp_layer_forward: mean and inv_std go together"); return PPENV_EINVAL; }
    Args a{L->m, L->n, L->k, L->lda, L->ldw, L->ldo, L->elu, L->out_f32, L->in, (long long)L->in_stride, L->mean, L->inv_std, L->clip,
           reinterpret_cast<const _Float16*>(L->w), (long long)L->w_stride, reinterpret_cast<const _Float16*>(L->bias), (long long)L->bias_stride,
           L->out, (long long)L->out_stride};
    if (bw) { a.aux = bw->aux; a.aux_stride = bw->aux_stride; a.ldaux = bw->ldaux; a.colsum = bw->colsum; a.colsum_stride = bw->colsum_stride; a.ldcs = bw->ldcs; }
    // Tile choice.  Measured on the reference's layers (tools/gpu_mlp_layers.py, M = 16384, TFLOP/s on the 2048 -> 1536 layer):
    //   128 x 128, 4 waves of 64 x 64, BK 64, register staging (two workgroups per CU)            600
    //   256 x 128, 4 waves of 128 x 64 (one workgroup of four waves per CU)                        507   too few waves to hide anything
    //   256 x 256, 4 waves of 128 x 128                                                            157   512 registers and still spilling
    //   256 x 256, 8 waves of 128 x 64, BK 64, register staging (one workgroup of 8 waves per CU)  727
    //   256 x 256, 8 waves in two alternating groups, LDS-DMA staging (mlp_layer_pp_kernel)         910-960   (hipBLASLt + a separate ELU: 880)
    // and at M = 4096, where the 256 x 256 grid of the narrower layers covers half the chip or less (us per layer, 256^2 / 128 x 256 / 128^2):
    //   1536 -> 1024: 47.6 / 33.3 / 38.3      1024 -> 1024: 33.9 / 24.5 / 27.2      1024 -> 512: 30.5 / 20.1 / 14.6      512 -> 512: 19.9 / 13.6 / 10.0
    // The 256 x 256 tile on v_mfma_f32_16x16x32_f16 instead of 32x32x16 (516): 986 / 979 / 855 TF against 924 / 936 / 832 on the K = 2048 /
    // 1536 / 1024 layers at M = 16384 (the chip runs these kernels at its power limit — the shader clock sags from 1.95 to 1.6 GHz as the
    // K loop gets denser — and the smaller instruction costs less per flop), but a slower epilogue: only where K >= 1024 amortises it.
    // so: among the LDS-DMA kernels the largest tile that still gives three quarters of the CUs a workgroup, else the smallest; the
    // register-staged kernels for what those cannot take (fp32 input, ragged K, unaligned rows, the narrow heads).
    // 517: the 16x16x32 kernel on 256 x 192 tiles, where the 256 x 256 grid is a single partial round that 192-column tiles fill
    // (2048 -> 1536 at M = 4096: 192 -> 256 workgroups).
    // PPENV_MLP_TILE = 128 | 129 | 384 | 385 | 512 | 513 | 514 | 515 | 516 | 517 | 518 | 520 | 521 forces one (518: K tiles taken in turn by the wave
    // groups, 64 x 64 per wave — 19.5 us against 15.4 on the 1024 -> 512 layer at M = 4096; its 64 x 128 form needs 256 + 153 registers).
    const char* env = getenv("PPENV_MLP_TILE");   // read per call: the tests switch it inside one process
    const int forced = env ? atoi(env) : 0;
    auto wgs = [&](int bm, int bn) { return (long long)((L->n + bn - 1) / bn) * ((L->m + bm - 1) / bm) * L->batch; };
    int cfg = forced;
    if (cfg == 0) {
        // the first layer (fp32 observations, K = 80 or 313) converts its obs tile once per column tile: wide tiles and a K step of 32
        // (less zero padding of K) — 54 us against 86 (256 x 256 / BK 32 vs 128 x 128 / BK 64, M = 4096, K = 313)
        if (L->in_f32) cfg = (wgs(256, 256) >= cus / 2 && L->n >= 256) ? 385 : 129;
        else if (L->n <= 32 && L->out_f32 && L->batch == 1 && L->k % 16 == 0 && L->lda % 8 == 0 && L->ldw % 8 == 0 &&
                 (reinterpret_cast<uintptr_t>(L->in) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->w) & 15) == 0) cfg = 600;   // the heads
        else if (L->n >= 128) {                                                                        // falls back below when the operands do not qualify
            const long long w256 = wgs(256, 256), w192 = wgs(256, 192);
            // rounds of workgroups x tile area: a launch of 1.5 rounds takes as long as one of 2 (2048 -> 1536 at M = 8192, the yaml's minibatch: 384
            // tiles of 256 x 256 against 512 of 256 x 192 = two FULL rounds at 0.75 the area, 3 % taken off for the narrower tile's efficiency)
            const long long r256 = (w256 + cus - 1) / cus, r192 = (w192 + cus - 1) / cus;
            const bool fits192 = L->n % 192 == 0 && !bw;           // (the 192-column kernel has no backward store pass)
            if (w256 >= 3 * cus / 4) cfg = (fits192 && w192 > w256 && 0.77 * (double)r192 < (double)r256) ? 517 : (L->k >= 1024 ? 516 : 512);   // (one partial round at M = 4096: 192-column tiles fill it)
            else {
                static const bool old_ring = getenv("PPENV_MLP_RING") && atoi(getenv("PPENV_MLP_RING")) == 1;     // A/B switch: 1 = the round-2 ring kernels
                cfg = wgs(128, 256) >= 3 * cus / 4 ? (old_ring ? 513 : 521) : (old_ring ? 514 : 520);               // 521 / 520: the DMA issue among the MFMAs
            }
        }
        else cfg = 128;
        // backward-input mode: the 32 x 32 x 16 kernel, whose store pass prefetches the ELU outputs (measured at M = 32768, dX of the 1024 -> 1024 /
        // 1536 -> 1024 / 2048 -> 1536 layers: 204 / 313 / 507 us against 227 / 322 / 521 on the 16 x 16 x 32 kernel, which PPENV_MLP_TILE=516 still selects)
        if (bw && (cfg == 516 || cfg == 517)) cfg = 512;
    }
    if (bw && (cfg == 517 || cfg == 600)) { ppenv_set_error("ppenv_mlp_layer_backward_input: PPENV_MLP_TILE names a kernel without the backward store pass"); return PPENV_EINVAL; }
#define PP_LAUNCH(WM_, WN_, TI_, TJ_, BK_)                                                                                                    \
    do {                                                                                                                                      \
        const dim3 grid((L->n + 32 * TJ_ * WN_ - 1) / (32 * TJ_ * WN_), (L->m + 32 * TI_ * WM_ - 1) / (32 * TI_ * WM_), L->batch), block(64 * WM_ * WN_); \
        if (L->in_f32) hipLaunchKernelGGL((mlp_layer_kernel<true, WM_, WN_, TI_, TJ_, BK_>), grid, block, 0, (hipStream_t)stream, a);        \
        else hipLaunchKernelGGL((mlp_layer_kernel<false, WM_, WN_, TI_, TJ_, BK_>), grid, block, 0, (hipStream_t)stream, a);                  \
    } while (0)
    if (cfg >= 512 && cfg <= 521 && cfg != 519) {
        const bool ok = !L->in_f32 && L->k % 64 == 0 && L->lda % 8 == 0 && L->ldw % 8 == 0 && L->in_stride % 8 == 0 && L->w_stride % 8 == 0 &&
                        (reinterpret_cast<uintptr_t>(L->in) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->w) & 15) == 0;
        if (!ok) cfg = (wgs(256, 256) >= 3 * cus / 4 && L->n >= 256) ? 384 : 128;
    }
    if (cfg == 600) {
        hipLaunchKernelGGL(mlp_heads_kernel, dim3((L->m + 31) / 32), dim3(256), 0, (hipStream_t)stream, a, SampleArgs{});
    } else if (cfg == 512) {
        const int tn = (L->n + 255) / 256, tm = (L->m + 255) / 256;
        hipLaunchKernelGGL((mlp_layer_pp_kernel<false, 4>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 516) {
        const int tn = (L->n + 255) / 256, tm = (L->m + 255) / 256;
        hipLaunchKernelGGL((mlp_layer_pp_kernel<true, 4>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 517) {
        const int tn = (L->n + 191) / 192, tm = (L->m + 255) / 256;
        hipLaunchKernelGGL((mlp_layer_pp_kernel<true, 3>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 513) {
        const int tn = (L->n + 255) / 256, tm = (L->m + 127) / 128;
        hipLaunchKernelGGL((mlp_layer_pp1_kernel<2, 3>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 520) {
        const int tn = (L->n + 127) / 128, tm = (L->m + 127) / 128;
        hipLaunchKernelGGL(mlp_layer_pp3_kernel<1>, dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 521) {
        const int tn = (L->n + 255) / 256, tm = (L->m + 127) / 128;
        hipLaunchKernelGGL(mlp_layer_pp3_kernel<2>, dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 518) {
        const int tn = (L->n + 127) / 128, tm = (L->m + 127) / 128;
        hipLaunchKernelGGL((mlp_layer_pp2_kernel<2, 4>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 514 || cfg == 515) {                                  // 515: the same with a five-deep ring (round 4: measured, no faster)
        const int tn = (L->n + 127) / 128, tm = (L->m + 127) / 128;
        if (cfg == 514) hipLaunchKernelGGL((mlp_layer_pp1_kernel<1, 3>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
        else hipLaunchKernelGGL((mlp_layer_pp1_kernel<1, 5>), dim3(tn * tm, L->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tm);
    } else if (cfg == 384) PP_LAUNCH(2, 4, 4, 2, 64);
    else if (cfg == 385) PP_LAUNCH(2, 4, 4, 2, 32);
    else if (cfg == 129) PP_LAUNCH(2, 2, 2, 2, 32);
    else PP_LAUNCH(2, 2, 2, 2, 64);
#undef PP_LAUNCH
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_layer_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
}  // namespace

extern "C" int ppenv_mlp_layer_forward(const ppenv_mlp_layer* L, void* stream) { return launch_layer(L, nullptr, stream); }

// 2 .. 4 consecutive hidden layers in one launch (include/ppenv_policy.h)
extern "C" size_t ppenv_mlp_chain_workspace_bytes(int32_t m, int32_t batch, int32_t count) {
    if (m <= 0 || batch <= 0 || count < 2 || count > 4) return 0;
    return sizeof(unsigned) * (size_t)(3 + (count - 1) * batch * ((m + 127) / 128));
}
extern "C" int ppenv_mlp_chain_status(const void* workspace) {
    unsigned w[3] = {0, 0, 0};
    if (!workspace || hipMemcpy(w, workspace, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) { ppenv_set_error("ppenv_mlp_chain_status: NULL or unreadable workspace"); return PPENV_EHIP; }
    return w[2] ? 1 : 0;
}
extern "C" int ppenv_mlp_chain_forward(const ppenv_mlp_layer* layers, int32_t count, void* workspace, void* stream) {
    if (!layers || !workspace || count < 2 || count > 4 || (reinterpret_cast<uintptr_t>(workspace) & 3)) {
        ppenv_set_error("ppenv_mlp_chain_forward: 2 .. 4 layers and a 4-byte aligned workspace of ppenv_mlp_chain_workspace_bytes (zeroed once, by the caller)");
        return PPENV_EINVAL;
    }
    ChainArgs c{};
    { const char* dbg = getenv("PPENV_CHAIN_DEBUG"); c.debug = dbg ? atoi(dbg) : 0; }
    c.count = count; c.batch = layers[0].batch; c.tiles_m = (layers[0].m + 127) / 128; c.sync = reinterpret_cast<unsigned*>(workspace);
    int ticket = 0;
    for (int l = 0; l < count; l++) {
        const ppenv_mlp_layer* L = &layers[l];
        const bool ok = L->in && L->w && L->out && L->m == layers[0].m && L->batch == c.batch && L->m > 0 && L->n >= 128 && L->k > 0 && !L->in_f32 && !L->out_f32 &&
                        L->k % 64 == 0 && L->lda >= L->k && L->ldw >= L->k && L->ldo >= L->n && L->lda % 8 == 0 && L->ldw % 8 == 0 && L->ldo % 8 == 0 &&
                        L->in_stride % 8 == 0 && L->w_stride % 8 == 0 && L->out_stride % 8 == 0 &&
                        (reinterpret_cast<uintptr_t>(L->in) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->w) & 15) == 0 && (reinterpret_cast<uintptr_t>(L->out) & 15) == 0;
        const bool chained = l == 0 || (L->in == layers[l - 1].out && L->k == layers[l - 1].n && L->lda == layers[l - 1].ldo && L->in_stride == layers[l - 1].out_stride);
        if (!ok || !chained) {
            ppenv_set_error("ppenv_mlp_chain_forward: every layer must qualify for the LDS-DMA ring tiles (fp16 in / out, k % 64 == 0, n >= 128, 16-byte aligned rows, "
                            "equal m and batch) and read exactly what the layer before it writes");
            return PPENV_EINVAL;
        }
        c.a[l] = Args{L->m, L->n, L->k, L->lda, L->ldw, L->ldo, L->elu, L->out_f32, L->in, (long long)L->in_stride, L->mean, L->inv_std, L->clip,
                      reinterpret_cast<const _Float16*>(L->w), (long long)L->w_stride, reinterpret_cast<const _Float16*>(L->bias), (long long)L->bias_stride,
                      L->out, (long long)L->out_stride};
        const long long w256 = (long long)((L->n + 255) / 256) * c.tiles_m * c.batch;
        c.tj[l] = w256 >= 192 ? 2 : 1;                              // as launch_layer: 128 x 256 where that still gives three quarters of the CUs a tile
        c.tiles_n[l] = (L->n + 128 * c.tj[l] - 1) / (128 * c.tj[l]);
        c.first[l] = ticket;
        ticket += c.tiles_n[l] * c.tiles_m * c.batch;
    }
    c.first[count] = ticket;
    static int cus = 0;
    if (cus == 0) { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { ppenv_set_error("hipGetDeviceProperties failed"); return PPENV_EHIP; } cus = prop.multiProcessorCount; }
    hipLaunchKernelGGL(mlp_chain_pp3_kernel, dim3(ticket < cus ? ticket : cus), dim3(512), 0, (hipStream_t)stream, c);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_chain_pp3_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

// The same layer with its grid sized for `cus` of the chip's 256 CUs (include/ppenv_policy.h): the tile choice above aims at one workgroup
// per CU of that share, so that the launch sequences of two (cus = 128) env groups on two streams run side by side instead of queueing.
extern "C" int ppenv_mlp_layer_forward_share(const ppenv_mlp_layer* L, int32_t cus, void* stream) {
    if (cus < 0 || cus > 256) { ppenv_set_error("ppenv_mlp_layer_forward_share: cus must be 0 (the whole chip) .. 256"); return PPENV_EINVAL; }
    return launch_layer(L, nullptr, stream, cus);
}

// dX of a layer as a launch of the forward kernels (include/ppenv_policy.h): g describes dx[m, k_fwd] = dz[m, n_fwd] . wt[k_fwd, n_fwd]^T in
// forward form (in = dz, w = wt: the transposed weights, K-contiguous for this product), bias NULL, elu 0, fp16 out.
extern "C" int ppenv_mlp_layer_backward_input(const ppenv_mlp_layer* g, const uint16_t* elu_out, int64_t elu_out_stride, int32_t ld_elu_out,
                                              float* colsum_partial, int64_t colsum_stride, int32_t ld_colsum, void* stream) {
    if (!g || g->bias || g->elu || g->out_f32) {
        ppenv_set_error("ppenv_mlp_layer_backward_input: the descriptor must have bias NULL, elu 0 and an fp16 output");
        return PPENV_EINVAL;
    }
    if ((elu_out && ld_elu_out < g->n) || (colsum_partial && ld_colsum < g->n)) {
        ppenv_set_error("ppenv_mlp_layer_backward_input: ld_elu_out / ld_colsum smaller than the row length");
        return PPENV_EINVAL;
    }
    const BwdInput bw{reinterpret_cast<const _Float16*>(elu_out), (long long)elu_out_stride, ld_elu_out, colsum_partial, (long long)colsum_stride, ld_colsum};
    return launch_layer(g, &bw, stream);
}
