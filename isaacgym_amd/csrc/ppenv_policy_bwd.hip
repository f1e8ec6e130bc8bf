// ppenv_policy_bwd.hip — the learner's half of the policy MLP on the matrix cores (include/ppenv_policy.h):
// the weight gradient dW = dZ^T . X, the reductions behind it, the fp32 -> fp16 weight images, and rl_games' RunningMeanStd update.
// (The input gradient dX = dZ . W runs on the forward kernels of ppenv_policy.hip with the transposed weight image.)
//
// dW[n, k] = sum_m dZ[m, n] X[m, k]: the contraction runs over the ROW index of both operands, so neither is "K-contiguous" the way
// v_mfma_f32_32x32x16_f16 wants its fragments (eight consecutive contraction values of one output row / column per lane).  CDNA4 has the
// instruction for exactly this: ds_read_b64_tr_b16 reads a 4 (rows) x 16 (columns) block of 16-bit values per 16 lanes and hands every
// lane one COLUMN of it.  Both tiles therefore stay row-major in LDS as they arrive from HBM — [64 contraction rows][64 columns] pieces,
// filled by global_load_lds_dwordx4 without touching a register — and both MFMA operands are read transposed: two tr reads (rows m .. m+3
// and m+4 .. m+7 of the lane's column) make one operand fragment.
//
// Tile and schedule are the forward's 256 x 256 kernel (ppenv_policy.hip mlp_layer_pp_kernel): 512 threads = 8 waves as 2 (n) x 4 (k),
// 128 x 64 of dW per wave (4 x 2 MFMA tiles, 128 accumulator registers), contraction in tiles of 64 rows, the two waves of a SIMD in
// different groups that take turns on the matrix core, counted vmcnt.  What differs is the LDS image: rows of a piece are 128 bytes, a
// 32-lane half of a tr read takes rows r0 .. r0+3 (r0 a multiple of 4) x four 16-byte chunks c0 .. c0+3 (c0 = 0 or 4): 256 bytes = all 64
// banks exactly once iff rows r0 and r0+2 (same parity = same half of the 256-byte bank row) use different chunk groups: slot = chunk ^
// (4 if row & 2).  The DMA writes lane-linear, so the swizzle is applied on the SOURCE address (which chunk a lane fetches).
//
// M = 32768 rows (the learner's minibatch) against N, K <= 2048: few output tiles, a long contraction.  The contraction is split over
// `splits` workgroups per tile; a split's workgroups are placed on ONE XCD (consecutive workgroup ids go round the eight XCDs), so the
// rows of dZ and X a split walks through are fetched from HBM once and shared through that XCD's L2 by all of its tiles.  Partial tiles
// go to a workspace [splits][N][K] fp32 and are summed by reduce_rows_kernel in a fixed order (deterministic; splits = 1 writes dW itself).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/ppenv.h"
#include "../../include/ppenv_policy.h"

void ppenv_set_error(const char* msg);   // ppenv.hip

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct DwArgs {
    int m, n, k, lddz, ldx, lddw, splits, accumulate;
    const _Float16* dz; long long dz_stride;
    const _Float16* x; long long x_stride;
    float* out; long long out_stride;      // dW (splits == 1) or the workspace; per batch
    long long split_stride;                // between the partial images of consecutive splits (elements)
};

// one MFMA operand fragment = rows r .. r+3 and r+4 .. r+7 of the lane's column, read transposed (EXEC must be all ones).
// Inline assembly, not __builtin_amdgcn_ds_read_tr16_b64: the compiler cannot tell which LDS bytes the builtin reads and puts an
// s_waitcnt vmcnt(0) in front of every group of them while LDS-DMA writes are in flight — which would drain the pipeline this kernel is
// built around twice per tile.  The price: the compiler does not know these reads are asynchronous either, so every use of a fragment
// sits behind an explicit s_waitcnt lgkmcnt(0) + scheduling barrier (mfmas()).  addr: LDS byte address; OFF: immediate byte offset.
template <int OFF>
__device__ __forceinline__ h8 tr_fragment(unsigned addr) {
    s4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 4 * 64 * 2) : "memory");
    return __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(512) void mlp_dw_kernel(const DwArgs a, const int tiles_n, const int tiles_k) {
    constexpr int TN = 256, TK = 256, BM = 64, PIECE = 64 * 64, TILE = 8 * PIECE, NB = 4;   // a piece: [64 rows][64 columns] fp16 = 8 KiB
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * TILE];                          // two [A 0..3 | B 0..3] tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    // workgroup -> (split, tile): all tiles of a split on one XCD (or, with fewer than 8 splits, on the 8 / splits XCDs that share it)
    const int T = tiles_n * tiles_k, S = a.splits, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    int split, tile;
    if (S >= 8) { split = xcd + 8 * (local / T); tile = local % T; }
    else { const int R = 8 / S; split = xcd % S; tile = local * R + xcd / S; }
    if (tile >= T || split >= S) return;                           // the whole workgroup: no barrier has been reached
    const int n0 = (tile % tiles_n) * TN, k0 = (tile / tiles_n) * TK, b = blockIdx.y;
    const int ktiles_all = a.m / BM, kt0 = (int)((long long)split * ktiles_all / S), ktiles = (int)((long long)(split + 1) * ktiles_all / S) - kt0;
    const _Float16* dz = a.dz + (size_t)b * a.dz_stride;
    const _Float16* xx = a.x + (size_t)b * a.x_stride;

    // staging: thread -> (row tid >> 3 of the 64-row piece, 16-byte slot tid & 7); it fetches chunk slot ^ (4 if row & 2) of that row.
    // Columns beyond n / k are clamped to the last whole chunk (garbage in accumulators that are never stored).
    const int srow = tid >> 3, kc = (lane & 7) ^ (((srow >> 1) & 1) << 2);
    const _Float16* pa[4];
    const _Float16* pb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int ca = n0 + i * 64 + kc * 8, cb = k0 + i * 64 + kc * 8;
        ca = ca + 8 <= a.n ? ca : a.n - 8;
        cb = cb + 8 <= a.k ? cb : a.k - 8;
        pa[i] = dz + (size_t)(kt0 * BM + srow) * a.lddz + ca;
        pb[i] = xx + (size_t)(kt0 * BM + srow) * a.ldx + cb;
    }
    const long long step_a = (long long)BM * a.lddz, step_b = (long long)BM * a.ldx;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // A piece i = columns n0 + 64 i .. + 63 (wave row i >> 1, its half i & 1); B piece i = columns k0 + 64 i .. + 63 (wave column i)
    auto stage_a = [&](int buf, int t, int i) {
        __builtin_amdgcn_global_load_lds((glb_ptr)(pa[i] + t * step_a), (lds_ptr)(smem + buf * TILE + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };
    auto stage_b = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < NB; i++)
            __builtin_amdgcn_global_load_lds((glb_ptr)(pb[i] + t * step_b), (lds_ptr)(smem + buf * TILE + 4 * PIECE + (i * 512 + wave * 64) * 8), 16, 0, 0);
    };

    f16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // transposed fragment reads.  16-lane group g = lane >> 4 (h = g >> 1: which eight contraction rows of the 16-deep step; g & 1: which
    // 16 of the MFMA tile's 32 columns), lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3 of the block and receives
    // column (lane & 15) of its four rows.  Tile ti of the piece = its columns 32 ti .. 32 ti + 31 = chunks 4 ti .. 4 ti + 3.
    const int q = (lane >> 2) & 3, p = lane & 3, g = lane >> 4;
    const int sw = ((q >> 1) & 1) << 2;                            // the row's swizzle term: rows 16 kk + 8 h + q (+ 4) have bit 1 of q
    typedef __attribute__((address_space(3))) _Float16* lds_h;
    const unsigned smem_addr = (unsigned)(uintptr_t)(lds_h)smem;   // LDS byte address of the tile buffers
    unsigned loff[2];                                              // bytes, within a piece
#pragma unroll
    for (int ti = 0; ti < 2; ti++) loff[ti] = 2u * ((8 * (g >> 1) + q) * 64 + (((4 * ti) ^ sw) + 2 * (g & 1) + (p >> 1)) * 8 + (p & 1) * 4);
    h8 fa[2][4], fb[2][4];
    auto read_a = [&](int buf, int half) {
        const unsigned piece = smem_addr + 2u * (buf * TILE + (2 * wm + half) * PIECE);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const unsigned ad = piece + loff[i];
            fa[i][0] = tr_fragment<0>(ad); fa[i][1] = tr_fragment<2048>(ad); fa[i][2] = tr_fragment<4096>(ad); fa[i][3] = tr_fragment<6144>(ad);   // 16 rows x 128 bytes per k-step
        }
    };
    auto read_b = [&](int buf) {
        const unsigned piece = smem_addr + 2u * (buf * TILE + (4 + wn) * PIECE);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned ad = piece + loff[j];
            fb[j][0] = tr_fragment<0>(ad); fb[j][1] = tr_fragment<2048>(ad); fb[j][2] = tr_fragment<4096>(ad); fb[j][3] = tr_fragment<6144>(ad);
        }
    };
    auto mfmas = [&](int half) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the fragments (see tr_fragment): nothing below may move above this
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[half * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][kk], fb[j][kk], acc[half * 2 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

    // the forward kernel's phase structure and wait counts (see there): 2 + NB + 2 pieces per tile, A halves 1 / 3 issued last
    if (ktiles > 0) {
        stage_a(0, 0, 0); stage_a(0, 0, 2);
        stage_b(0, 0);
        stage_a(0, 0, 1); stage_a(0, 0, 3);
        PP_VM(2);
        PP_BARRIER();
        if (wm == 1) PP_BARRIER();                                 // group 1 runs one barrier behind
        for (int t = 0; t < ktiles; t++) {
            const bool more = t + 1 < ktiles;
            const int cbuf = t & 1, nbuf = (t + 1) & 1;
            // phase A: columns 0-63 of the wave's 128 (n), all of its 64 (k)
            read_a(cbuf, 0);
            read_b(cbuf);
            if (more) { stage_a(nbuf, t + 1, 0); stage_a(nbuf, t + 1, 2); stage_b(nbuf, t + 1); }
            if (wm == 1) { if (more) PP_VM(2 + NB); else PP_VM(0); }
            PP_BARRIER();
            mfmas(0);
            if (wm == 0) { if (more) PP_VM(2 + NB); else PP_VM(0); }
            PP_BARRIER();
            // phase B: columns 64-127 (n)
            read_a(cbuf, 1);
            if (more) { stage_a(nbuf, t + 1, 1); stage_a(nbuf, t + 1, 3); }
            if (wm == 1) PP_VM(2);
            PP_BARRIER();
            mfmas(1);
            if (wm == 0) PP_VM(2);
            PP_BARRIER();
        }
        if (wm == 0) PP_BARRIER();                                 // as many barriers as group 1
    }
#undef PP_BARRIER
#undef PP_VM

    // C layout of the 32 x 32 MFMA: column (k) = lane & 31, row (n) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): 32 lanes store 128 contiguous bytes
    float* out = a.out + (size_t)b * a.out_stride + (size_t)split * a.split_stride;
    const int r = lane & 31, h = lane >> 5;
    const bool add = a.accumulate && S == 1;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = k0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = n0 + wm * 128 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (row < a.n && col < a.k) {
                    float* dst = out + (size_t)row * a.lddw + col;
                    *dst = add ? *dst + acc[i][j][reg] : acc[i][j][reg];
                }
            }
        }
}

// dst[i] (+)= sum_r src[r * row_stride + i], rows in order: the split-K partials of dW, the per-block column sums behind a bias gradient
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ src, int rows, long long row_stride, long long n, float* __restrict__ dst,
                                                          int accumulate) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n && (row_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        f4v s = accumulate ? *reinterpret_cast<const f4v*>(dst + i4) : f4v{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rows; r++) s += *reinterpret_cast<const f4v*>(src + (size_t)r * row_stride + i4);
        *reinterpret_cast<f4v*>(dst + i4) = s;
    } else {
        for (long long i = i4; i < n && i < i4 + 4; i++) {
            float s = accumulate ? dst[i] : 0.f;
            for (int r = 0; r < rows; r++) s += src[(size_t)r * row_stride + i];
            dst[i] = s;
        }
    }
}

// the same for a destination [n, ld_dst] with ld_dst > k (a padded weight-gradient image): src rows are [n, k] contiguous
__global__ __launch_bounds__(256) void reduce_rows_2d_kernel(const float* __restrict__ src, int rows, long long row_stride, int n, int k, float* __restrict__ dst,
                                                             int ld_dst, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * k) return;
    const int r = (int)(i / k), c = (int)(i - (long long)r * k);
    float* d = dst + (size_t)r * ld_dst + c;
    float s = accumulate ? *d : 0.f;
    for (int p = 0; p < rows; p++) s += src[(size_t)p * row_stride + i];
    *d = s;
}

// the same sum for MANY rows of a narrow matrix (the 512 per-block column sums behind a bias gradient at M = 32768): one thread walking all the
// rows would be 512 dependent round trips to memory.  A workgroup owns 64 columns: thread -> float4 column tid & 15, row class tid >> 4 (rows
// rc, rc + 16, ... four loads in flight), then the 16 classes are added through LDS in class order — a fixed order, so still deterministic.
__global__ __launch_bounds__(256) void reduce_tall_kernel(const float* __restrict__ src, int rows, long long row_stride, long long n, float* __restrict__ dst,
                                                          int accumulate) {
    __shared__ float part[16][65];
    const int cg = threadIdx.x & 15, rc = threadIdx.x >> 4;
    const long long c0 = (long long)blockIdx.x * 64 + cg * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < n) {
        const bool vec = c0 + 4 <= n && (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        int r = rc;
        for (; r + 48 < rows && vec; r += 64) {
            f4v v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const f4v*>(src + (size_t)(r + 16 * u) * row_stride + c0);
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) s[e] += v[u][e];
        }
        for (; r < rows; r += 16)
#pragma unroll
            for (int e = 0; e < 4; e++) if (c0 + e < n) s[e] += src[(size_t)r * row_stride + c0 + e];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) part[rc][cg * 4 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
        if (c < n) {
            float t = accumulate ? dst[c] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) t += part[i][threadIdx.x];
            dst[c] = t;
        }
    }
}

// column sums of an fp32 [m, n] matrix per block of 1024 rows (the heads' bias gradient: n = num_actions + 1): thread -> column t & 31 of
// the 32-column group blockIdx.y, row class t >> 5; partial[blockIdx.x][n]
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, int m, int n, int ld, float* __restrict__ partial) {
    __shared__ float part[8][33];
    const int c = blockIdx.y * 32 + (threadIdx.x & 31), rc = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 1024, r1 = r0 + 1024 < m ? r0 + 1024 : m;
    float s = 0.f;
    if (c < n)
        for (int r = r0 + rc; r < r1; r += 8) s += in[(size_t)r * ld + c];
    part[rc][threadIdx.x & 31] = s;
    __syncthreads();
    if (rc == 0 && c < n) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) t += part[i][threadIdx.x & 31];
        partial[(size_t)blockIdx.x * n + c] = t;
    }
}

// fp32 master weights [n, k] -> the two fp16 operand images: w16 [n, ldw] (the forward's and dW's layout, rows zero-padded to ldw) and
// wt16 [k_pad, ldwt] = its transpose (dX's operand: K-contiguous for the product over n), zero beyond n / k.  32 x 32 tiles through LDS.
__device__ __forceinline__ void cast_tile(const float* __restrict__ w32, int n, int k, int ldw32, _Float16* __restrict__ w16, int ldw, _Float16* __restrict__ wt16, int ldwt,
                                          int kpad, int npad, int bx, int by, float (*t)[33]) {
    const int c0 = bx * 32, r0 = by * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        const float v = (r < n && c < k) ? w32[(size_t)r * ldw32 + c] : 0.f;
        t[ty + 8 * i][tx] = v;
        if (w16 && r < n && c < ldw) w16[(size_t)r * ldw + c] = (_Float16)v;
    }
    __syncthreads();
    if (wt16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int kk = c0 + ty + 8 * i, nn = r0 + tx;          // wt16[kk][nn] = w[nn][kk]
            if (kk < kpad && nn < npad && nn < ldwt) wt16[(size_t)kk * ldwt + nn] = (_Float16)t[tx][ty + 8 * i];
        }
    }
}
__global__ __launch_bounds__(256) void cast_weights_kernel(const float* __restrict__ w32, int n, int k, int ldw32, _Float16* __restrict__ w16, int ldw,
                                                           _Float16* __restrict__ wt16, int ldwt, int kpad, int npad) {
    __shared__ float t[32][33];
    cast_tile(w32, n, k, ldw32, w16, ldw, wt16, ldwt, kpad, npad, blockIdx.x, blockIdx.y, t);
}
// every matrix of the network in ONE launch (the per-optimizer-step cast: 13 weight matrices and 7 bias vectors were 13 launches + 9 copies): the items and the
// prefix sums of their 32 x 32 tile counts travel by value in the kernel argument; a workgroup finds its item by a scan of at most 32 entries
constexpr int kCastMax = 32;
struct CastBatch {
    int count;
    int tile_off[kCastMax + 1];
    int tiles_x[kCastMax];
    ppenv_mlp_cast it[kCastMax];
};
__global__ __launch_bounds__(256) void cast_weights_batch_kernel(const CastBatch b) {
    __shared__ float t[32][33];
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.tile_off[i + 1]) i++;
    const ppenv_mlp_cast& c = b.it[i];
    const int local = (int)blockIdx.x - b.tile_off[i];
    cast_tile(c.w32, c.n, c.k, c.ldw32, reinterpret_cast<_Float16*>(c.w16), c.ldw16, reinterpret_cast<_Float16*>(c.wt16), c.ldwt16, c.wt_rows, c.ldwt16,
              local % b.tiles_x[i], local / b.tiles_x[i], t);
}

// rl_games' RunningMeanStd in training mode on one batch of observations [m, k] (fp32): per column the batch mean and the UNBIASED batch
// variance (torch.var's default), merged into the running (mean, var, count) by the parallel-moments rule of
// _update_mean_var_count_from_moments; state in float64 as rl_games keeps it.  One launch, one pass over obs: workgroup (x, y) sums rows
// 128 x .. 128 x + 127 of columns 64 y .. 64 y + 63 in float64 (thread -> column tid & 63, row class tid >> 6: coalesced rows, eight loads
// in flight); the workgroup that draws the last ticket OF ITS COLUMN CHUNK merges that chunk's row blocks in a fixed order, updates the
// state of its 64 columns and writes the fp32 mean and 1 / sqrt(var + eps) the forward's normaliser reads; the last chunk to finish
// advances the count and re-arms the tickets (every chunk has read the old count by then).
constexpr int kRmsRows = 128, kRmsMaxChunks = 254;
__global__ __launch_bounds__(256) void rms_update_kernel(const float* __restrict__ obs, int m, int k, int ld, double* __restrict__ partial /*[row blocks][2][k]*/,
                                                         unsigned int* __restrict__ tickets /*[kRmsMaxChunks] per chunk, [255] chunks done*/, double* __restrict__ mean,
                                                         double* __restrict__ var, double* __restrict__ count, float* __restrict__ mean32, float* __restrict__ inv_std32,
                                                         float eps) {
    __shared__ double red[2][4][64];
    __shared__ bool last;
    const int nrb = gridDim.x, nchunks = gridDim.y, cx = threadIdx.x & 63, rc = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx, r0 = blockIdx.x * kRmsRows, r1 = r0 + kRmsRows < m ? r0 + kRmsRows : m;
    const bool col = c < k;
    double s = 0.0, s2 = 0.0;
    if (col) {
        int r = r0 + rc;
        for (; r + 28 < r1; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = obs[(size_t)(r + 4 * u) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; u++) { s += (double)v[u]; s2 += (double)v[u] * (double)v[u]; }
        }
        for (; r < r1; r += 4) { const double v = (double)obs[(size_t)r * ld + c]; s += v; s2 += v * v; }
    }
    red[0][rc][cx] = s; red[1][rc][cx] = s2;
    __syncthreads();
    if (rc == 0 && col) {
        partial[((size_t)blockIdx.x * 2 + 0) * k + c] = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
        partial[((size_t)blockIdx.x * 2 + 1) * k + c] = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&tickets[blockIdx.y], 1u) == (unsigned)nrb - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // this chunk's row blocks, rows rc, rc + 4, ... per thread, then the four classes in order
    s = 0.0; s2 = 0.0;
    if (col) {
        int b = rc;
        for (; b + 28 < nrb; b += 32) {                 // eight row blocks' partials in flight (a plain loop is one dependent round trip per block)
            double v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { v[u] = partial[((size_t)(b + 4 * u) * 2 + 0) * k + c]; w[u] = partial[((size_t)(b + 4 * u) * 2 + 1) * k + c]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { s += v[u]; s2 += w[u]; }
        }
        for (; b < nrb; b += 4) { s += partial[((size_t)b * 2 + 0) * k + c]; s2 += partial[((size_t)b * 2 + 1) * k + c]; }
    }
    __syncthreads();
    red[0][rc][cx] = s; red[1][rc][cx] = s2;
    __syncthreads();
    if (rc == 0 && col) {
        s = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
        s2 = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
        const double bc = (double)m, c0 = *count, tot = c0 + bc;
        const double bmean = s / bc;
        const double bvar = m > 1 ? (s2 - bc * bmean * bmean) / (bc - 1.0) : 0.0;       // torch.var: unbiased
        const double delta = bmean - mean[c];
        const double new_mean = mean[c] + delta * bc / tot;
        const double m2 = var[c] * c0 + bvar * bc + delta * delta * c0 * bc / tot;
        const double new_var = m2 / tot;
        mean[c] = new_mean;
        var[c] = new_var;
        if (mean32) mean32[c] = (float)new_mean;
        if (inv_std32) inv_std32[c] = 1.0f / sqrtf((float)new_var + eps);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        tickets[blockIdx.y] = 0u;                                                       // re-armed for the next launch
        if (atomicAdd(&tickets[255], 1u) == (unsigned)nchunks - 1) {                    // every chunk has read the old count
            *count = *count + (double)m;
            tickets[255] = 0u;
        }
    }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// splits: the power of two that minimises a small cost model fitted to measurements (tools/gpu_mlp_bwd_layers.py --splits, M = 8192 and 32768):
//   rounds of 256 workgroups x contraction tiles per workgroup x time per 256 x 256 x 64 tile (1.25 us on a lightly used chip, 1.9 us when every CU
//   has a workgroup and the clock sags) + the partial images' round trip through memory (written once, read once; ~60 % of it hidden behind other
//   workgroups' compute at ~4 TB/s).  More splits fill the chip and shorten each workgroup's loop, but the partials grow with them while the GEMM
//   shrinks with M: at M = 8192 the widest layer wants 2 splits, the narrowest 16; at M = 32768 they want 8 and 32.
int choose_splits(const ppenv_mlp_dw* d) {
    const double tiles = (double)((d->n + 255) / 256) * ((d->k + 255) / 256) * d->batch;
    const int ktiles = d->m / 64;
    const double image_mb = (double)d->n * d->k * d->batch * 4.0 / 1e6;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 64 && s * 2 <= ktiles; s *= 2) {
        const double wgs = tiles * s, busy = wgs < 256.0 ? wgs / 256.0 : 1.0, rounds = (double)(((long long)wgs + 255) / 256);
        const double t = rounds * ((double)ktiles / s) * (1.25 + 0.65 * busy) + (s > 1 ? 0.3 * s * image_mb : 0.0);
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}
}  // namespace

extern "C" size_t ppenv_mlp_dw_workspace_bytes(const ppenv_mlp_dw* d) {
    if (!d || d->m <= 0 || d->n <= 0 || d->k <= 0 || d->batch <= 0) return 0;
    const int s = d->splits > 0 ? d->splits : choose_splits(d);
    return s <= 1 ? 0 : (size_t)s * d->batch * d->n * d->k * sizeof(float);
}

extern "C" int ppenv_mlp_layer_backward_weight(const ppenv_mlp_dw* d, void* stream) {
    if (!d || !d->dz || !d->x || !d->dw || d->m <= 0 || d->n <= 0 || d->k <= 0 || d->batch <= 0 || d->lddz < d->n || d->ldx < d->k || d->lddw < d->k) {
        ppenv_set_error("ppenv_mlp_layer_backward_weight: NULL pointer or inconsistent sizes (need lddz >= n, ldx >= k, lddw >= k)");
        return PPENV_EINVAL;
    }
    if (d->m % 64 || d->n % 8 || d->k % 8 || d->lddz % 8 || d->ldx % 8 || d->dz_stride % 8 || d->x_stride % 8 ||
        (reinterpret_cast<uintptr_t>(d->dz) & 15) || (reinterpret_cast<uintptr_t>(d->x) & 15)) {
        ppenv_set_error("ppenv_mlp_layer_backward_weight: needs m % 64 == 0, n % 8 == 0, k % 8 == 0 and 16-byte aligned rows (lddz, ldx, strides multiples of 8)");
        return PPENV_EINVAL;
    }
    const int s = d->splits > 0 ? d->splits : choose_splits(d);
    if (!pow2(s) || s > d->m / 64) { ppenv_set_error("ppenv_mlp_layer_backward_weight: splits must be a power of two, at most m / 64"); return PPENV_EINVAL; }
    const size_t need = s <= 1 ? 0 : (size_t)s * d->batch * d->n * d->k * sizeof(float);
    if (need && (!d->workspace || d->workspace_bytes < need)) {
        ppenv_set_error("ppenv_mlp_layer_backward_weight: workspace missing or smaller than ppenv_mlp_dw_workspace_bytes()");
        return PPENV_EINVAL;
    }
    const int tn = (d->n + 255) / 256, tk = (d->k + 255) / 256, T = tn * tk;
    DwArgs a{d->m, d->n, d->k, d->lddz, d->ldx, s > 1 ? d->k : d->lddw, s, d->accumulate, reinterpret_cast<const _Float16*>(d->dz), (long long)d->dz_stride,
             reinterpret_cast<const _Float16*>(d->x), (long long)d->x_stride,
             s > 1 ? reinterpret_cast<float*>(d->workspace) : d->dw, s > 1 ? (long long)d->n * d->k : (long long)d->dw_stride,
             s > 1 ? (long long)d->batch * d->n * d->k : 0};
    const int per_xcd = s >= 8 ? (s / 8) * T : (T + (8 / s) - 1) / (8 / s);
    hipLaunchKernelGGL(mlp_dw_kernel, dim3(8 * per_xcd, d->batch), dim3(512), 0, (hipStream_t)stream, a, tn, tk);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching mlp_dw_kernel failed"); return PPENV_EHIP; }
    if (s > 1) {
        // workspace [s][batch][n][k] -> dw [batch][n, lddw]: one reduce per batch entry and, when lddw != k, per row block — the common
        // case (lddw == k, dw_stride == n k) is ONE launch over batch * n * k elements
        const long long nk = (long long)d->n * d->k;
        if (d->lddw == d->k && (d->batch == 1 || d->dw_stride == nk)) {
            const long long tot = nk * d->batch;
            hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((tot / 4 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const float*>(d->workspace), s, nk * d->batch, tot, d->dw, d->accumulate);
        } else {                                     // a padded leading dimension or batch stride: one launch per batch entry, (row, column) addressing
            for (int b = 0; b < d->batch; b++)
                hipLaunchKernelGGL(reduce_rows_2d_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                   reinterpret_cast<const float*>(d->workspace) + (size_t)b * nk, s, nk * d->batch, d->n, d->k,
                                   d->dw + (size_t)b * d->dw_stride, d->lddw, d->accumulate);
        }
        if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching reduce_rows_kernel failed"); return PPENV_EHIP; }
    }
    return PPENV_OK;
}

extern "C" int ppenv_mlp_reduce_rows(const float* partial, int32_t rows, int64_t row_stride, int64_t n, float* out, int32_t accumulate, void* stream) {
    if (!partial || !out || rows <= 0 || n <= 0 || row_stride < n) { ppenv_set_error("ppenv_mlp_reduce_rows: NULL pointer or inconsistent sizes"); return PPENV_EINVAL; }
    if (rows > 32)      // many rows (the per-64-row-block column sums): rows spread over the workgroup
        hipLaunchKernelGGL(reduce_tall_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, partial, rows, (long long)row_stride, (long long)n, out, accumulate);
    else
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, partial, rows, (long long)row_stride,
                           (long long)n, out, accumulate);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching reduce_rows_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" size_t ppenv_mlp_bias_grad_workspace_bytes(int32_t m, int32_t n) { return (m <= 0 || n <= 0) ? 0 : (size_t)((m + 1023) / 1024) * n * sizeof(float); }

extern "C" int ppenv_mlp_bias_grad_f32(const float* dz, int32_t m, int32_t n, int32_t ld, void* workspace, float* out, int32_t accumulate, void* stream) {
    if (!dz || !workspace || !out || m <= 0 || n <= 0 || ld < n) { ppenv_set_error("ppenv_mlp_bias_grad_f32: NULL pointer or inconsistent sizes"); return PPENV_EINVAL; }
    const int blocks = (m + 1023) / 1024;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(blocks, (n + 31) / 32), dim3(256), 0, (hipStream_t)stream, dz, m, n, ld, reinterpret_cast<float*>(workspace));
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(workspace), blocks,
                       (long long)n, (long long)n, out, accumulate);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching the bias-gradient kernels failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_mlp_cast_weights(const float* w32, int32_t n, int32_t k, int32_t ldw32, uint16_t* w16, int32_t ldw16, uint16_t* wt16, int32_t ldwt16,
                                      int32_t wt_rows, void* stream) {
    if (!w32 || (!w16 && !wt16) || n <= 0 || k <= 0 || ldw32 < k || (w16 && ldw16 < k) || (wt16 && (ldwt16 < n || wt_rows < k))) {
        ppenv_set_error("ppenv_mlp_cast_weights: NULL pointer or inconsistent sizes (need ldw32 >= k, ldw16 >= k, ldwt16 >= n, wt_rows >= k)");
        return PPENV_EINVAL;
    }
    const int cols = w16 ? (ldw16 > wt_rows ? ldw16 : wt_rows) : wt_rows, rows = wt16 ? (ldwt16 > n ? ldwt16 : n) : n;   // cover the zero padding of both images
    hipLaunchKernelGGL(cast_weights_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, w32, n, k, ldw32,
                       reinterpret_cast<_Float16*>(w16), ldw16, reinterpret_cast<_Float16*>(wt16), ldwt16, wt_rows, ldwt16);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching cast_weights_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" int ppenv_mlp_cast_weights_batch(const ppenv_mlp_cast* items, int32_t count, void* stream) {
    if (!items || count <= 0 || count > kCastMax) { ppenv_set_error("ppenv_mlp_cast_weights_batch: NULL items or count outside 1 .. 32"); return PPENV_EINVAL; }
    CastBatch b;
    b.count = count;
    int off = 0;
    for (int i = 0; i < count; i++) {
        const ppenv_mlp_cast& c = items[i];
        if (!c.w32 || (!c.w16 && !c.wt16) || c.n <= 0 || c.k <= 0 || c.ldw32 < c.k || (c.w16 && c.ldw16 < c.k) || (c.wt16 && (c.ldwt16 < c.n || c.wt_rows < c.k))) {
            ppenv_set_error("ppenv_mlp_cast_weights_batch: an item has a NULL pointer or inconsistent sizes (as ppenv_mlp_cast_weights)");
            return PPENV_EINVAL;
        }
        const int cols = c.w16 ? (c.ldw16 > c.wt_rows ? c.ldw16 : c.wt_rows) : c.wt_rows, rows = c.wt16 ? (c.ldwt16 > c.n ? c.ldwt16 : c.n) : c.n;
        b.it[i] = c;
        if (!c.wt16) { b.it[i].wt_rows = 0; b.it[i].ldwt16 = 0; }
        b.tiles_x[i] = (cols + 31) / 32;
        b.tile_off[i] = off;
        off += b.tiles_x[i] * ((rows + 31) / 32);
    }
    b.tile_off[count] = off;
    hipLaunchKernelGGL(cast_weights_batch_kernel, dim3(off), dim3(256), 0, (hipStream_t)stream, b);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching cast_weights_batch_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}

extern "C" size_t ppenv_running_mean_std_workspace_bytes(int32_t m, int32_t k) {
    return (m <= 0 || k <= 0) ? 0 : 1024 + (size_t)((m + kRmsRows - 1) / kRmsRows) * 2 * k * sizeof(double);
}

extern "C" int ppenv_running_mean_std_update(const float* obs, int32_t m, int32_t k, int32_t ld, double* mean, double* var, double* count, float* mean_f32,
                                             float* inv_std_f32, float eps, void* workspace, void* stream) {
    if (!obs || !mean || !var || !count || !workspace || m <= 0 || k <= 0 || ld < k || (reinterpret_cast<uintptr_t>(workspace) & 7) || (k + 63) / 64 > kRmsMaxChunks) {
        ppenv_set_error("ppenv_running_mean_std_update: NULL pointer or inconsistent sizes (need ld >= k, k <= 16256, an 8-byte aligned workspace of "
                        "ppenv_running_mean_std_workspace_bytes() whose first 1024 bytes were zeroed once)");
        return PPENV_EINVAL;
    }
    unsigned int* tickets = reinterpret_cast<unsigned int*>(workspace);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 1024);
    hipLaunchKernelGGL(rms_update_kernel, dim3((m + kRmsRows - 1) / kRmsRows, (k + 63) / 64), dim3(256), 0, (hipStream_t)stream, obs, m, k, ld, partial, tickets, mean, var,
                       count, mean_f32, inv_std_f32, eps);
    if (hipGetLastError() != hipSuccess) { ppenv_set_error("launching rms_update_kernel failed"); return PPENV_EHIP; }
    return PPENV_OK;
}
