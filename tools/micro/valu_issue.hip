// Micro-benchmark: cycles per fp32 VALU operation for ONE wave alone on a SIMD (dependent chain, 4 / 8 / 16 independent chains; all 64,
// the low 32 or the low 16 lanes active), and with 256 .. 2048 waves on the chip.
// Built with plain -O3 the independent chains are SLP-packed into v_pk_fma_f32 (check with -S): the figures for >= 2 chains are then
// per fp32 OPERATION, i.e. half the issue interval of the packed instruction (~4.3 cycles).  Built with -fno-slp-vectorize they are
// per instruction.  tools/micro/ifetch.hip pins the encodings with inline assembly.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ __launch_bounds__(64) void chain_kernel(float* out, unsigned long long* cyc, int active, int iters) {
    const int lane = threadIdx.x;
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = 1.0f + lane * 1e-3f + c;
    const float m = 1.0000001f, b = 1e-7f;
    unsigned long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++)
#pragma unroll
                for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fmaf(a[c], m, b);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
#pragma unroll
        for (int c = 0; c < CHAINS; c++) s += a[c];
        out[blockIdx.x * 64 + lane] = s;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
}

// the same with three VGPR source operands per instruction (multiplier and addend differ per lane), and with v_mul + v_add pairs
template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void chain3_kernel(float* out, unsigned long long* cyc, int active, int iters, const float* in) {
    const int lane = threadIdx.x;
    float a[CHAINS], m[CHAINS], b[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c] = 1.0f + lane * 1e-3f + c; m[c] = in[lane + c]; b[c] = in[64 + lane + c]; }
    unsigned long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++)
#pragma unroll
                for (int c = 0; c < CHAINS; c++) {
                    if (KIND == 0) a[c] = __builtin_fmaf(a[c], m[c], b[c]);
                    else if (KIND == 1) a[c] = a[c] * m[c];
                    else a[c] = a[c] + b[c];
                }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
#pragma unroll
        for (int c = 0; c < CHAINS; c++) s += a[c];
        out[blockIdx.x * 64 + lane] = s;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    }
}
template <int CHAINS, int KIND>
void run3(const char* name) {
    float *out, *in; unsigned long long* cyc;
    hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&in, 256 * sizeof(float)); hipMalloc(&cyc, sizeof(unsigned long long));
    float h_in[256];
    for (int i = 0; i < 256; i++) h_in[i] = i < 64 + 16 ? 1.0f + 1e-7f * i : 1e-7f * i;
    hipMemcpy(in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL((chain3_kernel<CHAINS, KIND>), dim3(1), dim3(64), 0, 0, out, cyc, 64, iters, in);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-40s %.2f clk/instr\n", name, (double)h / ((double)iters * 16 * CHAINS));
    hipFree(out); hipFree(in); hipFree(cyc);
}

template <int CHAINS>
void run(const char* name, int active, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int iters = 2000;
    hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, cyc, active, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, cyc, active, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * CHAINS;
    printf("%-28s active=%2d blocks=%4d: %.2f clk/instr (s_memtime-style counter), %.3f ns/instr wall\n", name, active, blocks, (double)h / n, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int active : {64, 32, 16}) {
        run<1>("dependent chain", active, 1);
        run<4>("4 independent chains", active, 1);
        run<8>("8 independent chains", active, 1);
    }
    run3<1, 0>("v_fma 3 VGPR operands, dependent");
    run3<4, 0>("v_fma 3 VGPR operands, 4 chains");
    run3<8, 0>("v_fma 3 VGPR operands, 8 chains");
    run3<16, 0>("v_fma 3 VGPR operands, 16 chains");
    run3<1, 1>("v_mul 2 VGPR operands, dependent");
    run3<8, 1>("v_mul 2 VGPR operands, 8 chains");
    run3<16, 1>("v_mul 2 VGPR operands, 16 chains");
    run3<8, 2>("v_add 2 VGPR operands, 8 chains");
    run<4>("4 chains, 256 blocks", 64, 256);
    run<4>("4 chains, 1024 blocks", 64, 1024);
    run<4>("4 chains, 2048 blocks", 64, 2048);
    run<4>("4 chains, 2048 blocks", 32, 2048);
    return 0;
}
