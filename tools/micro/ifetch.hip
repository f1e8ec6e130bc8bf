// Micro-benchmark: does one wave's straight-line VALU stream run at the issue rate, or at the instruction-fetch rate?
// 4096 independent FMAs (16 accumulators) laid out as straight-line code, with 4-byte (VOP2) and 8-byte (VOP3 / literal)
// encodings, against the same count executed as a 64-instruction loop.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN16(OP) \
    OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) OP(24) OP(25)
#define VOP2(r) "v_fmac_f32_e32 v" #r ", v40, v41\n"
#define VOP3(r) "v_fma_f32 v" #r ", v" #r ", v40, v41\n"
#define LIT(r) "v_fmaak_f32 v" #r ", v" #r ", v40, 0x3f800001\n"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v40", "v41"

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* cyc, int reps) {
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        if (KIND == 0) asm volatile(".rept 256\n" CHAIN16(VOP2) ".endr\n" ::: CLOB);
        if (KIND == 1) asm volatile(".rept 256\n" CHAIN16(VOP3) ".endr\n" ::: CLOB);
        if (KIND == 2) asm volatile(".rept 256\n" CHAIN16(LIT) ".endr\n" ::: CLOB);
        if (KIND == 3) for (int i = 0; i < 64; i++) asm volatile(".rept 4\n" CHAIN16(VOP3) ".endr\n" ::: CLOB);   // 64-instruction loop body
        if (KIND == 4) for (int i = 0; i < 16; i++) asm volatile(".rept 16\n" CHAIN16(VOP3) ".endr\n" ::: CLOB);  // 256-instruction loop body
        if (KIND == 5) for (int i = 0; i < 4; i++) asm volatile(".rept 64\n" CHAIN16(VOP3) ".endr\n" ::: CLOB);   // 1024-instruction loop body
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int blocks) {
    unsigned long long* cyc;
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int reps = 8;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, reps);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, reps);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-58s blocks=%4d: %.2f clk/instr\n", name, blocks, (double)h / (4096.0 * reps));
    hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256, 512, 1024}) {   // 1, 2, 4 waves per CU on a 256-CU part (each wave on a SIMD of its own up to 1024)
        run<0>("straight-line 4096 x v_fmac_f32_e32 (4 B each, 16 KB)", blocks);
        run<1>("straight-line 4096 x v_fma_f32 (8 B each, 32 KB)", blocks);
        run<2>("straight-line 4096 x v_fmaak_f32 literal (8 B each)", blocks);
        run<3>("loop of 64 x v_fma_f32 (512 B body)", blocks);
        run<4>("loop of 256 x v_fma_f32 (2 KB body)", blocks);
        run<5>("loop of 1024 x v_fma_f32 (8 KB body)", blocks);
    }
    return 0;
}
