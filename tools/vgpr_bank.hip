// Does a VALU instruction whose VGPR operands share a register bank (register number mod 4) issue slower on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/vgpr_bank.hip -o /tmp/vgpr_bank && /tmp/vgpr_bank
// 64 v_fmac_f32 per block with explicit registers; dst cycles over 16 accumulators of one bank, sources fixed.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
// accumulators v16, v20, ... (bank 0) / v17, v21, ... (bank 1) etc.
#define BLOCK(D0, D1, D2, D3, S0, S1)                                                                              \
    "v_fmac_f32 v" #D0 ", v" #S0 ", v" #S1 "\n v_fmac_f32 v" #D1 ", v" #S0 ", v" #S1 "\n"                          \
    "v_fmac_f32 v" #D2 ", v" #S0 ", v" #S1 "\n v_fmac_f32 v" #D3 ", v" #S0 ", v" #S1 "\n"
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

template <int CASE> __global__ void k(float *out, int iters, unsigned long long *cyc)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (CASE == 0) /* no conflict: dst bank 0 (v16, v20, v24, v28), src banks 1, 2 */
            asm volatile(REP16(BLOCK(16, 20, 24, 28, 1, 2)) ::: CLOB);
        if (CASE == 1) /* src0 and src1 in one bank (v1, v5), dst bank 0 */
            asm volatile(REP16(BLOCK(16, 20, 24, 28, 1, 5)) ::: CLOB);
        if (CASE == 2) /* dst in the bank of src0 (bank 1: v17, v21, v25, v29 with v1), src1 bank 2 */
            asm volatile(REP16(BLOCK(17, 21, 25, 29, 1, 2)) ::: CLOB);
        if (CASE == 3) /* all three in one bank */
            asm volatile(REP16(BLOCK(17, 21, 25, 29, 1, 5)) ::: CLOB);
        if (CASE == 4) /* dst rotating over the four banks, src banks 1, 2: what a compiler's consecutive accumulators give */
            asm volatile(REP16(BLOCK(16, 17, 18, 19, 1, 2)) ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = t1 - t0;
    if (iters < 0)
        out[0] = 1.f;
}

template <int CASE> static void run(const char *name)
{
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&cyc, 8 * 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * 4 * wps; /* one-wave workgroups: wps waves per SIMD */
        hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        /* per SIMD: wps waves x iters x 64 instructions in ms at ~2.1 GHz (clock from memtime not used: 100 MHz counter) */
        printf("%-44s waves/SIMD=%d  %.3f ms  => %.2f cycles per instruction per SIMD at 2.1 GHz\n", name, wps, ms,
               ms * 1e-3 * 2.1e9 / ((double)wps * iters * 64));
    }
}

int main()
{
    run<0>("no bank shared");
    run<1>("src0, src1 one bank");
    run<2>("dst in src0's bank");
    run<3>("dst, src0, src1 one bank");
    run<4>("dst rotating over banks (src banks 1, 2)");
    return 0;
}
