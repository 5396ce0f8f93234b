// Microbenchmark: f64 issue rates on gfx950 -- v_fma_f64 / v_add_f64 / v_mul_f64 / 64-bit select, independent and
// dependent, v_mfma_f64_16x16x4_f64 (dependent accumulate chains, 1 or 2 interleaved), and MFMA with VALU fillers --
// as a function of waves per SIMD.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/f64_rate.hip -o /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ void k(double *out, int iters, double a, double b)
{
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // 8 independent FMAs x 8
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
        } else if (MODE == 1) { // dependent FMA chain
#pragma unroll
            for (int r = 0; r < 64; ++r) x[0] = __builtin_fma(x[0], a, b);
        } else if (MODE == 2) { // independent adds
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = x[i] + a;
        } else if (MODE == 3) { // independent muls
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = x[i] * a;
        } else if (MODE == 4) { // 64-bit selects (2 x v_cndmask_b32 each)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = (x[(i + 1) & 7] > b) ? x[i] : x[(i + 3) & 7];
        } else if (MODE == 5) { // one dependent MFMA chain
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[r & 7], acc0, 0, 0, 0);
        } else if (MODE == 6) { // two interleaved MFMA chains
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[r], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, x[r], acc1, 0, 0, 0);
            }
        } else if (MODE == 7) { // two MFMA chains + 8 independent f64 FMAs per MFMA pair
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
            }
        } else if (MODE == 8) { // two MFMA chains + 24 FMAs per MFMA pair
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int MODE> void run(const char *name, int units_per_iter, int waves_per_simd)
{
    const int iters = 4000;
    int block = 64, grid = 256 * 4 * waves_per_simd;
    double *out; hipMalloc(&out, (size_t)grid * block * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, 50, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double units = (double)iters * units_per_iter;
    double ns = ms * 1e6 / (units * waves_per_simd);
    printf("%-44s waves/SIMD=%d  %8.3f ms  %7.2f cycles per unit per SIMD @2.4GHz\n", name, waves_per_simd, ms, ns * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64 independent (unit=instr)", 64, w);
        run<1>("v_fma_f64 dependent (unit=instr)", 64, w);
        run<2>("v_add_f64 independent", 64, w);
        run<3>("v_mul_f64 independent", 64, w);
        run<4>("f64 select (unit = one 64-bit select)", 64, w);
        run<5>("mfma_f64_16x16x4 one chain (unit=MFMA)", 16, w);
        run<6>("mfma_f64_16x16x4 two chains (unit=MFMA)", 16, w);
        run<7>("2 MFMA + 8 fma_f64 (unit = that group)", 8, w);
        run<8>("2 MFMA + 24 fma_f64 (unit = that group)", 8, w);
    }
    return 0;
}
