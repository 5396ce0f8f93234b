// Microbenchmark: issue cost of the f64 "special" instructions the NUTS leaf uses besides FMA -- division sequence,
// v_rcp_f64, v_rndne_f64, v_cvt, v_ldexp, compares + 64-bit selects, v_mov_b64, accvgpr moves, permlane swaps,
// v_mad_u64_u32 -- one wave per SIMD, dependent chains where the use is a chain.  Build on the GPU box:
// hipcc --offload-arch=gfx950 -O3 tools/f64_special_rate.hip -o /tmp/f64s
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE> __global__ void k(double *out, int iters, double a, double b)
{
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1.0, x2 = x0 + 2.0, x3 = x0 + 3.0;
    unsigned long long m0 = threadIdx.x + 12345u, m1 = m0 * 3u;
    unsigned int u0 = threadIdx.x, u1 = u0 + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) { // IEEE division, 4 independent
                x0 = a / x0; x1 = a / x1; x2 = a / x2; x3 = a / x3;
            } else if (MODE == 1) { // v_rcp_f64
                x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3);
            } else if (MODE == 2) { // v_rndne_f64
                x0 = __builtin_rint(x0 * a); x1 = __builtin_rint(x1 * a); x2 = __builtin_rint(x2 * a); x3 = __builtin_rint(x3 * a);
            } else if (MODE == 3) { // compare + 64-bit select
                x0 = (x1 < b) ? x2 : x0; x1 = (x2 < b) ? x3 : x1; x2 = (x3 < b) ? x0 : x2; x3 = (x0 < b) ? x1 : x3;
                x0 += a;
            } else if (MODE == 4) { // v_mad_u64_u32 dependent pairs (Philox round shape)
                m0 = (unsigned long long)(unsigned)m0 * 0xD2511F53u + (m1 >> 32); m1 = (unsigned long long)(unsigned)m1 * 0xCD9E8D57u + (m0 >> 32);
                m0 = (unsigned long long)(unsigned)m0 * 0xD2511F53u + (m1 >> 32); m1 = (unsigned long long)(unsigned)m1 * 0xCD9E8D57u + (m0 >> 32);
            } else if (MODE == 5) { // u32 -> f64 converts + ldexp
                x0 = (double)u0 + x0; x1 = (double)u1 + x1; u0 += 3; u1 += 5;
                x2 = __builtin_ldexp(x2, (int)(u0 & 1)); x3 = __builtin_ldexp(x3, (int)(u1 & 1));
            } else if (MODE == 6) { // sqrt f64
                x0 = __builtin_sqrt(x0 + a); x1 = __builtin_sqrt(x1 + a); x2 = __builtin_sqrt(x2 + a); x3 = __builtin_sqrt(x3 + a);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (double)(m0 ^ m1) + (double)(u0 ^ u1);
}

template <int MODE> void run(const char *name, int units_per_iter)
{
    const int iters = 2000;
    int block = 64, grid = 256 * 4;
    double *out; (void)hipMalloc(&out, (size_t)grid * block * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, 20, 1.0000001, 2.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<grid, block>>>(out, iters, 1.0000001, 2.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.3f ms  %7.1f cycles per unit @2.4GHz (one wave per SIMD)\n", name, ms, ms * 1e6 * 2.4 / ((double)iters * units_per_iter));
    (void)hipFree(out);
}

int main()
{
    run<0>("f64 division a / x (unit = one division)", 64);
    run<1>("v_rcp_f64 (unit = instr)", 64);
    run<2>("v_mul_f64 + v_rndne_f64 (unit = pair)", 64);
    run<3>("f64 compare + 64-bit select (unit = one select)", 64);
    run<4>("v_mad_u64_u32 dependent (unit = instr)", 64);
    run<5>("cvt u32->f64 + add, ldexp (unit = 1 of 4 ops)", 64);
    run<6>("f64 sqrt (unit = one sqrt)", 64);
    return 0;
}
