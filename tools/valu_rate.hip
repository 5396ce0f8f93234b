// Microbenchmark: VALU issue rate on gfx950 for plain / packed f32 FMA, 32-bit integer ops and v_mad_u64_u32,
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x2}, p5 = {x3, x4}, p6 = {x5, x6}, p7 = {x7, x0};
    float2v av = {a, a}, bv = {b, b};
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned long long m0 = u0, m1 = u1, m2 = u2, m3 = u3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 8 independent plain FMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) { // 8 independent packed FMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
                p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
                p4 = __builtin_elementwise_fma(p4, av, bv); p5 = __builtin_elementwise_fma(p5, av, bv);
                p6 = __builtin_elementwise_fma(p6, av, bv); p7 = __builtin_elementwise_fma(p7, av, bv);
            }
        } else if (MODE == 2) { // 8 independent xor/add integer ops
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u0 = (u0 ^ u1) + 0x9E3779B9u; u1 = (u1 ^ u2) + 0x9E3779B9u; u2 = (u2 ^ u3) + 0x9E3779B9u; u3 = (u3 ^ u4) + 0x9E3779B9u;
                u4 = (u4 ^ u5) + 0x9E3779B9u; u5 = (u5 ^ u6) + 0x9E3779B9u; u6 = (u6 ^ u7) + 0x9E3779B9u; u7 = (u7 ^ u0) + 0x9E3779B9u;
            }
        } else if (MODE == 3) { // v_mad_u64_u32 (32x32->64 multiply)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                m0 = (unsigned long long)(unsigned)m0 * 0xD2511F53u + (m0 >> 32); m1 = (unsigned long long)(unsigned)m1 * 0xCD9E8D57u + (m1 >> 32);
                m2 = (unsigned long long)(unsigned)m2 * 0xD2511F53u + (m2 >> 32); m3 = (unsigned long long)(unsigned)m3 * 0xCD9E8D57u + (m3 >> 32);
            }
        } else if (MODE == 4) { // dependent chain of plain FMAs (latency)
#pragma unroll
            for (int r = 0; r < 64; ++r) x0 = __builtin_fmaf(x0, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
        (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(m0 ^ m1 ^ m2 ^ m3);
}

template <int MODE> void run(const char *name, int ops_per_iter, int waves_per_simd)
{
    const int iters = 20000;
    int block = 64, grid = 256 * 4 * waves_per_simd; // one wave per workgroup
    float *out; hipMalloc(&out, (size_t)grid * block * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(out, 100, 1.0001f, 1e-4f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(out, iters, 1.0001f, 1e-4f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)iters * ops_per_iter;                 // wave-instructions per wave
    double ns_per_inst_per_simd = ms * 1e6 / (inst * waves_per_simd);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 (independent)", 64, w);
        run<1>("v_pk_fma_f32 (independent)", 64, w);
        run<2>("v_xor+v_add u32", 128, w);
        run<3>("v_mad_u64_u32", 32, w);
        run<4>("v_fma_f32 (dependent chain)", 64, w);
    }
    return 0;
}
