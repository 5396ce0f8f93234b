// VALU issue rate on gfx950 as a function of waves per SIMD and of instruction-level parallelism, with the instruction
// streams written in inline asm (the compiler's SLP vectoriser turned the "independent plain FMAs" of tools/valu_rate.hip
// into v_pk_fma_f32, which made a lone wave look twice as fast as it is) and timed in shader cycles with s_memtime, so the
// numbers do not depend on the clock the chip holds.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
// One workgroup per CU (a 100 KB LDS request keeps a second one out); a workgroup of 64 W threads puts W / 4 waves on
// each SIMD.  Output: cycles per wave-instruction as one wave sees it, and per SIMD (= that / waves per SIMD).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE> __global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink, int iters, float a, float b)
{
    extern __shared__ float lds[];
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    unsigned long long m0 = u0, m1 = u1;
    typedef double d4 __attribute__((ext_vector_type(4)));
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, e0 = x4, e1 = x5, e2 = x6, e3 = x7, da = a, db = b;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 64 independent plain FMAs (8 registers round robin)
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"))
                         REP4(REP4("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) { // 64 dependent plain FMAs
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
        } else if (MODE == 2) { // two interleaved dependent chains
            asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") REP16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                         REP4(REP4("v_fma_f32 %0, %0, %2, %3\n")) : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        } else if (MODE == 3) { // packed FMAs, independent (4 register pairs)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, av = {a, a}, bv = {b, b};
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         REP4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(av), "v"(bv));
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (MODE == 4) { // Philox multiplies: v_mul_hi_u32 + v_mul_lo_u32, independent
            unsigned h0, h1, l0, l1;
            asm volatile(REP16("v_mul_hi_u32 %0, %4, %6\n v_mul_lo_u32 %1, %4, %6\n v_mul_hi_u32 %2, %5, %7\n v_mul_lo_u32 %3, %5, %7\n")
                         REP4("v_mul_hi_u32 %0, %4, %6\n v_mul_lo_u32 %1, %4, %6\n v_mul_hi_u32 %2, %5, %7\n v_mul_lo_u32 %3, %5, %7\n")
                         : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            u0 ^= h0 ^ l1; u1 ^= h1 ^ l0;
        } else if (MODE == 5) { // v_mad_u64_u32 independent (two accumulators)
            asm volatile(REP16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n") REP16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n")
                         REP4(REP4("v_mad_u64_u32 %0, vcc, %2, %3, %0\n")) : "+v"(m0), "+v"(m1) : "v"(u0), "v"(u1) : "vcc");
        } else if (MODE == 7) { // dependent plain FMAs with only the low 32 lanes enabled: does a half-empty wave issue in one pass?
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n" REP64("v_fma_f32 %0, %0, %1, %2\n") REP16("v_fma_f32 %0, %0, %1, %2\n")
                         "s_mov_b64 exec, s[20:21]\n" : "+v"(x0) : "v"(a), "v"(b) : "s20", "s21");
        } else if (MODE == 8) { // f64 FMAs, independent (4 registers)
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         REP4("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));
        } else if (MODE == 9) { // f64 FMAs, dependent chain
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") REP16("v_fma_f64 %0, %0, %1, %2\n") : "+v"(d0) : "v"(da), "v"(db));
        } else if (MODE == 10) { // f64 MFMA 16x16x4, two accumulators, back to back (16 per iteration)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d3, acc1, 0, 0, 0);
            }
        } else if (MODE == 11) { // 16 f64 MFMAs with 4 independent f64 FMAs after each (64 FMAs): do the two pipes overlap?
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(da), "v"(db));
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d3, acc1, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(da), "v"(db));
            }
        } else if (MODE == 12 || MODE == 13) { // 16 f64 MFMAs with 8 independent 32-bit VALU ops after each: integer (12) / f32 FMA (13)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
                if (MODE == 12)
                    asm volatile("v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));
                else
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(d2, d3, acc1, 0, 0, 0);
                if (MODE == 12)
                    asm volatile("v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));
                else
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            }
        } else if (MODE == 14) { // 16 f64 MFMAs with 4 ds_read_b64 after each
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(d0, d1, acc0, 0, 0, 0);
                e0 += ((volatile double *)lds)[threadIdx.x & 63];
                e1 += ((volatile double *)lds)[64 + (threadIdx.x & 63)];
            }
        } else if (MODE == 6) { // integer xor / add, independent
            asm volatile(REP16("v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         REP4("v_xor_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        cyc[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(m0 ^ m1) + lds[0] +
        (float)(d0 + d1 + d2 + d3 + e0 + e1 + e2 + e3 + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3]);
}

template <int MODE> void run(const char *name, int inst_per_iter, int waves_per_simd)
{
    const int iters = 2000, block = 256 * waves_per_simd, grid = 256;
    unsigned long long *cyc;
    float *sink;
    hipMalloc(&cyc, (size_t)grid * (block / 64) * 8);
    hipMalloc(&sink, (size_t)grid * block * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, block, 100 * 1024>>>(cyc, sink, 50, 1.0001f, 1e-4f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block, 100 * 1024>>>(cyc, sink, iters, 1.0001f, 1e-4f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * (block / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double inst = (double)iters * inst_per_iter;
    const double med = (double)h[h.size() / 2] / inst, mx = (double)h.back() / inst;
    printf("%-34s waves/SIMD=%d  per wave: median %.2f max %.2f cycles/instr   per SIMD: %.2f cycles/instr   wall %.3f ms (%.2f GHz implied)\n", name,
           waves_per_simd, med, mx, mx / waves_per_simd, ms, (double)h.back() / (ms * 1e6));
    hipFree(cyc);
    hipFree(sink);
}

int main()
{
    for (int w : {1, 2}) {
        run<0>("v_fma_f32 independent (8 regs)", 192, w);
        run<1>("v_fma_f32 dependent chain", 80, w);
        run<2>("v_fma_f32 two chains interleaved", 80, w);
        run<3>("v_pk_fma_f32 independent", 80, w);
        run<4>("v_mul_hi/lo_u32 independent", 80, w);
        run<5>("v_mad_u64_u32 (2 accumulators)", 80, w);
        run<6>("v_xor/v_add_u32 independent", 80, w);
        run<7>("v_fma_f32 dependent, 32 lanes on", 80, w);
        run<8>("v_fma_f64 independent", 80, w);
        run<9>("v_fma_f64 dependent chain", 80, w);
        run<10>("v_mfma_f64_16x16x4 back to back", 16, w);
        run<11>("16 MFMA f64 + 64 v_fma_f64 (per MFMA)", 16, w);
        run<12>("16 MFMA f64 + 128 int VALU (per MFMA)", 16, w);
        run<13>("16 MFMA f64 + 128 v_fma_f32 (per MFMA)", 16, w);
        run<14>("16 MFMA f64 + 32 LDS f64 reads+adds (per MFMA)", 16, w);
    }
    return 0;
}
