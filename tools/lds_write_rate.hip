// LDS store / load rates by access width on gfx950: bytes per clock per CU for ds_write_b64 / ds_write_b128 / ds_read_b64 /
// ds_read_b128 from W waves per SIMD, every lane on its own contiguous slot (no bank conflicts).  The diagnostics' wave-level
// FFT (mm_stats_fft.h) exchanges through LDS with 8-byte accesses; is a 16-byte layout worth a redesign?  No: stores run at
// 76 - 84 bytes per clock per CU whatever their width (profiles/r6zx_lds_write_rate.log; its first run printed the figures as
// 'per 100 MHz tick' -- they are per shader clock); the read figures of modes 3 / 4 are bounded by the probe's own s_waitcnt.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_write_rate.hip -o /tmp/lds_write_rate && /tmp/lds_write_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *base = lds + wave * 2304; /* 9216 bytes per wave = the FFT's 8 rows of 72 complex */
    f4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { /* 8 x ds_write_b64: row b, column lane (the FFT's pass 1) */
#pragma unroll
            for (int b = 0; b < 8; ++b)
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"((unsigned)((size_t)(base + 2 * lane) - (size_t)lds)), "v"(__builtin_bit_cast(double, v.xy)), "n"(b * 576) : "memory");
        } else if (MODE == 1) { /* 4 x ds_write_b128: the same 4 KB */
#pragma unroll
            for (int b = 0; b < 4; ++b)
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"((unsigned)((size_t)(base + 4 * lane) - (size_t)lds)), "v"(v), "n"(b * 1152) : "memory");
        } else if (MODE == 2) { /* 4 x ds_write2_b64 (two rows per instruction), what the compiler emits today */
#pragma unroll
            for (int b = 0; b < 4; ++b)
                asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"((unsigned)((size_t)(base + 2 * lane + (b >> 1) * 576) - (size_t)lds)),
                             "v"(__builtin_bit_cast(double, v.xy)), "v"(__builtin_bit_cast(double, v.zw)), "n"((b & 1) * 144), "n"((b & 1) * 144 + 72) : "memory");
        } else if (MODE == 3) { /* 8 x ds_read_b64 */
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                double r;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)((size_t)(base + 2 * lane) - (size_t)lds)), "n"(b * 576) : "memory");
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                acc += (float)__double_as_longlong(r);
            }
        } else if (MODE == 4) { /* 4 x ds_read_b128 */
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f4 r;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)((size_t)(base + 4 * lane) - (size_t)lds)), "n"(b * 1152) : "memory");
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                acc += r.x;
            }
        }
        v.x += 1.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE> void run(const char *name, int wps)
{
    const int waves = 4 * wps, block = 64 * waves, grid = 256, iters = 20000;
    unsigned long long *cyc;
    float *sink;
    hipMalloc(&cyc, sizeof(*cyc) * grid * waves);
    hipMalloc(&sink, sizeof(float) * grid * block);
    const size_t ldsb = (size_t)waves * 9216;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    k<MODE><<<grid, block, ldsb>>>(cyc, sink, 100);
    k<MODE><<<grid, block, ldsb>>>(cyc, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * waves);
    hipMemcpy(h.data(), cyc, sizeof(*cyc) * h.size(), hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto c : h)
        mx = std::max(mx, (double)c);
    /* s_memtime counts shader clocks on this part (the 79 B/clk of tools/fft_probe.hip's ds_write2_b64 comes out again) */
    const double bytes_per_cu = (double)iters * waves * 4096.0;
    printf("%-28s %d waves/SIMD: %.1f bytes per clock per CU\n", name, wps, bytes_per_cu / mx);
    hipFree(cyc);
    hipFree(sink);
}

int main()
{
    for (int wps : {1, 2, 3, 4}) {
        run<0>("8 x ds_write_b64", wps);
        run<2>("4 x ds_write2_b64", wps);
        run<1>("4 x ds_write_b128", wps);
        run<3>("8 x ds_read_b64", wps);
        run<4>("4 x ds_read_b128", wps);
    }
    return 0;
}
