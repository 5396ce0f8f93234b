// Where do the cycles of the statistics kernel's wave-level FFT go?  Stand-alone probe of mm_stats_fft.h:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fno-slp-vectorize tools/fft_probe.hip -o /tmp/fft_probe && /tmp/fft_probe
// One workgroup of 64 W threads per CU (W / 4 waves per SIMD), every wave transforms ITERS register-resident inputs of 512
// points; shader cycles by s_memtime.  MODE 0: the three passes with their LDS exchanges (the kernel's inner loop without
// loads, centring and wave sums); MODE 1: the arithmetic alone (every exchange replaced by a register shuffle that costs
// nothing: wrong results, same vector instructions); MODE 2: the exchanges alone (LDS instructions, no butterflies).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../mini_mcmc_amd/csrc/mm_stats_fft.h"

#define SYNC()                                                                                                      \
    do {                                                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                      \
        __builtin_amdgcn_wave_barrier();                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                      \
    } while (0)

template <int MODE, int WPE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k(unsigned long long *cyc, float *sink, int iters, const mm_cx *tw)
{
    constexpr int R1 = 8;
    using pl = mm_fft_plan<R1>;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    mm_cx *lds = reinterpret_cast<mm_cx *>(lds_raw) + wave * pl::LDS_CX;
    mm_cx tw1[R1], tw2[8];
    for (int b = 0; b < R1; ++b)
        tw1[b] = tw[b * 64 + lane];
    for (int g = 0; g < 8; ++g)
        tw2[g] = tw[R1 * 64 + g * 8 + (lane & 7u)];
    float S[pl::J][8] = {};
    mm_cx z[pl::H];
    for (int a = 0; a < pl::H; ++a)
        z[a] = mm_cx{(float)(threadIdx.x + a) * 1e-3f, (float)(a + 1) * 0.5f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            mm_fft_pass1<R1>(z, [&](int b) { return tw1[b]; }, lds, (int)lane);
            SYNC();
            mm_cx v[pl::J][8];
            mm_fft_pass2_load<R1>(v, lds, (int)lane);
            SYNC();
            mm_fft_pass2_store<R1>(v, [&](int g) { return tw2[g]; }, lds, (int)lane);
            SYNC();
            mm_fft_pass3<R1>(lds, (int)lane, S);
            SYNC();
        } else if (MODE == 1) {
            /* arithmetic only: the passes write to / read from a private register "LDS" through identical index maps */
            mm_cx ev[pl::H], od[pl::H], x1[R1];
            mm_fft_turn<R1>(z, od, std::make_integer_sequence<int, pl::H>{});
            for (int a = 0; a < pl::H; ++a)
                ev[a] = z[a];
            mm_dft_dif<pl::H>(ev);
            mm_dft_dif<pl::H>(od);
#pragma unroll
            for (int p = 0; p < pl::H; ++p) {
                const int b0 = 2 * mm_bitrev(p, pl::HB);
                x1[b0] = b0 == 0 ? ev[p] : mm_cx_mul(ev[p], tw1[b0].re, tw1[b0].im);
                x1[b0 + 1] = mm_cx_mul(od[p], tw1[b0 + 1].re, tw1[b0 + 1].im);
            }
            mm_dft_dif<8>(x1);
#pragma unroll
            for (int p = 1; p < 8; ++p)
                x1[p] = mm_cx_mul(x1[p], tw2[p].re, tw2[p].im);
            mm_dft_dif<8>(x1);
#pragma unroll
            for (int p = 0; p < 8; ++p)
                S[0][p] = fmaf(x1[p].im, x1[p].im, fmaf(x1[p].re, x1[p].re, S[0][p]));
        } else {
#pragma unroll
            for (int b = 0; b < R1; ++b)
                lds[b * pl::PITCH + lane] = z[b & 3];
            SYNC();
            mm_cx v[pl::J][8];
            mm_fft_pass2_load<R1>(v, lds, (int)lane);
            SYNC();
#pragma unroll
            for (int g = 0; g < 8; ++g)
                lds[(lane >> 3) * pl::PITCH + 9 * g + (lane & 7)] = v[0][g];
            SYNC();
            mm_cx w[8];
            mm_lds_load8<1>(lds + (lane >> 3) * pl::PITCH + 9 * (lane & 7), w);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                S[0][e] += w[e].re + w[e].im;
            SYNC();
        }
        /* the next input depends on nothing of this transform: consecutive transforms may overlap as in the kernel */
#pragma unroll
        for (int a = 0; a < pl::H; ++a)
            z[a].re += 1e-3f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        cyc[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int j = 0; j < pl::J; ++j)
        for (int h = 0; h < 8; ++h)
            s += S[j][h];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WPE> void run(const char *name, const mm_cx *tw)
{
    const int iters = 2000, block = 256 * WPE, grid = 256;
    unsigned long long *cyc;
    float *sink;
    hipMalloc(&cyc, (size_t)grid * (block / 64) * 8);
    hipMalloc(&sink, (size_t)grid * block * 4);
    const size_t lds = 100 * 1024; /* one workgroup per CU */
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE, WPE><<<grid, block, lds>>>(cyc, sink, 50, tw);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, WPE><<<grid, block, lds>>>(cyc, sink, iters, tw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * (block / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[h.size() / 2] / iters;
    printf("%-28s waves/SIMD=%d  cycles per transform: per wave %.0f, per SIMD %.0f   wall %.3f ms (%.2f GHz implied)\n", name, WPE, per_wave,
           (double)h.back() / iters / WPE, ms, (double)h.back() / (ms * 1e6));
    hipFree(cyc);
    hipFree(sink);
}

int main()
{
    std::vector<mm_cx> h(8 * 64 + 64);
    for (int b = 0; b < 8; ++b)
        for (int l = 0; l < 64; ++l) {
            const double th = 6.283185307179586 * ((l * b) % 512) / 512.0;
            h[b * 64 + l] = mm_cx{(float)cos(th), (float)-sin(th)};
        }
    for (int g = 0; g < 8; ++g)
        for (int e = 0; e < 8; ++e) {
            const double th = 6.283185307179586 * (e * g) / 64.0;
            h[8 * 64 + g * 8 + e] = mm_cx{(float)cos(th), (float)-sin(th)};
        }
    mm_cx *tw;
    hipMalloc(&tw, h.size() * sizeof(mm_cx));
    hipMemcpy(tw, h.data(), h.size() * sizeof(mm_cx), hipMemcpyHostToDevice);
    run<0, 1>("passes + LDS exchanges", tw);
    run<0, 2>("passes + LDS exchanges", tw);
    run<0, 3>("passes + LDS exchanges", tw);
    run<0, 4>("passes + LDS exchanges", tw);
    run<1, 1>("arithmetic alone", tw);
    run<1, 2>("arithmetic alone", tw);
    run<1, 3>("arithmetic alone", tw);
    run<1, 4>("arithmetic alone", tw);
    run<2, 1>("exchanges alone", tw);
    run<2, 2>("exchanges alone", tw);
    run<2, 3>("exchanges alone", tw);
    run<2, 4>("exchanges alone", tw);
    return 0;
}
