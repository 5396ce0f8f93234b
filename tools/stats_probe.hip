// Stand-alone driver of the statistics' half-chain kernel (mm_stats.hip) for A/B timing of its phases:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 [-DMM_STATS_PROBE_SKIP_TILES] [-DMM_STATS_PROBE_SKIP_CENTER] tools/stats_probe.hip -o /tmp/stats_probe
#include "../mini_mcmc_amd/csrc/mm_stats.hip"
#include <cstdio>
#ifndef MM_PROBE_VEC
#define MM_PROBE_VEC true
#endif
int main()
{
    const unsigned long long C = 65536;
    const unsigned int n = 400, D = 3, m = n / 2, n_slabs = 8192; // the largest launch below
    float *x, *means, *ssq, *slabs;
    (void)hipMalloc(&x, C * n * D * 4);
    (void)hipMemset(x, 0, C * n * D * 4);
    (void)hipMalloc(&means, 2 * C * D * 4);
    (void)hipMalloc(&ssq, 2 * C * D * 4);
    (void)hipMalloc(&slabs, (size_t)n_slabs * D * m * 4);
    const size_t pitch = 12 * ((m + 8 + 16 + 8 + 7) / 8) + 4, lds = (D * pitch + D * m + 64 * 16) * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (unsigned int waves : {1024u, 2048u, 4096u, 8192u}) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((mm_half_chain_tile_kernel<float, 8, MM_PROBE_VEC>), dim3(waves), dim3(64), lds, 0, x, C, n, D, m, means, ssq, slabs);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("tile<8> [65536, 400, 3], %u waves: %.4f ms\n", waves, best);
    }
    /* one parameter per wave (mm_half_chain_tile1_kernel<float, 3>): blocks = slabs x D */
    const size_t lds1 = (pitch + m + 64 * 16 + 64) * 4;
    for (unsigned int slabs_n : {683u, 1024u, 1365u, 2048u, 2730u, 4096u}) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((mm_half_chain_tile1_kernel<float, 3, 4>), dim3(slabs_n * D), dim3(64), lds1, 0, x, C, n, D, m, 1u, means, ssq, slabs);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("tile1<3> [65536, 400, 3], %u slabs x 3 waves: %.4f ms\n", slabs_n, best);
    }
    return 0;
}
