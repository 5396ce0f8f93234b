// Where do the cycles of the headline HMC kernel go?  Loops over the same 450 transitions of 65 536 chains (3-D Rosenbrock,
// f32, L = 10): noise only, transitions only (noise replaced by a cheap perturbation), both (noise from the global table),
// and "fused": the next pair's Philox rounds / table look-ups issued between the leapfrog steps (table in LDS), with and
// without sched_barrier fences.  History of the numbers quoted in DESIGN.md 3 / 5.1: with Box-Muller noise (before the
// inversion table, git history) noise 1091 / transitions 759 / both 1823 cycles per transition and wave.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/hmc_split.hip -o /tmp/hmc_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../mini_mcmc_amd/csrc/mm_samplers.h"
using Tgt = mm_target<float, MM_ROSENBROCK_ND, 3>;
// experiment 2: the next pair's Philox rounds / table look-ups issued between the leapfrog steps of the current pair
template <int SB> __global__ __launch_bounds__(64) void kf(float *o, uint64_t seed, unsigned n, float eps)
{
    constexpr int D = 3;
    __shared__ __attribute__((aligned(16))) float tabs[MM_ICDF_ROWS * 4];
    mm_icdf_lds_fill(tabs, threadIdx.x, 64);
    __syncthreads();
    mm_icdf_lds tab;
    tab.tab = (mm_lds_f4_ptr)reinterpret_cast<const mm_v4f *>(tabs);
    const uint64_t chain = blockIdx.x * 64 + threadIdx.x;
    mm_tparams<float> P{};
    float x[D] = {0.1f * threadIdx.x, 0.2f, 0.3f}, g[D], lp = Tgt::logp_grad(P, x, g);
    float sink = 0, za[D], zb[D], lna, lnb;
    mm_draw_noise_pair<D>(seed, chain, 0, za, &lna, zb, &lnb, tab);
    for (unsigned t = 0; t < n; t += 2) {
        float zna[D], znb[D], lnna = 0, lnnb = 0;
        mm_philox_pair st = mm_philox_pair_init(seed, chain, t + 2, 0);
        sink += mm_hmc_step_noise<float, Tgt, 10>(P, eps, 10, x, &lp, g, za, lna, [&](int) {
            mm_philox_pair_round(st);
            if (SB) __builtin_amdgcn_sched_barrier(0);
        });
        const mm_u32x4x2 blk = mm_philox_pair_words(st);
        sink += mm_hmc_step_noise<float, Tgt, 10>(P, eps, 10, x, &lp, g, zb, lnb, [&](int l) {
            if (l < D) {
                zna[l] = mm_icdf_f32(blk.w[l][0], tab);
                znb[l] = mm_icdf_f32(blk.w[l][1], tab);
            }
            if (l == D) {
                const mm_f2 uu = mm_spare_u24x2(blk);
                lnna = mm_lnu_f32(uu[0], tab);
                lnnb = mm_lnu_f32(uu[1], tab);
            }
            if (SB) __builtin_amdgcn_sched_barrier(0);
        });
        for (int i = 0; i < D; ++i) { za[i] = zna[i]; zb[i] = znb[i]; }
        lna = lnna; lnb = lnnb;
    }
    o[chain] = sink + x[0] + x[1] + x[2];
}
template <int MODE> __global__ __launch_bounds__(64) void k(float *o, uint64_t seed, unsigned n, float eps)
{
    constexpr int D = 3;
    const uint64_t chain = blockIdx.x * 64 + threadIdx.x;
    mm_tparams<float> P{};
    float x[D] = {0.1f * threadIdx.x, 0.2f, 0.3f}, g[D], lp = Tgt::logp_grad(P, x, g);
    float sink = 0;
    float za[D] = {0.3f, -0.2f, 0.5f}, zb[D] = {-0.4f, 0.1f, 0.2f}, lna = -0.5f, lnb = -0.7f;
    if (MODE != 1)
        mm_draw_noise_pair<D>(seed, chain, 0, za, &lna, zb, &lnb);
    for (unsigned t = 0; t < n; t += 2) {
        float zna[D], znb[D], lnna, lnnb;
        if (MODE != 1) {
            mm_draw_noise_pair<D>(seed, chain, t + 2, zna, &lnna, znb, &lnnb);
        } else {
            for (int i = 0; i < D; ++i) { zna[i] = -zb[i]; znb[i] = za[i]; }
            lnna = lnb; lnnb = lna;
        }
        if (MODE != 0) {
            float p[D];
            for (int i = 0; i < D; ++i) p[i] = za[i];
            sink += mm_hmc_step_noise<float, Tgt, 10>(P, eps, 10, x, &lp, g, p, lna);
            for (int i = 0; i < D; ++i) p[i] = zb[i];
            sink += mm_hmc_step_noise<float, Tgt, 10>(P, eps, 10, x, &lp, g, p, lnb);
        } else {
            for (int i = 0; i < D; ++i) sink += za[i] + zb[i];
            sink += lna + lnb;
        }
        for (int i = 0; i < D; ++i) { za[i] = zna[i]; zb[i] = znb[i]; }
        lna = lnna; lnb = lnnb;
    }
    o[chain] = sink + x[0] + x[1] + x[2];
}
int main()
{
    float *o; hipMalloc(&o, 65536 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *name[3] = {"noise only", "transitions only", "both"};
    for (int m = 0; m < 3; ++m) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 1) k<1><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 2) k<2><<<1024, 64>>>(o, 7, 450, 0.05f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-18s %.4f ms  = %.0f cycles per transition per wave at 2.4 GHz\n", name[m], best, best * 1e-3 * 2.4e9 / 450);
    }
    for (int m = 0; m < 2; ++m) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (m == 0) kf<0><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 1) kf<1><<<1024, 64>>>(o, 7, 450, 0.05f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("fused, sched_barrier=%d  %.4f ms  = %.0f cycles per transition per wave at 2.4 GHz\n", m, best, best * 1e-3 * 2.4e9 / 450);
    }
    return 0;
}
