// Where do the cycles of the headline HMC kernel go?  Three loops over the same 450 transitions of 65 536 chains
// (3-D Rosenbrock, f32, L = 10): noise only, transitions only (noise replaced by a cheap perturbation), both.
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
                mm_f2 ln = mm_logf2(mm_spare_u24x2(blk));
                lnna = ln[0];
                lnnb = ln[1];
            }
            if (SB) __builtin_amdgcn_sched_barrier(0);
        });
        for (int i = 0; i < D; ++i) { za[i] = zna[i]; zb[i] = znb[i]; }
        lna = lnna; lnb = lnnb;
    }
    o[chain] = sink + x[0] + x[1] + x[2];
}
// experiment: normals by table-driven inverse CDF (one Philox word per normal) instead of Box-Muller
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float icdf1(uint32_t w, const f4 *tab)
{
    uint32_t b = __float_as_uint((float)((w >> 8) | 1u)); // 2m+1, 24 bits: exact
    f4 c = tab[(b >> 19) - (127u << 4)];                  // exponent and top 4 mantissa bits
    float t = __uint_as_float((b & 0x7ffffu) | 0x3f800000u);
    float z = fmaf(fmaf(fmaf(c[3], t, c[2]), t, c[1]), t, c[0]);
    return __uint_as_float(__float_as_uint(z) ^ (w & 0x80000000u));
}
template <int D> __device__ __forceinline__ void icdf_pair(uint64_t seed, uint64_t chain, uint32_t it, const f4 *tab, float *za,
                                                           float *lna, float *zb, float *lnb)
{
    mm_u32x4x2 blk = mm_block_pair(seed, chain, it, 0);
    mm_f2 ln = mm_logf2(mm_spare_u24x2(blk));
    *lna = ln[0]; *lnb = ln[1];
    for (int i = 0; i < D; ++i) { za[i] = icdf1(blk.w[i][0], tab); zb[i] = icdf1(blk.w[i][1], tab); }
}
template <int MODE> __global__ __launch_bounds__(64) void k(float *o, uint64_t seed, unsigned n, float eps)
{
    constexpr int D = 3;
    if (MODE >= 3) {
        __shared__ f4 tab[24 * 16];
        for (int i = threadIdx.x; i < 24 * 16; i += 64) tab[i] = f4{2.5f - 0.1f * (i >> 4), 0.01f, 0.001f, 0.0001f};
        __syncthreads();
        const uint64_t chain = blockIdx.x * 64 + threadIdx.x;
        mm_tparams<float> P{};
        float x[D] = {0.1f * threadIdx.x, 0.2f, 0.3f}, g[D], lp = Tgt::logp_grad(P, x, g);
        float sink = 0, za[D], zb[D], lna, lnb;
        icdf_pair<D>(seed, chain, 0, tab, za, &lna, zb, &lnb);
        for (unsigned t = 0; t < n; t += 2) {
            float zna[D], znb[D], lnna, lnnb;
            icdf_pair<D>(seed, chain, t + 2, tab, zna, &lnna, znb, &lnnb);
            if (MODE == 4) {
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
        return;
    }
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
    const char *name[5] = {"noise only", "transitions only", "both", "icdf noise only", "icdf both"};
    for (int m = 0; m < 5; ++m) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 1) k<1><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 2) k<2><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 3) k<3><<<1024, 64>>>(o, 7, 450, 0.05f);
            if (m == 4) k<4><<<1024, 64>>>(o, 7, 450, 0.05f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        if (0) printf("");
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
