/*
 * mm_hmc_lg.h -- HMC on the dense Gaussian target with the lane-group mapping of mm_nuts_lg.h: 16 chains per wave, four
 * lanes per chain, the gradient -A x on v_mfma_f64_16x16x4 (f64) or v_mfma_f32_16x16x4 (f32, second half of the file);
 * device only.
 *
 * One chain per lane keeps x, p, g, the proposal and its gradient in registers and reads A through the scalar cache:
 * at D = 32 in f64 that spills and runs at 0.28 G leapfrog-steps/s for 65 536 chains (tools/hmc_dense_gaussian.py).
 * Here a leapfrog step is 16 MFMAs + ~40 vector instructions for 16 chains.  Transition = HMC::step + leapfrog
 * (hmc.rs:304-377, 397-431) exactly as mm_hmc_step_noise (mm_samplers.h) states it, with the dot products summed as
 * four interleaved partial sums: the host build mm_hmc_step_noise<double, mm_target_gnd_grp4<D>, 0, mm_red_grp4<D>>
 * is this kernel's bit-exact twin (oracle/engine_host.cpp: eh_hmc_grouped_run).
 * Stream (mm_rng.h, f64 schedule): momentum z[d] = element d & 1 of block d >> 1, accept uniform = u53 of block AUX.
 */
#ifndef MM_HMC_LG_H
#define MM_HMC_LG_H

#include "mm_nuts_lg.h"

struct mm_hmc_lg_args {
    const double *mat;            /* precision matrix A, row-major [D, D] */
    double *state;                /* [C, D] */
    double *out;                  /* [C, n_total, D] or NULL */
    unsigned long long *accept;   /* [C] += accepted proposals, or NULL */
    unsigned long long *accept_total; /* += over all chains (wave ballots), or NULL */
    unsigned long long n_chains, seed, chain_offset, n_total;
    unsigned int iter0, n_discard, n_collect, out_t0;
    double eps;
    int n_leapfrog;
};

hipError_t mm_launch_hmc_lg(int dim, const mm_hmc_lg_args &a, hipStream_t stream); /* mm_nuts_lg_inst.hip */

/* the f32 kernel's arguments (same meaning) */
struct mm_hmc_lg32_args {
    const float *mat;
    float *state, *out;
    unsigned long long *accept, *accept_total;
    unsigned long long n_chains, seed, chain_offset, n_total;
    unsigned int iter0, n_discard, n_collect, out_t0;
    float eps;
    int n_leapfrog;
};
hipError_t mm_launch_hmc_lg32(int dim, const mm_hmc_lg32_args &a, hipStream_t stream);

#ifdef MM_HMC_LG_KERNELS
/* the value held by the partner lane group q ^ 1 (lanes 16 apart in the pairs (0,1) and (2,3)): v_permlane16_swap with both
 * operands = x returns {rows (0,0,2,2), rows (1,1,3,3)} of x, so an even row reads the second, an odd row the first */
__device__ __forceinline__ double mm_lg_from_partner(double x, int q)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const unsigned int lo = (unsigned int)__double2loint(x), hi = (unsigned int)__double2hiint(x);
    const u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double from_even = __hiloint2double((int)h[0], (int)l[0]), from_odd = __hiloint2double((int)h[1], (int)l[1]);
    return (q & 1) ? from_even : from_odd;
}

template <int D> __global__ __launch_bounds__(64) void mm_hmc_lg_kernel(const mm_hmc_lg_args a)
{
    constexpr int NS = D / 4;
    mm_lg_lane<D> L; /* only the mapping part (lane, q, Aop) is used */
    L.lane = threadIdx.x & 63;
    L.q = L.lane >> 4;
    const unsigned long long cl = (unsigned long long)blockIdx.x * 16 + (L.lane & 15);
    const bool active = cl < a.n_chains;
    const unsigned long long chain = a.chain_offset + cl;
    mm_lg_load_A<D>(L, a.mat);
    double x[NS], y[NS]; /* y = A x = -gradient (mm_lg_logp_ax) */
#pragma unroll
    for (int s = 0; s < NS; ++s)
        x[s] = active ? a.state[cl * D + 4 * s + L.q] : 0.0;
    double lp = mm_lg_logp_ax<D>(L, x, y);
    const double eps = a.eps, h = eps * 0.5;
    unsigned long long n_acc = 0, wave_acc = 0;
    const unsigned int total = a.n_discard + a.n_collect;
    for (unsigned int t = 0; t < total; ++t) {
        const unsigned int iter = a.iter0 + t;
        double p[NS];
        /* Coordinates 4 s + q and 4 s + (q ^ 1) are the two normals of ONE Box-Muller block (f64 schedule: z[d] = element
         * d & 1 of block d >> 1), held by lane groups q and q ^ 1.  Each of the two evaluates every other block and hands
         * the partner its half (v_permlane16_swap: rows q <-> q ^ 1), instead of both evaluating all of them and throwing
         * one normal away: half the Philox + Box-Muller work, the same values. */
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) {
            const int s_mine = 2 * j + (L.q & 1), d = 4 * s_mine + L.q;
            const mm_u32x4 blk = mm_block(a.seed, chain, iter, (uint32_t)(d >> 1));
            double z0, z1;
            mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
            const double own = (d & 1) ? z1 : z0, other = (d & 1) ? z0 : z1; /* other: the partner's coordinate d ^ 1 */
            const double recv = mm_lg_from_partner(other, L.q);
            p[2 * j] = (L.q & 1) ? recv : own;
            p[2 * j + 1] = (L.q & 1) ? own : recv;
        }
        const double u = mm_aux_u53(a.seed, chain, iter, 0u);
        const double ke = mm_lg_dot<NS>(p, p);
        const double h_current = ke * 0.5 - lp;
        double xn[NS], yn[NS], lpn = lp;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xn[s] = x[s];
            yn[s] = y[s];
        }
        if (a.n_leapfrog > 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p[s] = fma(-h, yn[s], p[s]); /* first half kick: p + h g, g = -y */
        }
        for (int l = 0; l < a.n_leapfrog; ++l) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                xn[s] = fma(eps, p[s], xn[s]);
            lpn = mm_lg_logp_ax<D>(L, xn, yn);
            const double k = (l + 1 == a.n_leapfrog) ? -h : -eps; /* merged half kicks between steps (Q6) */
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p[s] = fma(k, yn[s], p[s]);
        }
        const double kp = mm_lg_dot<NS>(p, p);
        const double h_proposed = kp * 0.5 - lpn;
        const double accept_logp = h_current - h_proposed;
        const bool acc = accept_logp >= mm_log(u); /* hmc.rs:367 `>=`; NaN rejects */
        if (acc) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                x[s] = xn[s];
                y[s] = yn[s];
            }
            lp = lpn;
        }
        n_acc += acc ? 1ull : 0ull;
        wave_acc += (unsigned long long)__popcll(__ballot(acc && active && L.q == 0));
        if (t >= a.n_discard && a.out && active) {
            double *dst = a.out + (cl * a.n_total + a.out_t0 + (t - a.n_discard)) * D;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                dst[4 * s + L.q] = x[s];
        }
    }
    if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            a.state[cl * D + 4 * s + L.q] = x[s];
        if (a.accept && L.q == 0)
            a.accept[cl] += n_acc;
    }
    if (a.accept_total && L.lane == 0 && wave_acc)
        atomicAdd(a.accept_total, wave_acc);
}

/* ---- f32: the same kernel on v_mfma_f32_16x16x4_f32 --------------------------------------------------------------------
 * The f32 instruction returns its result in row blocks (lane group q: rows 4 q .. 4 q + 3 of the tile) where the f64 one
 * interleaves (rows q, q + 4, ...), so here lane (c, q) owns the coordinates coord(q, s) = 16 (s >> 2) + 4 q + (s & 3) of
 * chain c -- slot s = 4 t + r is row 4 q + r of tile t -- and the A operand's columns are loaded in that order; then
 * B operand and result coincide again and A x needs no lane movement.  The four slots of a tile are four consecutive
 * coordinates: one Philox block for their momenta (mm_rng.h, f32 schedule), one 16-byte store for the sample.
 * Host twin: mm_hmc_step_noise<float, mm_target_gnd_blk4<float, D>, 0, mm_red_blk4<float, D>> (mm_targets.h), bit for bit
 * (tools/mfma_f32_check.hip: the instruction accumulates as the in-order fmaf chain). */
__device__ __forceinline__ float mm_lg32_group_sum(float c)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    unsigned int b = __float_as_uint(c);
    u2 r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    c = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    b = __float_as_uint(c);
    r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int NS> __device__ __forceinline__ float mm_lg32_dot(const float *a, const float *b)
{
    float c = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        c = fmaf(a[s], b[s], c);
    return mm_lg32_group_sum(c);
}
typedef float mm_f4v __attribute__((ext_vector_type(4)));
template <int D>
__device__ __forceinline__ float mm_lg32_logp_ax(const float (&Aop)[D / 16][D / 4], const float *x, float *y)
{
    constexpr int NS = D / 4, NT = D / 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mm_f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[t][s], x[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            y[4 * t + r] = acc[r];
    }
    return -0.5f * mm_lg32_dot<NS>(x, y);
}

template <int D> __global__ __launch_bounds__(64) void mm_hmc_lg32_kernel(const mm_hmc_lg32_args a)
{
    constexpr int NS = D / 4, NT = D / 16;
    const int lane = threadIdx.x & 63, q = lane >> 4, c16 = lane & 15;
    const unsigned long long cl = (unsigned long long)blockIdx.x * 16 + c16;
    const bool active = cl < a.n_chains;
    const unsigned long long chain = a.chain_offset + cl;
    const mm_icdf_global tab;
    float Aop[NT][NS]; /* lane (row c16, k index q): A[16 t + c16][coord(q, s)] */
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            Aop[t][s] = a.mat[(size_t)(16 * t + c16) * D + mm_blk4_coord(q, s)];
    float x[NS], y[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        x[s] = active ? a.state[cl * D + mm_blk4_coord(q, s)] : 0.f;
    float lp = mm_lg32_logp_ax<D>(Aop, x, y);
    const float eps = a.eps, h = eps * 0.5f;
    unsigned long long n_acc = 0, wave_acc = 0;
    const unsigned int total = a.n_discard + a.n_collect;
    for (unsigned int t = 0; t < total; ++t) {
        const unsigned int iter = a.iter0 + t;
        float p[NS];
        /* slots 4 tt .. 4 tt + 3 = coordinates 16 tt + 4 q + (0..3) = the four words of block 4 tt + q */
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const mm_u32x4 blk = mm_block(a.seed, chain, iter, (uint32_t)(4 * tt + q));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[4 * tt + r] = mm_icdf_f32(blk.w[r], tab);
        }
        const float u = mm_spare_u24(mm_block(a.seed, chain, iter, 0u)); /* accept uniform: spare bits of block 0 */
        const float ke = mm_lg32_dot<NS>(p, p);
        const float h_current = ke * 0.5f - lp;
        float xn[NS], yn[NS], lpn = lp;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xn[s] = x[s];
            yn[s] = y[s];
        }
        if (a.n_leapfrog > 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p[s] = fmaf(-h, yn[s], p[s]); /* first half kick: p + h g, g = -y */
        }
        for (int l = 0; l < a.n_leapfrog; ++l) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                xn[s] = fmaf(eps, p[s], xn[s]);
            lpn = mm_lg32_logp_ax<D>(Aop, xn, yn);
            const float k = (l + 1 == a.n_leapfrog) ? -h : -eps; /* merged half kicks between steps (Q6) */
#pragma unroll
            for (int s = 0; s < NS; ++s)
                p[s] = fmaf(k, yn[s], p[s]);
        }
        const float kp = mm_lg32_dot<NS>(p, p);
        const float h_proposed = kp * 0.5f - lpn;
        const float accept_logp = h_current - h_proposed;
        const bool acc = accept_logp >= mm_lnu_f32(u, tab); /* hmc.rs:367 `>=`; NaN rejects */
        if (acc) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                x[s] = xn[s];
                y[s] = yn[s];
            }
            lp = lpn;
        }
        n_acc += acc ? 1ull : 0ull;
        wave_acc += (unsigned long long)__popcll(__ballot(acc && active && q == 0));
        if (t >= a.n_discard && a.out && active) {
            float *dst = a.out + (cl * a.n_total + a.out_t0 + (t - a.n_discard)) * D;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const mm_f4v v = {x[4 * tt], x[4 * tt + 1], x[4 * tt + 2], x[4 * tt + 3]};
                *reinterpret_cast<mm_f4v *>(dst + 16 * tt + 4 * q) = v; /* 16-byte aligned: D is a multiple of 16 */
            }
        }
    }
    if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            a.state[cl * D + mm_blk4_coord(q, s)] = x[s];
        if (a.accept && q == 0)
            a.accept[cl] += n_acc;
    }
    if (a.accept_total && lane == 0 && wave_acc)
        atomicAdd(a.accept_total, wave_acc);
}
#endif /* MM_HMC_LG_KERNELS */

#endif /* MM_HMC_LG_H */
