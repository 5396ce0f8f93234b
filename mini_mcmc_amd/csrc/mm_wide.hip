/*
 * mm_wide.hip -- HMC with one chain per WORKGROUP: the coordinates of a chain across the lanes.
 *
 * The reference's own timing tests are few chains of a huge dimension (hmc.rs:882-916: 6 chains, D = 10 000,
 * RosenbrockND, eps = 0.01, L = 50, run(100, 100)).  One chain per lane (every other sampling kernel of this engine) gives
 * such a run six busy lanes; here a chain is the work of one workgroup of up to 1024 threads:
 *   * thread t owns the K consecutive coordinates t K .. t K + K - 1 (K = 4, 8, 16 or 32: the smallest for which
 *     ceil(D / K) threads fit a workgroup) of position, momentum and gradient, in registers for the whole launch;
 *   * RosenbrockND couples neighbours (t_j = x_{j+1} - x_j^2): a thread's first and last coordinate go through LDS to its
 *     neighbours, ONE workgroup barrier per gradient evaluation (two buffers in turn, so one barrier suffices); the
 *     Gaussian targets are separable and need none;
 *   * the three sums of a transition (kinetic energy before and after, log-density of the end point) are reduced ONCE,
 *     at its end: per thread in coordinate order, per wave by a butterfly, across the waves in wave order by every
 *     thread itself (a fixed order: results are reproducible, and independent of the number of chains);
 *   * noise: the engine's stream (mm_rng.h) -- the momentum of coordinate j is what every other kernel draws for it
 *     (block j / 4 of (chain, iteration) for f32, Box-Muller pair j / 2 for f64), the accept uniform is thread 0's;
 *   * the transition is mm_hmc_step_noise's (merged half-kicks, accept iff H0 - H1 >= ln u; hmc.rs:304-431), coordinate
 *     for coordinate the arithmetic of mm_targets.h; what differs from the one-chain-per-lane kernels is the ORDER of
 *     the three sums, i.e. rounding (tests/test_wide_hmc.py: f64 against oracle/mh_hmc.c and against the run-time-D kernel);
 *   * a collected state leaves as K contiguous elements per thread, a wave's stores contiguous.
 * Time per leapfrog step: a barrier and ~15 instructions per coordinate -- latency, not bandwidth: the sample is all the
 * HBM traffic there is.
 */
#include "mm_wide.h"

#include "mm_samplers.h"

namespace {

template <class T> __device__ __forceinline__ T wave_sum_t(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

/* momentum of this thread's K coordinates (first coordinate j0, a multiple of 4) for (chain, iteration) */
template <int K> __device__ __forceinline__ void wide_noise(unsigned long long seed, unsigned long long chain, unsigned int iter, int j0, float *z)
{
    const mm_icdf_global tab;
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)(j0 / 4 + q));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            z[4 * q + i] = mm_icdf_f32(blk.w[i], tab);
    }
}
template <int K> __device__ __forceinline__ void wide_noise(unsigned long long seed, unsigned long long chain, unsigned int iter, int j0, double *z)
{
#pragma unroll
    for (int q = 0; q < K / 2; ++q) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)(j0 / 2 + q));
        mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z[2 * q], &z[2 * q + 1]);
    }
}
__device__ __forceinline__ float wide_ln_u(unsigned long long seed, unsigned long long chain, unsigned int iter, float)
{
    return mm_lnu_f32(mm_spare_u24(mm_block(seed, chain, iter, 0u)), mm_icdf_global());
}
__device__ __forceinline__ double wide_ln_u(unsigned long long seed, unsigned long long chain, unsigned int iter, double)
{
    return mm_log(mm_aux_u53(seed, chain, iter, 0));
}

template <class T, int K>
__global__ __launch_bounds__(1024) void mm_hmc_wide_kernel(const mm_wide_args<T> a)
{
    __shared__ T halo[2][2][1024]; /* [buffer][first / last coordinate][thread] */
    __shared__ T part[2][3][16];   /* [buffer][sum][wave] */
    __shared__ T shared_ln_u[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = (blockDim.x + 63) >> 6;
    const int D = a.dim, j0 = tid * K;
    const unsigned long long c = blockIdx.x, chain = a.chain_offset + c;
    const bool rosen = a.kind == MM_ROSENBROCK_ND;
    const T inv_var = a.kind == MM_ISOTROPIC_GAUSSIAN ? a.P.p[0] : T(1);
    T x[K], g[K], p[K], xn[K], gn[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        x[k] = j0 + k < D ? a.state[c * D + j0 + k] : T(0);
    int hb = 0, pb = 0; /* LDS buffers in turn */

    /* gradient of v[] into gv[] and, if want_lp, this thread's share of the log-density; one barrier for RosenbrockND */
    auto grad = [&](const T(&v)[K], T(&gv)[K], bool want_lp) -> T {
        T acc = 0;
        if (rosen) {
            halo[hb][0][tid] = v[0];
            halo[hb][1][tid] = v[K - 1];
            __syncthreads();
            const T left = tid > 0 ? halo[hb][1][tid - 1] : T(0);
            const T right = tid + 1 < (int)blockDim.x ? halo[hb][0][tid + 1] : T(0);
            hb ^= 1;
            /* t_{j-1} of the first coordinate, from the left neighbour's last */
            T tprev = mm_fma(-left, left, v[0]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = j0 + k;
                const T xi = v[k], xr = k + 1 < K ? v[k + 1 < K ? k + 1 : k] : right;
                const T t = mm_fma(-xi, xi, xr);
                const T u = T(1) - xi;
                const T a400 = mm_fma(T(400) * xi, t, T(2) * u);
                T gj;
                if (j + 1 < D) {
                    gj = j > 0 ? mm_fma(T(-200), tprev, a400) : a400;
                    if (want_lp) {
                        acc = mm_fma(T(100) * t, t, acc);
                        acc = mm_fma(u, u, acc);
                    }
                } else {
                    gj = j == D - 1 && D >= 2 ? T(-200) * tprev : T(0);
                }
                gv[k] = j < D ? gj : T(0);
                tprev = t;
            }
            return -acc;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool in = j0 + k < D;
            gv[k] = in ? -(v[k] * inv_var) : T(0);
            if (want_lp && in)
                acc = mm_fma(v[k], v[k], acc);
        }
        return T(-0.5) * (acc * inv_var);
    };
    /* the sums s0, s1, s2 over the workgroup, every thread ending with the same three numbers; thread 0 also publishes u */
    auto reduce3 = [&](T &s0, T &s1, T &s2, T *ln_u_io) {
        s0 = wave_sum_t(s0);
        s1 = wave_sum_t(s1);
        s2 = wave_sum_t(s2);
        if (lane == 0) {
            part[pb][0][wave] = s0;
            part[pb][1][wave] = s1;
            part[pb][2][wave] = s2;
        }
        if (tid == 0 && ln_u_io)
            shared_ln_u[pb] = *ln_u_io;
        __syncthreads();
        T t0 = 0, t1 = 0, t2 = 0;
        for (int w = 0; w < n_waves; ++w) {
            t0 += part[pb][0][w];
            t1 += part[pb][1][w];
            t2 += part[pb][2][w];
        }
        if (ln_u_io)
            *ln_u_io = shared_ln_u[pb];
        pb ^= 1;
        s0 = t0;
        s1 = t1;
        s2 = t2;
    };

    T lp;
    {
        T s0 = grad(x, g, true), s1 = 0, s2 = 0;
        reduce3(s0, s1, s2, nullptr);
        lp = s0;
    }
    const T eps = a.eps, h = eps * T(0.5);
    unsigned long long n_acc = 0;
    unsigned int it = a.iter0;
    const unsigned int total = a.n_discard + a.n_collect;
    for (unsigned int t = 0; t < total; ++t, ++it) {
        wide_noise<K>(a.seed, chain, it, j0, p);
        T ln_u = tid == 0 ? wide_ln_u(a.seed, chain, it, T(0)) : T(0);
        T ke0 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (j0 + k >= D)
                p[k] = T(0);
            ke0 = mm_fma(p[k], p[k], ke0);
            xn[k] = x[k];
            gn[k] = g[k];
        }
        T lpn_part = 0;
        if (a.n_leapfrog > 0) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                p[k] = mm_fma(h, gn[k], p[k]);
            for (int l = 0; l < a.n_leapfrog; ++l) {
                const bool last = l + 1 == a.n_leapfrog;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    xn[k] = mm_fma(eps, p[k], xn[k]);
                lpn_part = grad(xn, gn, last);
                const T kk = last ? h : eps;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    p[k] = mm_fma(kk, gn[k], p[k]);
            }
        }
        T ke1 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k)
            ke1 = mm_fma(p[k], p[k], ke1);
        reduce3(ke0, ke1, lpn_part, &ln_u);
        const T lpn = a.n_leapfrog > 0 ? lpn_part : lp;
        const T h_current = ke0 * T(0.5) - lp;
        const T h_proposed = ke1 * T(0.5) - lpn;
        const T accept_logp = h_current - h_proposed;
        const bool acc = accept_logp >= ln_u; /* the same numbers in every thread: a uniform decision */
        if (acc) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                x[k] = xn[k];
                g[k] = gn[k];
            }
            lp = lpn;
            n_acc += 1;
        }
        if (a.out && t >= a.n_discard) {
            T *dst = a.out + (c * a.n_total + a.out_t0 + (t - a.n_discard)) * (size_t)D + j0;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (j0 + k < D)
                    dst[k] = x[k];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (j0 + k < D)
            a.state[c * D + j0 + k] = x[k];
    if (tid == 0) {
        if (a.accept)
            a.accept[c] += n_acc;
        if (a.accept_total && n_acc)
            atomicAdd(a.accept_total, n_acc);
    }
}

template <class T> hipError_t launch(const mm_wide_args<T> &a, hipStream_t stream)
{
    if (a.dim < 1 || a.dim > MM_WIDE_MAX_DIM || !mm_wide_kind_ok(a.kind))
        return hipErrorInvalidValue;
    int K = 4;
    while ((a.dim + K - 1) / K > 1024)
        K *= 2;
    const unsigned int threads = (unsigned int)(((a.dim + K - 1) / K + 63) / 64 * 64);
    const dim3 grid((unsigned int)a.n_chains), block(threads);
    switch (K) {
    case 4:
        hipLaunchKernelGGL((mm_hmc_wide_kernel<T, 4>), grid, block, 0, stream, a);
        break;
    case 8:
        hipLaunchKernelGGL((mm_hmc_wide_kernel<T, 8>), grid, block, 0, stream, a);
        break;
    case 16:
        hipLaunchKernelGGL((mm_hmc_wide_kernel<T, 16>), grid, block, 0, stream, a);
        break;
    default:
        hipLaunchKernelGGL((mm_hmc_wide_kernel<T, 32>), grid, block, 0, stream, a);
        break;
    }
    return hipGetLastError();
}

} // namespace

hipError_t mm_launch_hmc_wide_f32(const mm_wide_args<float> &a, hipStream_t stream) { return launch<float>(a, stream); }
hipError_t mm_launch_hmc_wide_f64(const mm_wide_args<double> &a, hipStream_t stream) { return launch<double>(a, stream); }
