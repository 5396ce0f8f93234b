/*
 * mm_generic.h -- MH / HMC transitions for ANY dimension (run-time D), host + device, one definition.
 *
 * The reference's N-dimensional targets take any dimension (RosenbrockND distributions.rs:531-547, IsotropicGaussian
 * :350-402, StandardNormal nuts.rs:1027-1037; hmc.rs:882-916 runs D = 10 000); the fixed-D kernels (mm_kernels.h,
 * mm_split_kernels.h) keep a chain's vectors in registers and exist for D in {1..8, 16, 32}.  Everything else runs
 * here: a chain's vectors (position, gradient, proposal, its gradient, momentum) live in a STORE -- LDS when they
 * fit, else an HBM scratch area, both lane-interleaved so that a wave's accesses are conflict-free / coalesced -- and
 * the arithmetic walks them element by element.
 *
 * Same functions as the fixed-D path, operation for operation: the noise schedule of mm_rng.h, the accept rules and
 * the integrator of mm_samplers.h, the densities of mm_targets.h with the same summation order.  So at a dimension
 * both paths support they give the same bits (tests/test_gpu_parity.py::test_generic_dimension_*), and the host build
 * of this header (oracle/engine_host.cpp) is the device path's bit-exact twin at every dimension.
 */
#ifndef MM_GENERIC_H
#define MM_GENERIC_H

#include "mm_samplers.h"

/* vectors of a chain in the store */
enum { MM_GV_X = 0, MM_GV_G = 1, MM_GV_XN = 2, MM_GV_GN = 3, MM_GV_P = 4, MM_GV_HMC = 5, MM_GV_MH = 3 };

/* element (v, i) of the chain at base[(v * dim + i) * stride]; Ptr: T * (host arrays: stride 1; HBM scratch: stride =
 * padded chain count) or an address-space-3 pointer (LDS: stride 64) */
template <class T, class Ptr> struct mm_gstore {
    Ptr base;
    size_t stride;
    int dim;
    MM_HD T ld(int v, int i) const { return base[((size_t)v * dim + i) * stride]; }
    MM_HD void st(int v, int i, T x) const { base[((size_t)v * dim + i) * stride] = x; }
};

MM_HD bool mm_generic_kind_ok(int kind)
{
    return kind == MM_ISOTROPIC_GAUSSIAN || kind == MM_ROSENBROCK_ND || kind == MM_STANDARD_NORMAL || kind == MM_GAUSSIAN_ND;
}

/* log-density of vector vx and, when vg >= 0, its gradient into vector vg: mm_targets.h element for element */
template <class T, class S> MM_HD T mm_gen_logp_grad(int kind, const mm_tparams<T> &P, const S &s, int vx, int vg)
{
    const int D = s.dim;
    switch (kind) {
    case MM_ISOTROPIC_GAUSSIAN: {
        T q = 0;
        for (int i = 0; i < D; ++i) {
            const T x = s.ld(vx, i);
            q = mm_fma(x, x, q);
            if (vg >= 0)
                s.st(vg, i, -(x * P.p[0]));
        }
        return (T(-0.5) * q) * P.p[0];
    }
    case MM_STANDARD_NORMAL: {
        T q = 0;
        for (int i = 0; i < D; ++i) {
            const T x = s.ld(vx, i);
            q = mm_fma(x, x, q);
            if (vg >= 0)
                s.st(vg, i, -x);
        }
        return T(-0.5) * q;
    }
    case MM_ROSENBROCK_ND: {
        T acc = 0, tprev = 0;
        T xi = s.ld(vx, 0);
        for (int i = 0; i + 1 < D; ++i) {
            const T xn = s.ld(vx, i + 1);
            const T t = mm_fma(-xi, xi, xn);
            const T u = T(1) - xi;
            acc = mm_fma(T(100) * t, t, acc);
            acc = mm_fma(u, u, acc);
            if (vg >= 0) {
                const T a = mm_fma(T(400) * xi, t, T(2) * u);
                s.st(vg, i, (i > 0) ? mm_fma(T(-200), tprev, a) : a);
            }
            tprev = t;
            xi = xn;
        }
        if (vg >= 0)
            s.st(vg, D - 1, D >= 2 ? T(-200) * tprev : T(0));
        return -acc;
    }
    case MM_GAUSSIAN_ND: {
        T q = 0;
        for (int i = 0; i < D; ++i) {
            T y = 0;
            for (int j = 0; j < D; ++j)
                y = mm_fma(P.mat[(size_t)i * D + j], s.ld(vx, j), y);
            q = mm_fma(s.ld(vx, i), y, q);
            if (vg >= 0)
                s.st(vg, i, -y);
        }
        return T(-0.5) * q;
    }
    default:
        return T(MM_NAN_F);
    }
}

/* the noise of (chain, iteration) in the order mm_draw_noise produces it: f(i, z_i) for i = 0 .. D-1; returns ln u */
/* mhp: the MH sampler's paired f32 stream at D <= 2 (mm_rng.h); callers pass sampler == MH && D <= 2 */
template <class F> MM_HD float mm_gen_noise(uint64_t seed, uint64_t chain, uint32_t iter, int D, float, F &&f, bool mhp = false, float *u_out = nullptr)
{
    float ln_u = 0;
    const mm_icdf_global tab;
    if (mhp) {
        const mm_u32x4 blk = mm_mhp_block(seed, chain, iter);
        const uint32_t h = iter & 1u;
        const uint32_t wa = h ? blk.w[2] : blk.w[0], wb = h ? blk.w[3] : blk.w[1];
        f(0, mm_icdf_f32(wa, tab));
        if (D > 1)
            f(1, mm_icdf_f32(wb, tab));
        const float u = mm_mhp_u(mm_mhp_s16(wa, wb), mm_mhp_low_byte(seed, chain, iter));
        if (u_out)
            *u_out = u;
        return mm_lnu_f32(u, tab);
    }
    for (int b = 0; 4 * b < D; ++b) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        if (b == 0)
            ln_u = mm_lnu_f32(mm_spare_u24(blk), tab);
        for (int k = 0; k < 4 && 4 * b + k < D; ++k)
            f(4 * b + k, mm_icdf_f32(blk.w[k], tab));
    }
    return ln_u;
}
template <class F> MM_HD double mm_gen_noise(uint64_t seed, uint64_t chain, uint32_t iter, int D, double, F &&f, bool = false, double * = nullptr)
{
    for (int b = 0; 2 * b < D; ++b) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        double z0, z1;
        mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
        f(2 * b, z0);
        if (2 * b + 1 < D)
            f(2 * b + 1, z1);
    }
    return mm_log(mm_aux_u53(seed, chain, iter, 0));
}

/* mm_mh_step_noise at run-time D: x in MM_GV_X, proposal in MM_GV_XN (the MH store has MM_GV_MH vectors; MM_GV_G unused) */
template <class T, class S>
MM_HD int mm_gen_mh_step(int kind, const mm_tparams<T> &P, T prop_std, const S &s, T *lp, uint64_t seed, uint64_t chain,
                         uint32_t iter)
{
    const int D = s.dim;
    const T ln_u = mm_gen_noise(seed, chain, iter, D, T(0), [&](int i, T z) { s.st(MM_GV_XN, i, mm_fma(prop_std, z, s.ld(MM_GV_X, i))); },
                                sizeof(T) == 4 && D <= 2);
    const T lpp = mm_gen_logp_grad<T, S>(kind, P, s, MM_GV_XN, -1);
    const T log_accept_ratio = lpp - *lp;
    const int acc = log_accept_ratio > ln_u;
    if (acc) {
        for (int i = 0; i < D; ++i)
            s.st(MM_GV_X, i, s.ld(MM_GV_XN, i));
        *lp = lpp;
    }
    return acc;
}

/* mm_hmc_step_noise (sequential reduction order) at run-time D */
template <class T, class S>
MM_HD int mm_gen_hmc_step(int kind, const mm_tparams<T> &P, T eps, int n_leapfrog, const S &s, T *lp, uint64_t seed,
                          uint64_t chain, uint32_t iter)
{
    const int D = s.dim;
    const T ln_u = mm_gen_noise(seed, chain, iter, D, T(0), [&](int i, T z) { s.st(MM_GV_P, i, z); });
    const T h = eps * T(0.5);
    T ke = 0;
    for (int i = 0; i < D; ++i) {
        const T p = s.ld(MM_GV_P, i);
        ke = mm_fma(p, p, ke);
        s.st(MM_GV_XN, i, s.ld(MM_GV_X, i));
        s.st(MM_GV_GN, i, s.ld(MM_GV_G, i));
    }
    const T h_current = ke * T(0.5) - *lp;
    T lpn = *lp;
    if (n_leapfrog > 0) {
        for (int i = 0; i < D; ++i)
            s.st(MM_GV_P, i, mm_fma(h, s.ld(MM_GV_GN, i), s.ld(MM_GV_P, i)));
        for (int l = 0; l < n_leapfrog; ++l) {
            for (int i = 0; i < D; ++i)
                s.st(MM_GV_XN, i, mm_fma(eps, s.ld(MM_GV_P, i), s.ld(MM_GV_XN, i)));
            lpn = mm_gen_logp_grad<T, S>(kind, P, s, MM_GV_XN, MM_GV_GN);
            const T k = (l + 1 == n_leapfrog) ? h : eps;
            for (int i = 0; i < D; ++i)
                s.st(MM_GV_P, i, mm_fma(k, s.ld(MM_GV_GN, i), s.ld(MM_GV_P, i)));
        }
    }
    T kp = 0;
    for (int i = 0; i < D; ++i) {
        const T p = s.ld(MM_GV_P, i);
        kp = mm_fma(p, p, kp);
    }
    const T h_proposed = kp * T(0.5) - lpn;
    const T accept_logp = h_current - h_proposed;
    const int acc = accept_logp >= ln_u;
    if (acc) {
        for (int i = 0; i < D; ++i) {
            s.st(MM_GV_X, i, s.ld(MM_GV_XN, i));
            s.st(MM_GV_G, i, s.ld(MM_GV_GN, i));
        }
        *lp = lpn;
    }
    return acc;
}

#if defined(__HIPCC__)
/* ---------------------------------------------------------------- device side */
template <class T> struct mm_gen_args {
    mm_tparams<T> P;
    int kind, dim, sampler; /* sampler: 0 MH, 1 HMC */
    T scale;
    int n_leapfrog;
    T *state;        /* [n_chains, dim] */
    T *out;          /* [n_chains, n_total, dim] or NULL */
    unsigned long long *accept, *accept_total;
    T *scratch;      /* HBM store: [n_vec * dim][c_pad], or NULL when the store is LDS */
    unsigned long long n_chains, c_pad, seed, chain_offset, n_total;
    unsigned int iter0, n_discard, n_collect, out_t0;
};

typedef __attribute__((address_space(3))) float mm_lds_f32;
typedef __attribute__((address_space(3))) double mm_lds_f64;
template <class T> struct mm_lds_ptr;
template <> struct mm_lds_ptr<float> { typedef mm_lds_f32 *type; };
template <> struct mm_lds_ptr<double> { typedef mm_lds_f64 *type; };

/* one chain per lane, one wave per workgroup, the whole run in one launch; collected states leave as per-lane row
 * stores (a chain's row is dim contiguous elements: the L2 merges the 4- / 8-byte pieces of a line) */
template <class T, bool LDS> __global__ __launch_bounds__(64) void mm_run_generic_kernel(const mm_gen_args<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    const int lane = threadIdx.x & 63;
    const unsigned long long c = (unsigned long long)blockIdx.x * 64ull + lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;
    using Ptr = typename mm_cond<LDS, typename mm_lds_ptr<T>::type, T *>::type;
    mm_gstore<T, Ptr> s;
    if constexpr (LDS) {
        s.base = (Ptr)reinterpret_cast<T *>(mm_lds_raw) + lane;
        s.stride = 64;
    } else {
        s.base = a.scratch + c;
        s.stride = (size_t)a.c_pad;
    }
    s.dim = a.dim;
    const int D = a.dim;
    for (int i = 0; i < D; ++i)
        s.st(MM_GV_X, i, active ? a.state[c * D + i] : T(0));
    T lp = mm_gen_logp_grad<T>(a.kind, a.P, s, MM_GV_X, a.sampler ? MM_GV_G : -1);
    unsigned long long n_acc = 0, wave_acc = 0;
    unsigned int it = a.iter0;
    const unsigned int total = a.n_discard + a.n_collect;
    for (unsigned int t = 0; t < total; ++t, ++it) {
        int acc = a.sampler ? mm_gen_hmc_step<T>(a.kind, a.P, a.scale, a.n_leapfrog, s, &lp, a.seed, chain, it)
                            : mm_gen_mh_step<T>(a.kind, a.P, a.scale, s, &lp, a.seed, chain, it);
        acc = acc && active;
        n_acc += (unsigned long long)acc;
        wave_acc += (unsigned long long)__popcll(__ballot(acc));
        if (a.out && t >= a.n_discard && active) {
            T *dst = a.out + (c * a.n_total + a.out_t0 + (t - a.n_discard)) * D;
            for (int i = 0; i < D; ++i)
                dst[i] = s.ld(MM_GV_X, i);
        }
    }
    if (active) {
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = s.ld(MM_GV_X, i);
        if (a.accept)
            a.accept[c] += n_acc;
    }
    if (a.accept_total && lane == 0 && wave_acc)
        atomicAdd(a.accept_total, wave_acc);
}

/* BatchedGradientTarget::unnorm_logp_batch / unnorm_logp_and_grad at run-time D: x [n, dim] -> logp [n], grad [n, dim] */
template <class T>
__global__ void mm_logp_grad_generic_kernel(const mm_tparams<T> P, int kind, int dim, const T *x, T *logp, T *grad, T *scratch,
                                            unsigned long long n, unsigned long long n_pad)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    mm_gstore<T, T *> s;
    s.base = scratch + (i < n_pad ? i : 0);
    s.stride = (size_t)n_pad;
    s.dim = dim;
    if (i >= n)
        return;
    for (int k = 0; k < dim; ++k)
        s.st(0, k, x[i * dim + k]);
    logp[i] = mm_gen_logp_grad<T>(kind, P, s, 0, grad ? 1 : -1);
    if (grad)
        for (int k = 0; k < dim; ++k)
            grad[i * dim + k] = s.ld(1, k);
}

template <class T>
__global__ void mm_noise_generic_kernel(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                        unsigned long long n, T *z, T *u, int mh)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    /* u itself, not its logarithm, is what mmcmc_draw_noise returns: the same words as mm_draw_noise */
    if (mh && sizeof(T) == 4 && dim <= 2) { /* the MH sampler's paired stream */
        T uu = T(0);
        mm_gen_noise(seed, chain_offset + i, iteration, dim, T(0), [&](int k, T zk) { z[i * dim + k] = zk; }, true, &uu);
        u[i] = uu;
        return;
    }
    mm_gen_noise(seed, chain_offset + i, iteration, dim, T(0), [&](int k, T zk) { z[i * dim + k] = zk; });
    if constexpr (sizeof(T) == 4)
        u[i] = mm_spare_u24(mm_block(seed, chain_offset + i, iteration, 0u));
    else
        u[i] = mm_aux_u53(seed, chain_offset + i, iteration, 0);
}

/* LDS budget of the generic kernel: the store of a wave (n_vec vectors of dim elements for 64 lanes) must leave room
 * for a second workgroup on the CU */
#define MM_GENERIC_LDS_MAX (64u * 1024u)
template <class T> inline size_t mm_generic_store_bytes(int sampler, int dim)
{
    return (size_t)(sampler ? MM_GV_HMC : MM_GV_MH) * (size_t)dim * 64 * sizeof(T);
}

hipError_t mm_launch_run_generic_f32(const mm_gen_args<float> &a, hipStream_t stream);
hipError_t mm_launch_run_generic_f64(const mm_gen_args<double> &a, hipStream_t stream);
hipError_t mm_launch_logp_grad_generic_f32(const mm_tparams<float> &P, int kind, int dim, const float *x, float *logp, float *grad,
                                           float *scratch, unsigned long long n, unsigned long long n_pad, hipStream_t stream);
hipError_t mm_launch_logp_grad_generic_f64(const mm_tparams<double> &P, int kind, int dim, const double *x, double *logp,
                                           double *grad, double *scratch, unsigned long long n, unsigned long long n_pad,
                                           hipStream_t stream);
hipError_t mm_launch_noise_generic_f32(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                       unsigned long long n, float *z, float *u, hipStream_t stream, int mh = 0);
hipError_t mm_launch_noise_generic_f64(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                       unsigned long long n, double *z, double *u, hipStream_t stream, int mh = 0);
#endif /* __HIPCC__ */

#endif /* MM_GENERIC_H */
