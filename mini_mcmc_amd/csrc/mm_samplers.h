/*
 * mm_samplers.h -- one transition of one chain (host + device, one definition).
 *
 *   mm_mh_step   <->  MHMarkovChain::step             metropolis_hastings.rs:303-315
 *                     IsotropicGaussian::sample       distributions.rs:364-372
 *   mm_hmc_step  <->  HMC::step + HMC::leapfrog        hmc.rs:304-377, 397-431   (one row of the batch)
 *
 * What is kept from the reference: the accept rules (MH strict `>`, HMC `>=`, NaN rejects), the leapfrog
 * integrator, H = -logp + 1/2 |p|^2, positions updated only on accept.
 * What is consciously different: the random stream (mm_rng.h instead of rand/burn:
 * Q1, Q2, Q6), log-density of the current state carried instead of recomputed (Q3: same value), the
 * symmetric proposal's q-terms dropped from the MH ratio (Q4: they cancel), Sigma^-1 precomputed (Q5), the two
 * half-kicks between consecutive leapfrog steps merged into one (Q6),
 * gradients analytic, and a*b+c written as ONE fused operation where noted -- so values agree with the
 * reference-ordered oracle to rounding, not bit for bit; tests/test_step_parity.py bounds the difference.
 */
#ifndef MM_SAMPLERS_H
#define MM_SAMPLERS_H

#include "mm_rng.h"
#include "mm_targets.h"

MM_HD float mm_logT(float x) { return mm_logf(x); }
MM_HD double mm_logT(double x) { return mm_log(x); }
MM_HD float mm_expT(float x) { return mm_expf(x); }
MM_HD double mm_expT(double x) { return mm_exp(x); }
/* exp on a kernel's hot path (NUTS: once per leaf): on the device the branch-free forms of mm_math.h, same bits */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float mm_exp_hotT(float x) { return mm_expf_sel(x); }
__device__ __forceinline__ double mm_exp_hotT(double x) { return mm_exp_sel(x); }
#else
inline float mm_exp_hotT(float x) { return mm_expf(x); }
inline double mm_exp_hotT(double x) { return mm_exp(x); }
#endif

/* noise of (chain, iteration): z[0..D) ~ N(0,1) and the accept uniform u in (0,1]  (schedule: mm_rng.h).
 * Tab: where the f32 normal's table is read from (mm_icdf_global by default, mm_icdf_lds in the sampling kernel). */
/* MHP: the MH sampler's paired stream (f32, D <= 2; mm_rng.h: two transitions per Philox block) -- u is the EXACT accept
 * uniform here (both blocks evaluated) */
template <int D, class Tab = mm_icdf_global, bool MHP = false>
MM_HD void mm_draw_noise(uint64_t seed, uint64_t chain, uint32_t iter, float *z, float *u, const Tab &tab = Tab())
{
    if constexpr (MHP) {
        static_assert(D <= 2, "paired MH stream: D <= 2");
        const mm_u32x4 blk = mm_mhp_block(seed, chain, iter);
        const uint32_t h = iter & 1u;
        const uint32_t wa = h ? blk.w[2] : blk.w[0], wb = h ? blk.w[3] : blk.w[1];
        z[0] = mm_icdf_f32(wa, tab);
        if (D > 1)
            z[1] = mm_icdf_f32(wb, tab);
        *u = mm_mhp_u(mm_mhp_s16(wa, wb), mm_mhp_low_byte(seed, chain, iter));
        return;
    }
    MM_UNROLL
    for (int b = 0; b < (D + 3) / 4; ++b) {
        mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        if (b == 0)
            *u = mm_spare_u24(blk);
        MM_UNROLL
        for (int i = 0; i < 4; ++i)
            if (4 * b + i < D)
                z[4 * b + i] = mm_icdf_f32(blk.w[i], tab);
    }
}

template <int D, class Tab = mm_icdf_global, bool MHP = false>
MM_HD void mm_draw_noise(uint64_t seed, uint64_t chain, uint32_t iter, double *z, double *u, const Tab & = Tab())
{
    static_assert(!MHP, "the paired MH stream is f32 only");
    MM_UNROLL
    for (int b = 0; b < (D + 1) / 2; ++b) {
        mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        double z0, z1;
        mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
        z[2 * b] = z0;
        if (2 * b + 1 < D)
            z[2 * b + 1] = z1;
    }
    *u = mm_aux_u53(seed, chain, iter, 0);
}

/* noise of (chain, iter) AND (chain, iter + 1) in one go: za/lna for iter, zb/lnb for iter + 1, ln* = log of the
 * accept uniform.  f32: the two Philox evaluations are interleaved; values are bit-identical to mm_draw_noise +
 * mm_ln_accept.  f64: two scalar evaluations (there is no packed f64 arithmetic to gain from). */
/* LN = false: *lna / *lnb receive the accept uniforms THEMSELVES (for mm_ratio_exceeds_lnu_f32: the MH split kernel) */
/* MHP (the MH sampler's paired stream): LN = true: ln of the exact uniforms (both blocks evaluated); LN = false: *lna / *lnb
 * receive u_hi = (s16 + 1) 2^-16, the upper end of the uniform's interval, for mm_ratio_exceeds_lnu_mhp (no second block).
 * An EVEN iter shares one block between the two transitions; an odd one (wave-uniform, after an odd burn-in) takes two. */
template <int D, class Tab = mm_icdf_global, bool LN = true, bool MHP = false>
MM_HD void mm_draw_noise_pair(uint64_t seed, uint64_t chain, uint32_t iter, float *za, float *lna, float *zb, float *lnb,
                              const Tab &tab = Tab())
{
    if constexpr (MHP) {
        static_assert(D <= 2, "paired MH stream: D <= 2");
        uint32_t a0, a1, b0, b1;
        if ((iter & 1u) == 0u) {
            const mm_u32x4 blk = mm_mhp_block(seed, chain, iter);
            a0 = blk.w[0], a1 = blk.w[1], b0 = blk.w[2], b1 = blk.w[3];
        } else {
            const mm_u32x4 ba = mm_mhp_block(seed, chain, iter), bb = mm_mhp_block(seed, chain, iter + 1u);
            a0 = ba.w[2], a1 = ba.w[3], b0 = bb.w[0], b1 = bb.w[1];
        }
        za[0] = mm_icdf_f32(a0, tab);
        zb[0] = mm_icdf_f32(b0, tab);
        if (D > 1) {
            za[1] = mm_icdf_f32(a1, tab);
            zb[1] = mm_icdf_f32(b1, tab);
        }
        const uint32_t sa = mm_mhp_s16(a0, a1), sb = mm_mhp_s16(b0, b1);
        if (LN) {
            *lna = mm_lnu_f32(mm_mhp_u(sa, mm_mhp_low_byte(seed, chain, iter)), tab);
            *lnb = mm_lnu_f32(mm_mhp_u(sb, mm_mhp_low_byte(seed, chain, iter + 1u)), tab);
        } else {
            *lna = mm_mhp_u_hi(sa);
            *lnb = mm_mhp_u_hi(sb);
        }
        return;
    }
    MM_UNROLL
    for (int b = 0; b < (D + 3) / 4; ++b) {
        mm_u32x4x2 blk = mm_block_pair(seed, chain, iter, (uint32_t)b);
        if (b == 0) {
            const mm_f2 u = mm_spare_u24x2(blk);
            *lna = LN ? mm_lnu_f32(u[0], tab) : u[0];
            *lnb = LN ? mm_lnu_f32(u[1], tab) : u[1];
        }
        MM_UNROLL
        for (int i = 0; i < 4; ++i)
            if (4 * b + i < D) {
                za[4 * b + i] = mm_icdf_f32(blk.w[i][0], tab);
                zb[4 * b + i] = mm_icdf_f32(blk.w[i][1], tab);
            }
    }
}

template <int D, class Tab = mm_icdf_global, bool LN = true, bool MHP = false>
MM_HD void mm_draw_noise_pair(uint64_t seed, uint64_t chain, uint32_t iter, double *za, double *lna, double *zb, double *lnb,
                              const Tab & = Tab())
{
    static_assert(LN && !MHP, "the f64 accept test takes ln u; the paired MH stream is f32 only");
    double u;
    mm_draw_noise<D>(seed, chain, iter, za, &u);
    *lna = mm_log(u);
    mm_draw_noise<D>(seed, chain, iter + 1u, zb, &u);
    *lnb = mm_log(u);
}

/* A target class may carry a proposal of its own: `typedef <functor> proposal;` with
 *     static void sample(T sigma, const T *x, const T *z, T *out);   // x' ~ q(. | x) from z[dim] i.i.d. N(0, 1)
 *     static T logp(T sigma, const T *from, const T *to);            // log q(to | from)
 * -- the open `Proposal` trait (distributions.rs:92-101) as a device functor (mm_rtc.hip: mmcmc_proposal_register_source).
 * No <type_traits>: this header also compiles under hipRTC. */
template <class...> struct mm_void_t {
    typedef void type;
};
template <class Tgt, class = void> struct mm_has_proposal {
    static constexpr bool value = false;
};
template <class Tgt> struct mm_has_proposal<Tgt, typename mm_void_t<typename Tgt::proposal>::type> {
    static constexpr bool value = true;
};

/* One Metropolis-Hastings transition, given its noise: z[D] ~ N(0,1) and ln_u = log of the accept uniform.  x[D],
 * lp = logp(x) are updated in place; returns 1 on accept.
 * Built-in proposal: the isotropic Gaussian random walk x' = x + sigma z, whose q-terms cancel and are dropped (Q4).
 * A target with a `proposal` of its own: MHMarkovChain::step with BOTH q-terms, in the reference's order
 * (metropolis_hastings.rs:303-315): log_accept_ratio = (lp' + log q(x | x')) - (lp + log q(x' | x)). */
/* the accept test as a functor of the log accept ratio: against a given ln u, or against the f32 uniform itself through
 * the table-free filter (mm_rng.h: mm_ratio_exceeds_lnu_f32) -- the same decision */
template <class T> struct mm_accept_ln {
    T ln_u;
    MM_HD bool operator()(T log_accept_ratio) const { return log_accept_ratio > ln_u; }
};
template <class Tab> struct mm_accept_u_f32 {
    float u;
    const Tab &tab;
    MM_HD bool operator()(float log_accept_ratio) const { return mm_ratio_exceeds_lnu_f32(log_accept_ratio, u, tab); }
};
/* ... against the paired stream's u_hi, the uniform's low byte drawn only when the decision needs it (mm_rng.h) */
template <class Tab> struct mm_accept_mhp_f32 {
    float u_hi;
    uint64_t seed, chain;
    uint32_t iter;
    const Tab &tab;
    MM_HD bool operator()(float log_accept_ratio) const
    {
        const uint64_t sd = seed, ch = chain;
        const uint32_t it = iter;
        return mm_ratio_exceeds_lnu_mhp(log_accept_ratio, u_hi, [sd, ch, it]() { return mm_mhp_low_byte(sd, ch, it); }, tab);
    }
};
template <class T, class Tgt, class Accept>
MM_HD int mm_mh_step_accept(const mm_tparams<T> &P, T prop_std, T *x, T *lp, const T *z, const Accept &accept, unsigned int *n_accepted = nullptr)
{
    constexpr int D = Tgt::dim;
    T prop[D];
    T lpp, log_accept_ratio;
    if constexpr (mm_has_proposal<Tgt>::value) {
        Tgt::proposal::sample(prop_std, x, z, prop);
        const T current_lp = *lp; /* carried instead of recomputed: the same value (Q3) */
        lpp = Tgt::logp(P, prop);
        const T log_q_forward = Tgt::proposal::logp(prop_std, x, prop);
        const T log_q_backward = Tgt::proposal::logp(prop_std, prop, x);
        log_accept_ratio = (lpp + log_q_backward) - (current_lp + log_q_forward);
    } else {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            prop[i] = mm_fma(prop_std, z[i], x[i]);
        lpp = Tgt::logp(P, prop);
        log_accept_ratio = lpp - *lp;
    }
    int acc = accept(log_accept_ratio);
    if (acc) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            x[i] = prop[i];
        *lp = lpp;
        if (n_accepted)
            ++*n_accepted; /* the kernels count here: one add under the accept mask instead of a select and an add */
    }
    return acc;
}
template <class T, class Tgt>
MM_HD int mm_mh_step_noise(const mm_tparams<T> &P, T prop_std, T *x, T *lp, const T *z, T ln_u, unsigned int *n_accepted = nullptr)
{
    return mm_mh_step_accept<T, Tgt>(P, prop_std, x, lp, z, mm_accept_ln<T>{ln_u}, n_accepted);
}

template <class T, class Tgt>
MM_HD int mm_mh_step(const mm_tparams<T> &P, T prop_std, T *x, T *lp, uint64_t seed, uint64_t chain, uint32_t iter)
{
    constexpr int D = Tgt::dim;
    T z[D], u;
    mm_draw_noise<D, mm_icdf_global, mm_mh_paired<T, D, 0>::value>(seed, chain, iter, z, &u);
    return mm_mh_step_noise<T, Tgt>(P, prop_std, x, lp, z, mm_ln_accept(u, mm_icdf_global()));
}

/* One HMC transition given its noise: p[D] ~ N(0,1) (momentum; clobbered) and ln_u.  x[D], lp = logp(x),
 * g = grad logp(x) are updated in place; returns 1 on accept. */
struct mm_no_hook {
    MM_HD void operator()(int) const {}
};

/* Hook: called after leapfrog step l = 0, 1, ... with independent work to be issued alongside the integrator's
 * dependent chain; it must not touch the transition's data.  The kernels pass mm_no_hook (nothing): threading the next
 * iterations' Philox rounds through here was measured (tools/hmc_split.hip) and is slower, DESIGN.md 5.1. */
template <class T, class Tgt, int LCT = 0, class Red = mm_red_seq<T, Tgt::dim>, class Hook = mm_no_hook>
MM_HD int mm_hmc_step_noise(const mm_tparams<T> &P, T eps, int n_leapfrog, T *x, T *lp, T *g, T *p, T ln_u,
                            Hook &&hook = Hook(), unsigned int *n_accepted = nullptr)
{
    /* Red: summation order of the two kinetic-energy dot products (mm_targets.h): sequential for one chain per lane,
     * grouped for the lane-group kernel (mm_hmc_lg.h), whose bit-exact host twin this then is */
    /* LCT > 0: the number of leapfrog steps is the compile-time constant LCT (the loop is fully unrolled so the
     * scheduler can overlap it with independent work); LCT == 0: run-time n_leapfrog.  Same arithmetic either way. */
    constexpr int D = Tgt::dim;
    T xn[D], gn[D];
    const T h = eps * T(0.5);
    const T ke = Red::dot(p, p);
    MM_UNROLL
    for (int i = 0; i < D; ++i) {
        xn[i] = x[i];
        gn[i] = g[i];
    }
    T h_current = ke * T(0.5) - *lp;
    T lpn = *lp;
    /* hmc.rs:403-429 kicks the momentum by eps/2 at the start and at the end of every leapfrog step; the end kick of one
     * step and the start kick of the next use the same gradient and are merged here into one kick by eps (SURVEY
     * App. B Q6: allowed on the GPU, the oracle keeps them apart; the results differ by rounding only) */
    auto leap = [&](bool last) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            xn[i] = mm_fma(eps, p[i], xn[i]); /* drift */
        lpn = Tgt::logp_grad(P, xn, gn);
        const T k = last ? h : eps;
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            p[i] = mm_fma(k, gn[i], p[i]);
    };
    if constexpr (LCT > 0) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            p[i] = mm_fma(h, gn[i], p[i]); /* first half kick, with the gradient of the current position */
        MM_UNROLL
        for (int l = 0; l < LCT; ++l) {
            leap(l + 1 == LCT);
            hook(l);
        }
    } else if (n_leapfrog > 0) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            p[i] = mm_fma(h, gn[i], p[i]);
        for (int l = 0; l < n_leapfrog; ++l) {
            leap(l + 1 == n_leapfrog);
            hook(l);
        }
    }
    const T kp = Red::dot(p, p);
    T h_proposed = kp * T(0.5) - lpn;
    T accept_logp = h_current - h_proposed;
    int acc = accept_logp >= ln_u;
    if (acc) {
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            x[i] = xn[i];
            g[i] = gn[i];
        }
        *lp = lpn;
        if (n_accepted)
            ++*n_accepted;
    }
    return acc;
}

template <class T, class Tgt>
MM_HD int mm_hmc_step(const mm_tparams<T> &P, T eps, int n_leapfrog, T *x, T *lp, T *g, uint64_t seed,
                      uint64_t chain, uint32_t iter)
{
    constexpr int D = Tgt::dim;
    T p[D], u;
    mm_draw_noise<D>(seed, chain, iter, p, &u);
    return mm_hmc_step_noise<T, Tgt>(P, eps, n_leapfrog, x, lp, g, p, mm_ln_accept(u, mm_icdf_global()));
}

#endif /* MM_SAMPLERS_H */
