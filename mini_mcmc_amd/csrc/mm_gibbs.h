/*
 * mm_gibbs.h -- Gibbs sweeps with built-in full conditionals (host + device, one definition).
 *
 * GibbsMarkovChain::step (gibbs.rs:89-105) updates every coordinate in turn from `Conditional::sample(i, &state)`.
 * A user-written `Conditional` cannot cross to the GPU; the one the reference itself ships -- the two-component
 * Gaussian mixture over the state [x, z] of gibbs.rs:231-285 (tests) and examples/mixture_gibbs.rs:24-76 -- is built in:
 *     i = 0:  x | z ~ N(mu_z, sigma_z^2)            (Normal::sample = mean + std_dev * StandardNormal)
 *     i = 1:  z | x = 1 with probability p1 / (p0 + p1),  p0 = pi0 N(x; mu0, sigma0^2),  p1 = (1 - pi0) N(x; mu1, sigma1^2)
 *             (0.5 if p0 + p1 == 0), drawn as `random::<f64>() < prob_z1`
 * with normal_pdf exactly as written there: coeff = 1 / sqrt(2 pi var), exp(-(x - mu)^2 / (2 var)).
 * Stream (mm_rng.h): per (chain, iteration) the normal is z[0] of the f64 block 0 and the uniform is auxiliary draw 0.
 * Arithmetic: the engine's (mm_exp, explicit fma for mean + std * z).
 */
#ifndef MM_GIBBS_H
#define MM_GIBBS_H

#include "mm_rng.h"

struct mm_mixture_params {
    double mu0, sigma0, mu1, sigma1, pi0;
};

MM_HD double mm_mixture_normal_pdf(double x, double mu, double sigma)
{
    const double var = sigma * sigma;
    const double coeff = 1.0 / sqrt(2.0 * 3.14159265358979323846 * var);
    const double d = x - mu;
#if defined(__HIP_DEVICE_COMPILE__)
    const double exp_val = mm_exp_sel(-(d * d) / (2.0 * var)); /* same bits, no branches (mm_math.h) */
#else
    const double exp_val = mm_exp(-(d * d) / (2.0 * var));
#endif
    return coeff * exp_val;
}

/* the two draws of a sweep: a pure function of (seed, chain, iteration) */
MM_HD void mm_gibbs_mixture_noise(uint64_t seed, uint64_t chain, uint32_t iter, double *z0, double *u)
{
    const mm_u32x4 b = mm_block(seed, chain, iter, 0u);
    double z1;
    mm_box_muller_f64(mm_u53(b.w[0], b.w[1]), mm_u53(b.w[2], b.w[3]), z0, &z1);
    (void)z1;
    *u = mm_aux_u53(seed, chain, iter, 0u);
}

/* one full sweep of the state s = [x, z] given its draws */
MM_HD void mm_gibbs_mixture_step_noise(const mm_mixture_params &P, double *s, double z0, double u)
{
    /* i = 0: x given z */
    s[0] = (s[1] < 0.5) ? fma(P.sigma0, z0, P.mu0) : fma(P.sigma1, z0, P.mu1);
    /* i = 1: z given the new x */
    const double p0 = P.pi0 * mm_mixture_normal_pdf(s[0], P.mu0, P.sigma0);
    const double p1 = (1.0 - P.pi0) * mm_mixture_normal_pdf(s[0], P.mu1, P.sigma1);
    const double total = p0 + p1;
    const double prob_z1 = (total > 0.0) ? p1 / total : 0.5;
    s[1] = (u < prob_z1) ? 1.0 : 0.0;
}

MM_HD void mm_gibbs_mixture_step(const mm_mixture_params &P, double *s, uint64_t seed, uint64_t chain, uint32_t iter)
{
    double z0, u;
    mm_gibbs_mixture_noise(seed, chain, iter, &z0, &u);
    mm_gibbs_mixture_step_noise(P, s, z0, u);
}

#endif /* MM_GIBBS_H */
