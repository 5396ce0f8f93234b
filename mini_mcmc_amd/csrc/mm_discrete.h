/*
 * mm_discrete.h -- Metropolis-Hastings on integer states (host + device, one definition).
 *
 * The reference's MHMarkovChain is generic over the state type; its discrete users are test / example code:
 *     Poisson(lambda) on i32, +-1 walk reflected at 0       tests/metrohast_poisson_test.rs:18-85
 *     Binomial(n, p) on i32, +-1 walk clamped to [0, n]      tests/metrohast_poisson_test.rs:150-212
 *     Poisson(lambda) on usize, 0 -> 1 always, else +-1      examples/poisson_mh.rs:8-90 (asymmetric proposal)
 * The step is MHMarkovChain::step (metropolis_hastings.rs:303-315) word for word:
 *     log_accept = (lp(prop) + log q(prop -> cur)) - (lp(cur) + log q(cur -> prop));   accept iff log_accept > ln u
 * A log-density of an integer is a table look-up: the table is built once on the host with the reference's operation
 * order (mm_discrete_fill_table) from the engine's mm_log, so the device, the host build and the table are the same
 * bits and states / accept counts are exactly comparable.  Stream (mm_rng.h): block 0 of (chain, iteration);
 * direction = top bit of word 0 (1 -> +1), accept uniform = u53(word 2, word 3).
 */
#ifndef MM_DISCRETE_H
#define MM_DISCRETE_H

#include "mm_rng.h"

#define MM_POISSON_REFLECT 16
#define MM_BINOMIAL_CLAMP 17
#define MM_POISSON_NONNEG 18

/* -inf as a double without <cmath> (this header also compiles under hipRTC) */
#define MM_NEG_INF (-(double)MM_INFINITY_F)

#define MM_DISCRETE_POISSON_TABLE 1024 /* states >= this have log-density -inf (lambda must be far below) */

struct mm_discrete_params {
    int kind;
    int n;                /* Binomial: number of trials */
    int table_len;
    const double *logp;   /* [table_len] unnormalised log-density of state k */
    double ln_half;       /* mm_log(0.5) */
};

/* ln(k!) as the reference sums it: 0 for k < 2, else ln 1 + ln 2 + ... + ln k from the left
 * (metrohast_poisson_test.rs:38-48, poisson_mh.rs:79-89) */
MM_HD double mm_ln_factorial(int k)
{
    if (k < 2)
        return 0.0;
    double acc = 0.0;
    for (int i = 1; i <= k; ++i)
        acc += mm_log((double)i);
    return acc;
}

/* the table of one target; `params`: {lambda} or {n, p}.  Returns the number of entries written (<= cap). */
MM_HD int mm_discrete_fill_table(int kind, const double *params, double *tab, int cap)
{
    if (kind == MM_BINOMIAL_CLAMP) {
        const int n = (int)params[0];
        const double p = params[1], nf = (double)n;
        const double ln_p = mm_log(p), ln_q = mm_log(1.0 - p);
        int len = n + 1 < cap ? n + 1 : cap;
        for (int k = 0; k < len; ++k) {
            const double kf = (double)k;
            /* binomial_coeff_ln(n, k) + kf * p.ln() + (nf - kf) * (1 - p).ln()   (test file :166-175) */
            const double bc = mm_ln_factorial(n) - mm_ln_factorial(k) - mm_ln_factorial(n - k);
            tab[k] = bc + kf * ln_p + (nf - kf) * ln_q;
        }
        return len;
    }
    const double lambda = params[0], ln_l = mm_log(lambda);
    const int len = MM_DISCRETE_POISSON_TABLE < cap ? MM_DISCRETE_POISSON_TABLE : cap;
    for (int k = 0; k < len; ++k) {
        const double kf = (double)k;
        const double lnf = mm_ln_factorial(k);
        if (kind == MM_POISSON_REFLECT)
            tab[k] = kf * ln_l - lambda - lnf; /* test file :31 */
        else
            tab[k] = -lambda + kf * ln_l - lnf; /* poisson_mh.rs:25 */
    }
    return len;
}

MM_HD double mm_discrete_logp(const mm_discrete_params &P, int k)
{
    return (k >= 0 && k < P.table_len) ? P.logp[k] : MM_NEG_INF;
}

/* NonnegativeProposal::logp (poisson_mh.rs:52-70) */
MM_HD double mm_nonneg_logq(const mm_discrete_params &P, int x, int y)
{
    if (x == 0)
        return y == 1 ? 0.0 : MM_NEG_INF;
    return (y == x + 1 || y + 1 == x) ? P.ln_half : MM_NEG_INF;
}

/* the proposal of a chain at `cur` for direction `step` (+1 / -1) */
MM_HD int mm_discrete_propose(const mm_discrete_params &P, int cur, int step)
{
    if (P.kind == MM_POISSON_REFLECT) {
        const int prop = cur + step;
        return prop < 0 ? 0 : prop; /* "reflect instead of going negative" = stay at 0 (:70-76) */
    }
    if (P.kind == MM_BINOMIAL_CLAMP) {
        const int prop = cur + step;
        return prop < 0 ? 0 : (prop > P.n ? P.n : prop);
    }
    return cur == 0 ? 1 : cur + step;
}
/* ... and its log accept ratio (metropolis_hastings.rs:303-310): a function of the state and the direction only, so
 * the split kernel tabulates it per state (mm_discrete.hip) */
MM_HD double mm_discrete_log_ratio(const mm_discrete_params &P, int cur, int step, int *prop_out)
{
    const int prop = mm_discrete_propose(P, cur, step);
    double qf = P.ln_half, qb = P.ln_half;
    if (P.kind != MM_POISSON_REFLECT && P.kind != MM_BINOMIAL_CLAMP) {
        qf = mm_nonneg_logq(P, cur, prop);
        qb = mm_nonneg_logq(P, prop, cur);
    }
    const double current_lp = mm_discrete_logp(P, cur);
    const double proposed_lp = mm_discrete_logp(P, prop);
    *prop_out = prop;
    return (proposed_lp + qb) - (current_lp + qf);
}
/* one transition given its noise block b = mm_block(seed, chain, iteration, 0): direction = top bit of word 0, accept
 * uniform = u53 of words 2, 3; returns 1 iff the proposal was accepted */
MM_HD int mm_discrete_step_block(const mm_discrete_params &P, int32_t *x, const mm_u32x4 &b)
{
    int prop;
    const double log_accept_ratio = mm_discrete_log_ratio(P, *x, (b.w[0] >> 31) ? 1 : -1, &prop);
    const double u = mm_u53(b.w[2], b.w[3]);
    if (mm_ratio_exceeds_ln_u(log_accept_ratio, u)) { /* log_accept_ratio > ln u (metropolis_hastings.rs:311) */
        *x = prop;
        return 1;
    }
    return 0;
}
MM_HD int mm_discrete_step(const mm_discrete_params &P, int32_t *x, uint64_t seed, uint64_t chain, uint32_t iter)
{
    return mm_discrete_step_block(P, x, mm_block(seed, chain, iter, 0u));
}

#endif /* MM_DISCRETE_H */
