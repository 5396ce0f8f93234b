/*
 * oracle/discrete.c -- TEST INFRASTRUCTURE ONLY (never linked into, loaded by or called from the product).
 *
 * CPU restatement of the reference's Metropolis-Hastings on integer states, i.e. MHMarkovChain::step
 * (metropolis_hastings.rs:303-315) and run_chain (core.rs:55-73) instantiated with the discrete targets and
 * proposals the reference's own tests / examples define:
 *     PoissonDist + PoissonRandomWalk       tests/metrohast_poisson_test.rs:18-85     (kind 16)
 *     BinomialDist + BinomialRandomWalk     tests/metrohast_poisson_test.rs:150-212   (kind 17)
 *     PoissonTarget + NonnegativeProposal   examples/poisson_mh.rs:8-90               (kind 18)
 * Log-densities are evaluated by the formulas as written there (ln k! summed on every call), NOT from a table.
 * Two streams:
 *   reference stream -- per chain SmallRng::seed_from_u64(1 + seed + i) for the accept draw (:187-193), the
 *     proposal's own SmallRng (cloned into every chain, quirk Q1; `set_seed` on the proposal) for
 *     `random_bool(0.5)` = `next_u64() < 2^63` (rand 0.9 Bernoulli); kind 18 draws from thread_rng in the reference
 *     (unseedable): a SmallRng stands in.  Parity at this level is statistical (the reference's tests check
 *     histograms to 0.05): tests/test_discrete.py.
 *   engine stream -- block 0 of Philox(seed; chain, iteration): direction = top bit of word 0, accept uniform =
 *     u53(word 2, word 3), logs by libm (the product's mm_log differs from it by an ulp at most, which moves a decision
 *     only when the ratio is within an ulp of ln u: never observed): states and accept counts must equal the GPU's exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orng.h"

typedef double (*ln_fn)(double);
static double ln_libm(double x) { return log(x); }
static double ln_engine(double x) { return log(x); } /* libm in every mode: the product's mm_log is not consulted */

typedef struct {
    int32_t state;
    rc_rng prop_rng, rng;
    uint64_t n_accept, iter;
} dchain;

typedef struct o_discrete {
    int kind, n, n_chains, engine;
    double lambda, p;
    uint64_t seed, chain_offset;
    dchain *chains;
} o_discrete;

static double ln_factorial(ln_fn ln, int k)
{
    if (k < 2)
        return 0.0;
    double acc = 0.0;
    for (int i = 1; i <= k; ++i)
        acc += ln((double)i);
    return acc;
}

static double target_logp(const o_discrete *s, ln_fn ln, int k)
{
    if (s->kind == 16) { /* metrohast_poisson_test.rs:24-35 */
        if (k < 0)
            return -INFINITY;
        double kf = (double)k;
        return kf * ln(s->lambda) - s->lambda - ln_factorial(ln, k);
    }
    if (s->kind == 17) { /* :156-175 */
        if (k < 0 || k > s->n)
            return -INFINITY;
        double kf = (double)k, nf = (double)s->n;
        double bc = ln_factorial(ln, s->n) - ln_factorial(ln, k) - ln_factorial(ln, s->n - k);
        return bc + kf * ln(s->p) + (nf - kf) * ln(1.0 - s->p);
    }
    /* poisson_mh.rs:19-26 (usize: k >= 0 by type) */
    double kf = (double)k;
    return -s->lambda + kf * ln(s->lambda) - ln_factorial(ln, k);
}

static double nonneg_logq(ln_fn ln, int x, int y)
{
    if (x == 0)
        return y == 1 ? 0.0 : -INFINITY;
    return (y == x + 1 || y + 1 == x) ? ln(0.5) : -INFINITY;
}

o_discrete *o_discrete_create(int kind, const double *params, const int32_t *init, int n_chains)
{
    if (kind < 16 || kind > 18 || !params || !init || n_chains <= 0)
        return NULL;
    o_discrete *s = (o_discrete *)calloc(1, sizeof *s);
    s->kind = kind;
    s->n_chains = n_chains;
    if (kind == 17) {
        s->n = (int)params[0];
        s->p = params[1];
    } else {
        s->lambda = params[0];
    }
    s->chains = (dchain *)calloc((size_t)n_chains, sizeof(dchain));
    for (int i = 0; i < n_chains; ++i) {
        s->chains[i].state = init[i];
        rc_seed_from_u64(&s->chains[i].prop_rng, 0x9e3779b97f4a7c15ULL); /* from_os_rng, cloned into every chain */
        rc_seed_from_u64(&s->chains[i].rng, 0x5eed0000ULL + (uint64_t)i);
    }
    return s;
}

void o_discrete_destroy(o_discrete *s)
{
    if (s) {
        free(s->chains);
        free(s);
    }
}

/* MetropolisHastings::seed :187-193 */
void o_discrete_seed(o_discrete *s, uint64_t seed)
{
    s->engine = 0;
    for (int i = 0; i < s->n_chains; ++i)
        rc_seed_from_u64(&s->chains[i].rng, 1 + seed + (uint64_t)i);
}

/* Proposal::set_seed on the proposal handed to MetropolisHastings::new (every chain gets the same clone) */
void o_discrete_proposal_seed(o_discrete *s, uint64_t seed)
{
    for (int i = 0; i < s->n_chains; ++i)
        rc_seed_from_u64(&s->chains[i].prop_rng, seed);
}

void o_discrete_use_engine_stream(o_discrete *s, uint64_t seed, uint64_t chain_offset)
{
    s->engine = 1;
    s->seed = seed;
    s->chain_offset = chain_offset;
}

static void dstep(o_discrete *s, int ci)
{
    dchain *c = &s->chains[ci];
    ln_fn ln = s->engine ? ln_engine : ln_libm;
    uint32_t blk[4] = {0, 0, 0, 0};
    if (s->engine)
        o_engine_block(s->seed, s->chain_offset + (uint64_t)ci, (uint32_t)c->iter, 0u, blk);
    const int cur = c->state;
    int prop;
    double qf, qb;
    if (s->kind == 18 && cur == 0) {
        prop = 1; /* no draw (poisson_mh.rs:37-39) */
    } else {
        /* random_bool(0.5): Bernoulli p_int = 2^63, sample = next_u64() < p_int */
        int up = s->engine ? (int)(blk[0] >> 31) : (rc_next_u64(&c->prop_rng) < 0x8000000000000000ULL);
        int step = up ? 1 : -1;
        prop = cur + step;
        if (s->kind == 16 && prop < 0)
            prop = 0;
        if (s->kind == 17)
            prop = prop < 0 ? 0 : (prop > s->n ? s->n : prop);
    }
    if (s->kind == 18) {
        qf = nonneg_logq(ln, cur, prop);
        qb = nonneg_logq(ln, prop, cur);
    } else {
        qf = qb = ln(0.5);
    }
    double current_lp = target_logp(s, ln, cur);
    double proposed_lp = target_logp(s, ln, prop);
    double log_accept_ratio = (proposed_lp + qb) - (current_lp + qf);
    double u;
    if (s->engine) {
        uint64_t m = ((uint64_t)blk[2] << 21) | (uint64_t)(blk[3] >> 11);
        u = (double)(m + 1ull) * 0x1.0p-53;
    } else {
        u = rc_uniform_f64(&c->rng);
    }
    if (log_accept_ratio > ln(u)) {
        c->state = prop;
        c->n_accept += 1;
    }
    c->iter += 1;
}

/* ChainRunner::run core.rs:176-186 over run_chain :55-73; out [n_chains, n_collect] */
void o_discrete_run(o_discrete *s, size_t n_collect, size_t n_discard, int32_t *out, uint64_t *accept_counts)
{
    for (int ci = 0; ci < s->n_chains; ++ci) {
        uint64_t before = s->chains[ci].n_accept;
        for (size_t i = 0; i < n_collect + n_discard; ++i) {
            dstep(s, ci);
            if (i >= n_discard && out)
                out[(size_t)ci * n_collect + (i - n_discard)] = s->chains[ci].state;
        }
        if (accept_counts)
            accept_counts[ci] = s->chains[ci].n_accept - before;
    }
}

void o_discrete_state(o_discrete *s, int32_t *out)
{
    for (int i = 0; i < s->n_chains; ++i)
        out[i] = s->chains[i].state;
}
