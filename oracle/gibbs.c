/*
 * oracle/gibbs.c -- TEST INFRASTRUCTURE ONLY (never linked into, loaded by or called from the product).
 *
 * CPU restatement of the reference's Gibbs sampler instantiated with the Gaussian-mixture conditional of its own tests:
 *     GibbsMarkovChain::step      gibbs.rs:89-105   (every coordinate in turn: state[i] = target.sample(i, &state))
 *     GibbsSampler::{new,set_seed} gibbs.rs:139-187  (every chain gets a CLONE of the conditional)
 *     MixtureConditional           gibbs.rs:231-285  (x | z ~ Normal(mu_z, sigma_z); z | x ~ Bernoulli(p1 / (p0 + p1)))
 *     run through ChainRunner::run core.rs:176-186 / run_chain :55-73
 * Reference stream: the conditional carries its own SmallRng (seeded in the test, gibbs.rs:345), and because the
 * sampler clones the conditional into every chain, ALL chains draw the same numbers and differ only by their start
 * (the analogue of quirk Q1); the chains' own generators (seed + i) are never used by a Gibbs step.  Normal::sample =
 * mean + std_dev * StandardNormal (ziggurat), `random::<f64>() < prob_z1`.
 * Engine stream: block 0 / auxiliary draw 0 of Philox(seed; chain, iteration); libm's exp (the product's mm_exp is not
 * consulted) and mean + std * z evaluated unfused: z must agree exactly, x to 1 ulp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "orng.h"

typedef struct {
    double s[2];
    rc_rng cond_rng; /* the chain's clone of MixtureConditional::rng */
    uint64_t iter;
} gchain;

typedef struct o_gibbs {
    double mu0, sigma0, mu1, sigma1, pi0;
    int n_chains, engine;
    uint64_t seed, chain_offset;
    gchain *chains;
} o_gibbs;

static double normal_pdf(int engine, double x, double mu, double sigma)
{
    double var = sigma * sigma;
    double coeff = 1.0 / sqrt(2.0 * M_PI * var);
    double d = x - mu;
    double e = -(d * d) / (2.0 * var); /* -((x - mu).powi(2)) / (2.0 * var) */
    double exp_val = exp(e); /* libm in every mode */
    (void)engine;
    return coeff * exp_val;
}

o_gibbs *o_gibbs_create(const double *params, const double *init, int n_chains, uint64_t conditional_seed)
{
    if (!params || !init || n_chains <= 0)
        return NULL;
    o_gibbs *s = (o_gibbs *)calloc(1, sizeof *s);
    s->mu0 = params[0];
    s->sigma0 = params[1];
    s->mu1 = params[2];
    s->sigma1 = params[3];
    s->pi0 = params[4];
    s->n_chains = n_chains;
    s->chains = (gchain *)calloc((size_t)n_chains, sizeof(gchain));
    for (int i = 0; i < n_chains; ++i) {
        s->chains[i].s[0] = init[2 * i];
        s->chains[i].s[1] = init[2 * i + 1];
        rc_seed_from_u64(&s->chains[i].cond_rng, conditional_seed); /* identical clone in every chain */
    }
    return s;
}

void o_gibbs_destroy(o_gibbs *s)
{
    if (s) {
        free(s->chains);
        free(s);
    }
}

void o_gibbs_use_engine_stream(o_gibbs *s, uint64_t seed, uint64_t chain_offset)
{
    s->engine = 1;
    s->seed = seed;
    s->chain_offset = chain_offset;
}

static void gstep(o_gibbs *s, int ci)
{
    gchain *c = &s->chains[ci];
    double zn, u;
    if (s->engine) {
        const uint64_t chain = s->chain_offset + (uint64_t)ci;
        zn = o_engine_normal_f64(s->seed, chain, (uint32_t)c->iter, 0u);
        u = o_engine_aux_u53(s->seed, chain, (uint32_t)c->iter, 0u);
    } else {
        zn = rc_standard_normal(&c->cond_rng);
        u = 0.0; /* drawn below, after the normal, in program order */
    }
    /* i = 0 */
    if (c->s[1] < 0.5)
        c->s[0] = s->mu0 + s->sigma0 * zn;
    else
        c->s[0] = s->mu1 + s->sigma1 * zn;
    /* i = 1 */
    double p0 = s->pi0 * normal_pdf(s->engine, c->s[0], s->mu0, s->sigma0);
    double p1 = (1.0 - s->pi0) * normal_pdf(s->engine, c->s[0], s->mu1, s->sigma1);
    double total = p0 + p1;
    double prob_z1 = total > 0.0 ? p1 / total : 0.5;
    if (!s->engine)
        u = rc_uniform_f64(&c->cond_rng);
    c->s[1] = (u < prob_z1) ? 1.0 : 0.0;
    c->iter += 1;
}

void o_gibbs_run(o_gibbs *s, size_t n_collect, size_t n_discard, double *out)
{
    for (int ci = 0; ci < s->n_chains; ++ci)
        for (size_t i = 0; i < n_collect + n_discard; ++i) {
            gstep(s, ci);
            if (i >= n_discard && out) {
                out[((size_t)ci * n_collect + (i - n_discard)) * 2] = s->chains[ci].s[0];
                out[((size_t)ci * n_collect + (i - n_discard)) * 2 + 1] = s->chains[ci].s[1];
            }
        }
}

void o_gibbs_state(o_gibbs *s, double *out)
{
    for (int i = 0; i < s->n_chains; ++i) {
        out[2 * i] = s->chains[i].s[0];
        out[2 * i + 1] = s->chains[i].s[1];
    }
}
