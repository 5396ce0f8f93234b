/*
 * oracle/mh_hmc.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Faithful restatement of
 *   core.rs:394-435                 init / init_det / init_with_seed
 *   core.rs:55-73, 176-186          run_chain, ChainRunner::run
 *   metropolis_hastings.rs:149-193  MetropolisHastings::{new, seed}
 *   metropolis_hastings.rs:303-315  MHMarkovChain::step
 *   distributions.rs:364-386        IsotropicGaussian::{sample, logp}
 *   hmc.rs:87-158, 304-431          HMC::{new, run, step, leapfrog}
 * in the reference's operation order (no fused multiply-add; build with -ffp-contract=off), quirks kept:
 * Q1 cloned proposal generator, Q2 D+1 normals per proposal, Q3 logp(current) recomputed and strict '>',
 * Q4 the odd proposal normaliser, Q6 separate half-kicks and the extra final logp.
 * Elementary functions are libm's in every mode (also on the engine's noise stream): nothing of the product's
 * mm_math.h is consulted here.
 */
#include "oracle.h"


#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 512 /* the reference has no limit; the GPU parity tests go up to 257 */
/* HMC: the reference's own timing test runs D = 10 000 (hmc.rs:882-916); its row buffers live on the stack here */
#define HMC_MAXD 16384

/* ------------------------------------------------------------------ init */

void o_init_with_seed(size_t n, size_t d, uint64_t seed, double *out)
{
    rc_rng r;
    rc_seed_from_u64(&r, seed);
    for (size_t i = 0; i < n * d; ++i)
        out[i] = rc_standard_normal(&r);
}

/* ------------------------------------------------------------------ threads */

typedef void (*range_fn)(void *ctx, int lo, int hi);
typedef struct {
    range_fn fn;
    void *ctx;
    int lo, hi;
} par_job;
static void *par_thread(void *p)
{
    par_job *j = (par_job *)p;
    j->fn(j->ctx, j->lo, j->hi);
    return NULL;
}
/* static block partition over threads: what rayon's par_iter_mut amounts to for equal-cost chains */
static void par_for(range_fn fn, void *ctx, int n, int n_threads)
{
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > n)
        n_threads = n;
    if (n_threads == 1) {
        fn(ctx, 0, n);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    par_job *jobs = (par_job *)malloc(sizeof(par_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].fn = fn;
        jobs[t].ctx = ctx;
        jobs[t].lo = (int)((long)n * t / n_threads);
        jobs[t].hi = (int)((long)n * (t + 1) / n_threads);
        pthread_create(&th[t], NULL, par_thread, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------ MH */

typedef struct {
    double state[MAXD]; /* holds T values (f32 values are exactly representable) */
    o_rng prop_rng;     /* the chain's clone of the proposal's generator (Q1) */
    o_rng rng;          /* accept generator, MHMarkovChain::rng */
    uint64_t n_accept;
    uint64_t iter;
} mh_chain;

struct o_mh {
    ot_target target;
    int dim, n_chains, is_f32, engine;
    /* 0: IsotropicGaussian (distributions.rs:344-392).  1: a custom `impl Proposal` of the kind the reference's own tests
     * write (tests/metrohast_poisson_test.rs:50-85), ASYMMETRIC, so that both q-terms of metropolis_hastings.rs:308-309
     * matter: the multiplicative log-normal random walk on positive states,
     *     sample: x'_i = x_i exp(std z_i);   logp(from, to) = sum_i [ -ln to_i - d_i^2 / (2 std^2) ] - dim (ln std + ln sqrt(2 pi)),
     *     d_i = ln to_i - ln from_i                                   (tests/test_user_proposal.py holds the same as HIP source) */
    int prop_kind;
    double std;
    mh_chain *chains;
};

o_mh *o_mh_create(const ot_target *target, double proposal_std, const double *init, int n_chains, int is_f32)
{
    if (!target || !init || n_chains <= 0 || target->dim <= 0 || target->dim > MAXD)
        return NULL;
    o_mh *s = (o_mh *)calloc(1, sizeof *s);
    s->target = *target;
    s->dim = target->dim;
    s->n_chains = n_chains;
    s->is_f32 = is_f32;
    s->std = is_f32 ? (double)(float)proposal_std : proposal_std;
    s->chains = (mh_chain *)calloc((size_t)n_chains, sizeof(mh_chain));
    for (int i = 0; i < n_chains; ++i) {
        mh_chain *c = &s->chains[i];
        for (int k = 0; k < s->dim; ++k) {
            double v = init[(size_t)i * s->dim + k];
            c->state[k] = is_f32 ? (double)(float)v : v;
        }
        c->prop_rng.is_f32 = is_f32;
        c->rng.is_f32 = is_f32;
        /* from_os_rng() in the reference: proposal clone identical in every chain, accept rng distinct */
        o_rng_init_rand_compat(&c->prop_rng, 0x9e3779b97f4a7c15ULL);
        o_rng_init_rand_compat(&c->rng, 0x5eed0000ULL + (uint64_t)i);
    }
    return s;
}

void o_mh_destroy(o_mh *s)
{
    if (s) {
        free(s->chains);
        free(s);
    }
}

void o_mh_proposal_seed(o_mh *s, uint64_t seed)
{
    s->engine = 0;
    for (int i = 0; i < s->n_chains; ++i)
        o_rng_init_rand_compat(&s->chains[i].prop_rng, seed);
}

void o_mh_seed(o_mh *s, uint64_t seed)
{
    s->engine = 0;
    for (int i = 0; i < s->n_chains; ++i)
        o_rng_init_rand_compat(&s->chains[i].rng, 1 + seed + (uint64_t)i);
}

void o_mh_set_proposal(o_mh *s, int kind) { s->prop_kind = kind; }

void o_mh_use_engine_stream(o_mh *s, uint64_t seed, uint64_t chain_offset)
{
    s->engine = 1;
    for (int i = 0; i < s->n_chains; ++i) {
        o_rng_init_engine_stream(&s->chains[i].prop_rng, seed, chain_offset + (uint64_t)i);
        o_rng_init_engine_stream(&s->chains[i].rng, seed, chain_offset + (uint64_t)i);
        /* the engine's MH sampler draws from its paired stream in f32 at dim <= 2 (csrc/mm_rng.h) */
        o_rng_engine_mh_paired(&s->chains[i].prop_rng, s->dim);
        o_rng_engine_mh_paired(&s->chains[i].rng, s->dim);
    }
}

/* the custom proposal's log q(to | from) (prop_kind 1); libm's log / exp in every mode */
#define MH_LOGNORMAL_LOGQ(REAL, SUF, LN)                                                                     \
    static REAL lognormal_logq_##SUF(const o_mh *s, const REAL *from, const REAL *to)                        \
    {                                                                                                        \
        REAL std = (REAL)s->std, two = (REAL)2, var = std * std, acc = 0;                                    \
        for (int i = 0; i < s->dim; ++i) {                                                                   \
            REAL lt = LN(to[i]);                                                                             \
            REAL lf = LN(from[i]);                                                                           \
            REAL dd = lt - lf;                                                                               \
            acc += -lt - (dd * dd) / (two * var);                                                            \
        }                                                                                                    \
        REAL ls = LN(std);                                                                                   \
        return acc - (REAL)s->dim * (ls + (REAL)0.91893853320467274178);                                     \
    }
MH_LOGNORMAL_LOGQ(float, f32, logf)
MH_LOGNORMAL_LOGQ(double, f64, log)

#define MH_STEP(REAL, SUF, LOGP, LN, EXP)                                                                    \
    static void mh_step_##SUF(o_mh *s, mh_chain *c)                                                          \
    {                                                                                                        \
        int d = s->dim;                                                                                      \
        REAL cur[MAXD], prop[MAXD];                                                                          \
        for (int i = 0; i < d; ++i)                                                                          \
            cur[i] = (REAL)c->state[i];                                                                      \
        c->prop_rng.begin_iter(&c->prop_rng, c->iter);                                                       \
        c->rng.begin_iter(&c->rng, c->iter);                                                                 \
        REAL std = (REAL)s->std;                                                                             \
        if (s->prop_kind == 1) {                                                                             \
            /* MHMarkovChain::step metropolis_hastings.rs:303-315 with the custom proposal above */          \
            for (int i = 0; i < d; ++i) {                                                                    \
                REAL z = (REAL)c->prop_rng.normal_f64(&c->prop_rng);                                         \
                REAL e = EXP(std * z);                                                                       \
                prop[i] = cur[i] * e;                                                                        \
            }                                                                                                \
            REAL current_lp = LOGP(&s->target, cur);                                                         \
            REAL proposed_lp = LOGP(&s->target, prop);                                                       \
            REAL lqf = lognormal_logq_##SUF(s, cur, prop);                                                   \
            REAL lqb = lognormal_logq_##SUF(s, prop, cur);                                                   \
            REAL log_accept_ratio = (proposed_lp + lqb) - (current_lp + lqf);                                \
            REAL u = (REAL)c->rng.accept_uniform(&c->rng);                                                   \
            REAL ln_u = LN(u); /* libm in every mode */                                                      \
            if (log_accept_ratio > ln_u) {                                                                   \
                for (int i = 0; i < d; ++i)                                                                  \
                    c->state[i] = (double)prop[i];                                                           \
                c->n_accept += 1;                                                                            \
            }                                                                                                \
            c->iter += 1;                                                                                    \
            return;                                                                                          \
        }                                                                                                    \
        /* IsotropicGaussian::sample distributions.rs:364-372 : Normal(0,std).sample = 0 + std*z ; + current */ \
        for (int i = 0; i < d; ++i) {                                                                        \
            REAL z = (REAL)c->prop_rng.normal_f64(&c->prop_rng);                                             \
            REAL x = (REAL)0 + std * z;                                                                      \
            prop[i] = x + cur[i];                                                                            \
        }                                                                                                    \
        if (!s->engine) /* Q2: Zip pulls one more normal before it sees `current` is exhausted */           \
            (void)c->prop_rng.normal_f64(&c->prop_rng);                                                      \
        REAL current_lp = LOGP(&s->target, cur);                                                             \
        REAL proposed_lp = LOGP(&s->target, prop);                                                           \
        /* IsotropicGaussian::logp distributions.rs:374-386 (forward: from=cur,to=prop) */                   \
        REAL two = (REAL)2, var = std * std;                                                                 \
        REAL lqf = 0, lqb = 0;                                                                               \
        for (int i = 0; i < d; ++i) {                                                                        \
            REAL diff = prop[i] - cur[i];                                                                    \
            lqf += -(diff * diff) / (two * var);                                                             \
        }                                                                                                    \
        for (int i = 0; i < d; ++i) {                                                                        \
            REAL diff = cur[i] - prop[i];                                                                    \
            lqb += -(diff * diff) / (two * var);                                                             \
        }                                                                                                    \
        REAL norm = -(REAL)d * (REAL)0.5 * LN(var * (REAL)M_PI * std * std);                                 \
        lqf += norm;                                                                                         \
        lqb += norm;                                                                                         \
        REAL log_accept_ratio = (proposed_lp + lqb) - (current_lp + lqf);                                    \
        REAL u = (REAL)c->rng.accept_uniform(&c->rng);                                                       \
        REAL ln_u = LN(u); /* libm in every mode */                                                          \
        if (log_accept_ratio > ln_u) {                                                                       \
            for (int i = 0; i < d; ++i)                                                                      \
                c->state[i] = (double)prop[i];                                                               \
            c->n_accept += 1;                                                                                \
        }                                                                                                    \
        c->iter += 1;                                                                                        \
    }

MH_STEP(float, f32, ot_logp_f32, logf, expf)
MH_STEP(double, f64, ot_logp_f64, log, exp)

typedef struct {
    o_mh *s;
    size_t n_collect, n_discard;
    double *out;
} mh_ctx;

/* run_chain core.rs:55-73 */
static void mh_range(void *p, int lo, int hi)
{
    mh_ctx *x = (mh_ctx *)p;
    o_mh *s = x->s;
    int d = s->dim;
    size_t total = x->n_collect + x->n_discard;
    for (int ci = lo; ci < hi; ++ci) {
        mh_chain *c = &s->chains[ci];
        for (size_t i = 0; i < total; ++i) {
            if (s->is_f32)
                mh_step_f32(s, c);
            else
                mh_step_f64(s, c);
            if (i >= x->n_discard && x->out)
                memcpy(x->out + ((size_t)ci * x->n_collect + (i - x->n_discard)) * d, c->state,
                       sizeof(double) * d);
        }
    }
}

void o_mh_run(o_mh *s, size_t n_collect, size_t n_discard, int n_threads, double *out, uint64_t *accept_counts)
{
    uint64_t *before = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)s->n_chains);
    for (int i = 0; i < s->n_chains; ++i)
        before[i] = s->chains[i].n_accept;
    mh_ctx x = {s, n_collect, n_discard, out};
    par_for(mh_range, &x, s->n_chains, n_threads);
    if (accept_counts)
        for (int i = 0; i < s->n_chains; ++i)
            accept_counts[i] = s->chains[i].n_accept - before[i];
    free(before);
}

void o_mh_state(o_mh *s, double *out)
{
    for (int i = 0; i < s->n_chains; ++i)
        memcpy(out + (size_t)i * s->dim, s->chains[i].state, sizeof(double) * s->dim);
}

/* ------------------------------------------------------------------ HMC */

struct o_hmc {
    ot_target target;
    int dim, n_chains, is_f32, engine, n_leapfrog;
    double step_size;
    double *positions;  /* [C, D] holding T values */
    double *mom, *unif; /* per-step noise buffers [C, D], [C] */
    o_rng global;       /* rand_compat backend: ONE stream for everything (burn's global generator) */
    int blocked;        /* rand_compat backend, one stream per thread-block of chains (CPU baseline timing) */
    uint64_t blocked_seed;
    o_rng *chain_rng;   /* engine backend: one stream per chain */
    uint64_t *n_accept;
    uint64_t iter;
};

o_hmc *o_hmc_create(const ot_target *target, const double *init, int n_chains, double step_size, int n_leapfrog,
                    int is_f32)
{
    if (!target || !init || n_chains <= 0 || target->dim <= 0 || target->dim > HMC_MAXD || n_leapfrog < 0)
        return NULL;
    o_hmc *s = (o_hmc *)calloc(1, sizeof *s);
    s->target = *target;
    s->dim = target->dim;
    s->n_chains = n_chains;
    s->is_f32 = is_f32;
    s->n_leapfrog = n_leapfrog;
    s->step_size = is_f32 ? (double)(float)step_size : step_size;
    size_t cd = (size_t)n_chains * s->dim;
    s->positions = (double *)malloc(sizeof(double) * cd);
    s->mom = (double *)malloc(sizeof(double) * cd);
    s->unif = (double *)malloc(sizeof(double) * (size_t)n_chains);
    s->chain_rng = (o_rng *)calloc((size_t)n_chains, sizeof(o_rng));
    s->n_accept = (uint64_t *)calloc((size_t)n_chains, sizeof(uint64_t));
    for (size_t i = 0; i < cd; ++i)
        s->positions[i] = is_f32 ? (double)(float)init[i] : init[i];
    s->global.is_f32 = is_f32;
    o_rng_init_rand_compat(&s->global, 0x5eed0001ULL);
    return s;
}

void o_hmc_destroy(o_hmc *s)
{
    if (s) {
        free(s->positions);
        free(s->mom);
        free(s->unif);
        free(s->chain_rng);
        free(s->n_accept);
        free(s);
    }
}

void o_hmc_seed_global(o_hmc *s, uint64_t seed)
{
    s->engine = 0;
    s->blocked = 0;
    o_rng_init_rand_compat(&s->global, seed);
}

/* CPU-baseline mode: chains are block-partitioned over threads for the WHOLE run (what rayon's par_iter_mut over
 * chain blocks would do) and each block draws momenta / uniforms from its own rand-compatible stream, so no thread
 * waits on a global generator.  Same transition arithmetic as the global-stream mode. */
void o_hmc_seed_blocked(o_hmc *s, uint64_t seed)
{
    s->engine = 0;
    s->blocked = 1;
    s->blocked_seed = seed;
}

void o_hmc_use_engine_stream(o_hmc *s, uint64_t seed, uint64_t chain_offset)
{
    s->blocked = 0;
    s->engine = 1;
    for (int i = 0; i < s->n_chains; ++i) {
        s->chain_rng[i].is_f32 = s->is_f32;
        o_rng_init_engine_stream(&s->chain_rng[i], seed, chain_offset + (uint64_t)i);
    }
}

#define HMC_CHAIN_STEP(REAL, SUF, LOGPG, LOGP, LN)                                                           \
    /* hmc.rs:304-377 + leapfrog :397-431 for ONE row of the batched tensors */                              \
    static void hmc_chain_step_##SUF(o_hmc *s, int ci)                                                       \
    {                                                                                                        \
        int d = s->dim;                                                                                      \
        REAL eps = (REAL)s->step_size, half = (REAL)0.5;                                                     \
        REAL pos0[HMC_MAXD], pos[HMC_MAXD], mom[HMC_MAXD], g[HMC_MAXD], gs[HMC_MAXD];                        \
        for (int i = 0; i < d; ++i) {                                                                        \
            pos0[i] = (REAL)s->positions[(size_t)ci * d + i];                                                \
            pos[i] = pos0[i];                                                                                \
            mom[i] = (REAL)s->mom[(size_t)ci * d + i];                                                       \
        }                                                                                                    \
        REAL logp_current = LOGPG(&s->target, pos, g);                                                       \
        REAL es = eps * half; /* self.step_size * T::from(0.5) */                                            \
        for (int i = 0; i < d; ++i)                                                                          \
            gs[i] = g[i] * es; /* last_grad_summands */                                                      \
        REAL ke = 0;                                                                                         \
        for (int i = 0; i < d; ++i)                                                                          \
            ke = ke + mom[i] * mom[i];                                                                       \
        ke = ke * half;                                                                                      \
        REAL h_current = -logp_current + ke;                                                                 \
        for (int l = 0; l < s->n_leapfrog; ++l) {                                                            \
            for (int i = 0; i < d; ++i)                                                                      \
                mom[i] = mom[i] + gs[i];                                                                     \
            for (int i = 0; i < d; ++i)                                                                      \
                pos[i] = pos[i] + mom[i] * eps;                                                              \
            (void)LOGPG(&s->target, pos, g);                                                                 \
            for (int i = 0; i < d; ++i)                                                                      \
                gs[i] = g[i] * es;                                                                           \
            for (int i = 0; i < d; ++i)                                                                      \
                mom[i] = mom[i] + gs[i];                                                                     \
        }                                                                                                    \
        REAL logp_proposed = LOGP(&s->target, pos); /* the extra final evaluation, hmc.rs:429 */             \
        REAL kp = 0;                                                                                         \
        for (int i = 0; i < d; ++i)                                                                          \
            kp = kp + mom[i] * mom[i];                                                                       \
        kp = kp * half;                                                                                      \
        REAL h_proposed = -logp_proposed + kp;                                                               \
        REAL accept_logp = h_current - h_proposed;                                                           \
        REAL u = (REAL)s->unif[ci];                                                                          \
        REAL ln_u = LN(u); /* libm in every mode */                                                          \
        if (accept_logp >= ln_u) { /* '>=' here, '>' in MH (Q3) */                                           \
            for (int i = 0; i < d; ++i)                                                                      \
                s->positions[(size_t)ci * d + i] = (double)pos[i];                                           \
            s->n_accept[ci] += 1;                                                                            \
        }                                                                                                    \
    }

HMC_CHAIN_STEP(float, f32, ot_logp_grad_f32, ot_logp_f32, logf)
HMC_CHAIN_STEP(double, f64, ot_logp_grad_f64, ot_logp_f64, log)

static void hmc_range(void *p, int lo, int hi)
{
    o_hmc *s = (o_hmc *)p;
    for (int ci = lo; ci < hi; ++ci) {
        if (s->is_f32)
            hmc_chain_step_f32(s, ci);
        else
            hmc_chain_step_f64(s, ci);
    }
}

/* one batched step on the global stream (rand_compat backend) */
static void hmc_step_global(o_hmc *s, int n_threads)
{
    size_t cd = (size_t)s->n_chains * s->dim;
    /* Tensor::random(Normal(0,1)) over [C, D] row-major, f64 sample cast to the element type ... */
    for (size_t i = 0; i < cd; ++i) {
        double z = s->global.normal_f64(&s->global);
        s->mom[i] = s->is_f32 ? (double)(float)z : z;
    }
    /* ... and later Tensor::random(Default) over [C] (drawn here: nothing else consumes the stream) */
    for (int i = 0; i < s->n_chains; ++i) {
        double u = s->global.uniform_f64(&s->global);
        s->unif[i] = s->is_f32 ? (double)(float)u : u;
    }
    par_for(hmc_range, s, s->n_chains, n_threads);
    s->iter += 1;
}

typedef struct {
    o_hmc *s;
    size_t n_collect, n_discard;
    double *out;
} hmc_ctx;

/* engine backend: chains are independent, so each thread carries its block of chains through the whole run
 * (same results as stepping all chains in lock-step; the noise of (chain, iteration) is a pure function) */
static void hmc_engine_range(void *p, int lo, int hi)
{
    hmc_ctx *x = (hmc_ctx *)p;
    o_hmc *s = x->s;
    int d = s->dim;
    size_t total = x->n_collect + x->n_discard;
    for (int ci = lo; ci < hi; ++ci) {
        o_rng *r = &s->chain_rng[ci];
        for (size_t t = 0; t < total; ++t) {
            r->begin_iter(r, s->iter + t);
            for (int i = 0; i < d; ++i)
                s->mom[(size_t)ci * d + i] = r->normal_f64(r);
            s->unif[ci] = r->accept_uniform(r);
            if (s->is_f32)
                hmc_chain_step_f32(s, ci);
            else
                hmc_chain_step_f64(s, ci);
            if (t >= x->n_discard && x->out)
                memcpy(x->out + ((size_t)ci * x->n_collect + (t - x->n_discard)) * d,
                       s->positions + (size_t)ci * d, sizeof(double) * d);
        }
    }
}

static void hmc_blocked_range(void *p, int lo, int hi)
{
    hmc_ctx *x = (hmc_ctx *)p;
    o_hmc *s = x->s;
    int d = s->dim;
    size_t total = x->n_collect + x->n_discard;
    o_rng r;
    r.is_f32 = s->is_f32;
    o_rng_init_rand_compat(&r, s->blocked_seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(lo + 1) + s->iter);
    for (size_t t = 0; t < total; ++t) {
        for (int ci = lo; ci < hi; ++ci)
            for (int i = 0; i < d; ++i) {
                double z = r.normal_f64(&r);
                s->mom[(size_t)ci * d + i] = s->is_f32 ? (double)(float)z : z;
            }
        for (int ci = lo; ci < hi; ++ci) {
            double u = r.uniform_f64(&r);
            s->unif[ci] = s->is_f32 ? (double)(float)u : u;
        }
        for (int ci = lo; ci < hi; ++ci) {
            if (s->is_f32)
                hmc_chain_step_f32(s, ci);
            else
                hmc_chain_step_f64(s, ci);
            if (t >= x->n_discard && x->out)
                memcpy(x->out + ((size_t)ci * x->n_collect + (t - x->n_discard)) * d,
                       s->positions + (size_t)ci * d, sizeof(double) * d);
        }
    }
}

/* HMC::run hmc.rs:137-158 ; out [C, n_collect, D] (already permuted) */
void o_hmc_run(o_hmc *s, size_t n_collect, size_t n_discard, int n_threads, double *out, uint64_t *accept_counts)
{
    int d = s->dim;
    uint64_t *before = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)s->n_chains);
    memcpy(before, s->n_accept, sizeof(uint64_t) * (size_t)s->n_chains);
    if (s->engine || s->blocked) {
        hmc_ctx x = {s, n_collect, n_discard, out};
        par_for(s->engine ? hmc_engine_range : hmc_blocked_range, &x, s->n_chains, n_threads);
        s->iter += n_collect + n_discard;
    } else {
        for (size_t i = 0; i < n_discard; ++i)
            hmc_step_global(s, n_threads);
        for (size_t t = 0; t < n_collect; ++t) {
            hmc_step_global(s, n_threads);
            if (out)
                for (int ci = 0; ci < s->n_chains; ++ci)
                    memcpy(out + ((size_t)ci * n_collect + t) * d, s->positions + (size_t)ci * d,
                           sizeof(double) * d);
        }
    }
    if (accept_counts)
        for (int i = 0; i < s->n_chains; ++i)
            accept_counts[i] = s->n_accept[i] - before[i];
    free(before);
}

void o_hmc_state(o_hmc *s, double *out)
{
    memcpy(out, s->positions, sizeof(double) * (size_t)s->n_chains * s->dim);
}
