/*
 * oracle/nuts.c -- TEST INFRASTRUCTURE ONLY.
 * Public C entry points of the NUTS oracle (restating nuts.rs; see nuts_impl.inc for file:line map).
 *
 * modes:  0 = tensors f32, scalars f64   (reference `NUTS<f64, Autodiff<NdArray>, _>`)
 *         1 = tensors f32, scalars f32   (reference `NUTS<f32, Autodiff<NdArray>, _>`)
 *         2 = tensors f64, scalars f64   (no reference backend has f64 elements by default; used for
 *                                         BASELINE.json config 5 "f64")
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- mode 0 ---- */
#define TT float
#define ST double
#define SUF m0
#define TT_LOGP_GRAD ot_logp_grad_f32
#define ST_EXP exp
#define ST_LOG log
#define ST_SQRT sqrt
#define ST_POW pow
#define ST_FMIN fmin
#define ST_FABS fabs
#define ST_EPS 0x1.0p-52
#define ST_UNIFORM(r) ((r)->uniform_f64(r))
#include "nuts_impl.inc"
#undef TT
#undef ST
#undef SUF
#undef TT_LOGP_GRAD
#undef ST_EXP
#undef ST_LOG
#undef ST_SQRT
#undef ST_POW
#undef ST_FMIN
#undef ST_FABS
#undef ST_EPS
#undef ST_UNIFORM

/* ---- mode 1 ---- */
#define TT float
#define ST float
#define SUF m1
#define TT_LOGP_GRAD ot_logp_grad_f32
#define ST_EXP expf
#define ST_LOG logf
#define ST_SQRT sqrtf
#define ST_POW powf
#define ST_FMIN fminf
#define ST_FABS fabsf
#define ST_EPS 0x1.0p-23f
#define ST_UNIFORM(r) ((r)->uniform_f32(r))
#include "nuts_impl.inc"
#undef TT
#undef ST
#undef SUF
#undef TT_LOGP_GRAD
#undef ST_EXP
#undef ST_LOG
#undef ST_SQRT
#undef ST_POW
#undef ST_FMIN
#undef ST_FABS
#undef ST_EPS
#undef ST_UNIFORM

/* ---- mode 2 ---- */
#define TT double
#define ST double
#define SUF m2
#define TT_LOGP_GRAD ot_logp_grad_f64
#define ST_EXP exp
#define ST_LOG log
#define ST_SQRT sqrt
#define ST_POW pow
#define ST_FMIN fmin
#define ST_FABS fabs
#define ST_EPS 0x1.0p-52
#define ST_UNIFORM(r) ((r)->uniform_f64(r))
#include "nuts_impl.inc"
#undef TT
#undef ST
#undef SUF
#undef TT_LOGP_GRAD
#undef ST_EXP
#undef ST_LOG
#undef ST_SQRT
#undef ST_POW
#undef ST_FMIN
#undef ST_FABS
#undef ST_EPS
#undef ST_UNIFORM

struct o_nuts {
    int mode, n_chains, dim;
    void *chains; /* array of nuts_chain_m{0,1,2} */
};

#define DISPATCH(s, expr0, expr1, expr2)                                                                     \
    do {                                                                                                     \
        if ((s)->mode == 0) {                                                                                \
            expr0;                                                                                           \
        } else if ((s)->mode == 1) {                                                                         \
            expr1;                                                                                           \
        } else {                                                                                             \
            expr2;                                                                                           \
        }                                                                                                    \
    } while (0)

/* NUTS::new nuts.rs:123-129 */
o_nuts *o_nuts_create(const ot_target *target, const double *init, int n_chains, double target_accept_p,
                      int mode)
{
    if (!target || !init || n_chains <= 0 || target->dim <= 0 || target->dim > 128 || mode < 0 || mode > 2)
        return NULL;
    o_nuts *s = (o_nuts *)calloc(1, sizeof *s);
    s->mode = mode;
    s->n_chains = n_chains;
    s->dim = target->dim;
    size_t sz = mode == 0 ? sizeof(nuts_chain_m0) : mode == 1 ? sizeof(nuts_chain_m1) : sizeof(nuts_chain_m2);
    s->chains = calloc((size_t)n_chains, sz);
    for (int i = 0; i < n_chains; ++i) {
        const double *p = init + (size_t)i * s->dim;
        DISPATCH(s, nuts_chain_init_m0(&((nuts_chain_m0 *)s->chains)[i], target, p, target_accept_p),
                 nuts_chain_init_m1(&((nuts_chain_m1 *)s->chains)[i], target, p, target_accept_p),
                 nuts_chain_init_m2(&((nuts_chain_m2 *)s->chains)[i], target, p, target_accept_p));
        /* SmallRng::from_os_rng() in the reference; the oracle defaults to a fixed, distinct seed */
        o_nuts_set_chain_seed(s, i, 0x5eed0000ULL + (uint64_t)i);
    }
    return s;
}

void o_nuts_destroy(o_nuts *s)
{
    if (s) {
        free(s->chains);
        free(s);
    }
}

static o_rng *chain_rng(o_nuts *s, int i)
{
    o_rng *r;
    if (s->mode == 0)
        r = &((nuts_chain_m0 *)s->chains)[i].rng;
    else if (s->mode == 1)
        r = &((nuts_chain_m1 *)s->chains)[i].rng;
    else
        r = &((nuts_chain_m2 *)s->chains)[i].rng;
    r->is_f32 = (s->mode != 2); /* element type of the momentum draws */
    return r;
}

/* NUTSChain::set_seed nuts.rs:443-446 */
void o_nuts_set_chain_seed(o_nuts *s, int chain, uint64_t seed)
{
    o_rng_init_rand_compat(chain_rng(s, chain), seed);
}

/* NUTS::set_seed nuts.rs:347-353 : chain i gets seed + i + 1 */
void o_nuts_set_seed(o_nuts *s, uint64_t seed)
{
    for (int i = 0; i < s->n_chains; ++i)
        o_nuts_set_chain_seed(s, i, seed + (uint64_t)i + 1);
}

/* depth cap (0 = none, the reference's behaviour) */
void o_nuts_set_max_depth(o_nuts *s, int max_depth)
{
    for (int i = 0; i < s->n_chains; ++i) {
        if (s->mode == 0)
            ((nuts_chain_m0 *)s->chains)[i].max_depth = max_depth;
        else if (s->mode == 1)
            ((nuts_chain_m1 *)s->chains)[i].max_depth = max_depth;
        else
            ((nuts_chain_m2 *)s->chains)[i].max_depth = max_depth;
    }
}

/* switch every chain to the GPU engine's counter-based stream (orng.c) */
void o_nuts_use_engine_stream(o_nuts *s, uint64_t seed, uint64_t chain_offset)
{
    for (int i = 0; i < s->n_chains; ++i)
        o_rng_init_engine_stream(chain_rng(s, i), seed, chain_offset + (uint64_t)i);
}

typedef struct {
    o_nuts *s;
    size_t n_collect, n_discard;
    int progress, lo, hi;
    double *out;
} nuts_job;

static void nuts_run_range(nuts_job *jb)
{
    o_nuts *s = jb->s;
    size_t row = jb->n_collect * (size_t)s->dim;
    void *tmp = malloc((row ? row : 1) * sizeof(double));
    for (int i = jb->lo; i < jb->hi; ++i) {
        double *o = jb->out + (size_t)i * row;
        if (s->mode == 2) {
            nuts_chain_run_m2(&((nuts_chain_m2 *)s->chains)[i], jb->n_collect, jb->n_discard, jb->progress, o);
        } else {
            float *t = (float *)tmp;
            if (s->mode == 0)
                nuts_chain_run_m0(&((nuts_chain_m0 *)s->chains)[i], jb->n_collect, jb->n_discard,
                                  jb->progress, t);
            else
                nuts_chain_run_m1(&((nuts_chain_m1 *)s->chains)[i], jb->n_collect, jb->n_discard,
                                  jb->progress, t);
            for (size_t k = 0; k < row; ++k)
                o[k] = (double)t[k];
        }
    }
    free(tmp);
}

static void *nuts_thread(void *p)
{
    nuts_run_range((nuts_job *)p);
    return NULL;
}

/* NUTS::run nuts.rs:163-170 (rayon par_iter_mut over chains -> static block partition over threads).
 * out: [n_chains, n_collect, dim] as double (exact widening of the f32 tensors in modes 0/1). */
void o_nuts_run(o_nuts *s, size_t n_collect, size_t n_discard, int progress, int n_threads, double *out)
{
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > s->n_chains)
        n_threads = s->n_chains;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    nuts_job *jobs = (nuts_job *)malloc(sizeof(nuts_job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].s = s;
        jobs[t].n_collect = n_collect;
        jobs[t].n_discard = n_discard;
        jobs[t].progress = progress;
        jobs[t].lo = (int)((long)s->n_chains * t / n_threads);
        jobs[t].hi = (int)((long)s->n_chains * (t + 1) / n_threads);
        jobs[t].out = out;
        if (n_threads == 1)
            nuts_run_range(&jobs[t]);
        else
            pthread_create(&th[t], NULL, nuts_thread, &jobs[t]);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; ++t)
            pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* per-chain adaptation state, for traces: out = epsilon, epsilon_bar, h_bar, mu, m, last_depth, n_leapfrog_total */
void o_nuts_chain_state(o_nuts *s, int chain, double out[7])
{
#define FILL(c)                                                                                              \
    out[0] = (double)(c)->epsilon;                                                                           \
    out[1] = (double)(c)->epsilon_bar;                                                                       \
    out[2] = (double)(c)->h_bar;                                                                             \
    out[3] = (double)(c)->mu;                                                                                \
    out[4] = (double)(c)->m;                                                                                 \
    out[5] = (double)(c)->last_depth;                                                                        \
    out[6] = (double)(c)->n_leapfrog_total
    DISPATCH(s, FILL(&((nuts_chain_m0 *)s->chains)[chain]), FILL(&((nuts_chain_m1 *)s->chains)[chain]),
             FILL(&((nuts_chain_m2 *)s->chains)[chain]));
#undef FILL
}

/* transitions of all chains by tree depth (diagnostic, not in the reference); also every chain's epsilon / epsilon_bar */
void o_nuts_depth_hist(o_nuts *s, unsigned long long out[16], double *eps, double *eps_bar)
{
    memset(out, 0, 16 * sizeof(unsigned long long));
#define ACC(TYPE)                                                                                            \
    for (int i = 0; i < s->n_chains; ++i) {                                                                  \
        const TYPE *c = &((const TYPE *)s->chains)[i];                                                       \
        for (int k = 0; k < 16; ++k)                                                                         \
            out[k] += c->depth_hist[k];                                                                      \
        if (eps)                                                                                             \
            eps[i] = (double)c->epsilon;                                                                     \
        if (eps_bar)                                                                                         \
            eps_bar[i] = (double)c->epsilon_bar;                                                             \
    }
    DISPATCH(s, ACC(nuts_chain_m0), ACC(nuts_chain_m1), ACC(nuts_chain_m2));
#undef ACC
}

/* nuts.rs:695-761, for the known-answer test nuts.rs:1050-1055 */
double o_nuts_find_reasonable_epsilon(const ot_target *target, const double *position, const double *mom,
                                      int mode)
{
    int d = target->dim;
    if (mode == 2) {
        nuts_chain_m2 c;
        nuts_chain_init_m2(&c, target, position, 0.8);
        return find_reasonable_epsilon_m2(&c, position, mom);
    }
    float p[128], m[128];
    for (int i = 0; i < d; ++i) {
        p[i] = (float)position[i];
        m[i] = (float)mom[i];
    }
    if (mode == 0) {
        nuts_chain_m0 c;
        nuts_chain_init_m0(&c, target, position, 0.8);
        return find_reasonable_epsilon_m0(&c, p, m);
    }
    nuts_chain_m1 c;
    nuts_chain_init_m1(&c, target, position, 0.8);
    return (double)find_reasonable_epsilon_m1(&c, p, m);
}

/* nuts.rs:764-946, for the known-answer test nuts.rs:1057-1121.
 * vec_out: 8 vectors of length dim in the order pos_minus, mom_minus, grad_minus, pos_plus, mom_plus,
 * grad_plus, pos_prime, grad_prime; scal_out: logp_prime, n_prime, s_prime, alpha_prime, n_alpha_prime. */
void o_nuts_build_tree(const ot_target *target, const double *position, const double *mom, const double *grad,
                       double logu, int v, int j, double epsilon, double joint_0, uint64_t rng_seed, int mode,
                       double *vec_out, double *scal_out)
{
    int d = target->dim;
#define RUN(M, TTYPE, STYPE)                                                                                 \
    do {                                                                                                     \
        nuts_chain_##M c;                                                                                    \
        nuts_chain_init_##M(&c, target, position, 0.8);                                                      \
        o_rng_init_rand_compat(&c.rng, rng_seed);                                                            \
        TTYPE p[128], mo[128], g[128];                                                                         \
        for (int i = 0; i < d; ++i) {                                                                        \
            p[i] = (TTYPE)position[i];                                                                       \
            mo[i] = (TTYPE)mom[i];                                                                           \
            g[i] = (TTYPE)grad[i];                                                                           \
        }                                                                                                    \
        tree_##M t;                                                                                          \
        build_tree_##M(&c, p, mo, g, (STYPE)logu, v, (size_t)j, (STYPE)epsilon, (STYPE)joint_0, &t);         \
        const TTYPE *vs[8] = {t.pos_minus, t.mom_minus, t.grad_minus, t.pos_plus,                            \
                              t.mom_plus,  t.grad_plus, t.pos_prime,  t.grad_prime};                         \
        for (int k = 0; k < 8; ++k)                                                                          \
            for (int i = 0; i < d; ++i)                                                                      \
                vec_out[k * d + i] = (double)vs[k][i];                                                       \
        scal_out[0] = (double)t.logp_prime;                                                                  \
        scal_out[1] = (double)t.n_prime;                                                                     \
        scal_out[2] = (double)t.s_prime;                                                                     \
        scal_out[3] = (double)t.alpha_prime;                                                                 \
        scal_out[4] = (double)t.n_alpha_prime;                                                               \
    } while (0)
    if (mode == 0)
        RUN(m0, float, double);
    else if (mode == 1)
        RUN(m1, float, float);
    else
        RUN(m2, double, double);
#undef RUN
}
