/*
 * oracle/cpu_bench.c -- TEST INFRASTRUCTURE ONLY: the timing harness behind bench.py's `cpu_baseline` leg.
 *
 * What the reference's rayon path amounts to for equal-cost chains (core.rs:176-186: `par_iter_mut` over the chains,
 * every chain carried through its whole run by one worker): ONE spawn per measurement, every thread owns a private
 * sampler over its block of chains (state, noise buffers and the sample it writes are allocated and first touched by
 * the thread itself), all threads leave a barrier together, each runs `reps` x run(n_collect, n_discard) and stamps
 * its own start and end.  wall = last end - first start.  Nothing is shared between threads while the clock runs.
 * The transition arithmetic is the oracle's own (mh_hmc.c / nuts.c: the reference's order, quirks kept).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    int kind; /* 0 HMC, 1 MH (reference-faithful: Q1, Q2 kept), 2 NUTS */
    const ot_target *target;
    const double *init; /* this thread's [chains, dim] */
    int chains, is_f32, n_leapfrog, reps, nuts_mode, nuts_max_depth;
    double scale; /* HMC step size / MH proposal std / NUTS target accept */
    size_t n_collect, n_discard;
    uint64_t seed;
    pthread_barrier_t *gate;
    double t0, t1;
    double work; /* NUTS: leapfrog steps this thread took inside the timed region */
} bench_job;

static void *bench_thread(void *p)
{
    bench_job *j = (bench_job *)p;
    int d = j->target->dim;
    size_t out_n = (size_t)j->chains * j->n_collect * (size_t)d;
    double *out = (double *)malloc(sizeof(double) * (out_n ? out_n : 1));
    memset(out, 0, sizeof(double) * (out_n ? out_n : 1)); /* first touch by the owner */
    if (j->kind == 0) {
        o_hmc *s = o_hmc_create(j->target, j->init, j->chains, j->scale, j->n_leapfrog, j->is_f32);
        o_hmc_seed_blocked(s, j->seed);
        o_hmc_run(s, 4, 2, 1, out, NULL); /* code and tables warm */
        pthread_barrier_wait(j->gate);
        j->t0 = now_s();
        for (int r = 0; r < j->reps; ++r)
            o_hmc_run(s, j->n_collect, j->n_discard, 1, out, NULL);
        j->t1 = now_s();
        o_hmc_destroy(s);
    } else if (j->kind == 1) {
        o_mh *s = o_mh_create(j->target, j->scale, j->init, j->chains, j->is_f32);
        o_mh_proposal_seed(s, j->seed);
        o_mh_seed(s, j->seed);
        o_mh_run(s, 4, 2, 1, out, NULL);
        pthread_barrier_wait(j->gate);
        j->t0 = now_s();
        for (int r = 0; r < j->reps; ++r)
            o_mh_run(s, j->n_collect, j->n_discard, 1, out, NULL);
        j->t1 = now_s();
        o_mh_destroy(s);
    } else {
        o_nuts *s = o_nuts_create(j->target, j->init, j->chains, j->scale, j->nuts_mode);
        o_nuts_set_seed(s, j->seed);
        o_nuts_set_max_depth(s, j->nuts_max_depth);
        pthread_barrier_wait(j->gate);
        j->t0 = now_s();
        /* one run: NUTS adapts its step size over the warm-up, a second run would continue the adapted chains */
        o_nuts_run(s, j->n_collect, j->n_discard, 1, 1, out);
        j->t1 = now_s();
        double st[7];
        j->work = 0;
        for (int c = 0; c < j->chains; ++c) {
            o_nuts_chain_state(s, c, st);
            j->work += st[6];
        }
        o_nuts_destroy(s);
    }
    free(out);
    return NULL;
}

/* returns the wall time of the timed region in seconds (< 0: failed); thread_seconds[n_threads] (may be NULL) gets
 * every thread's own t1 - t0; work_out (may be NULL) the NUTS leapfrog count summed over threads */
double o_cpu_bench(int kind, const ot_target *target, const double *init, int chains_per_thread, int n_threads,
                   double scale, int n_leapfrog, int is_f32, size_t n_collect, size_t n_discard, int reps,
                   uint64_t seed, int nuts_mode, int nuts_max_depth, double *thread_seconds, double *work_out)
{
    if (!target || !init || chains_per_thread <= 0 || n_threads <= 0 || reps <= 0 || kind < 0 || kind > 2)
        return -1.0;
    pthread_barrier_t gate;
    if (pthread_barrier_init(&gate, NULL, (unsigned)n_threads))
        return -1.0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    bench_job *jobs = (bench_job *)calloc((size_t)n_threads, sizeof(bench_job));
    int d = target->dim;
    for (int t = 0; t < n_threads; ++t) {
        bench_job *j = &jobs[t];
        j->kind = kind;
        j->target = target;
        j->init = init + (size_t)t * chains_per_thread * d;
        j->chains = chains_per_thread;
        j->is_f32 = is_f32;
        j->n_leapfrog = n_leapfrog;
        j->reps = reps;
        j->scale = scale;
        j->n_collect = n_collect;
        j->n_discard = n_discard;
        j->seed = seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(t + 1);
        j->nuts_mode = nuts_mode;
        j->nuts_max_depth = nuts_max_depth;
        j->gate = &gate;
    }
    int started = 0;
    for (; started < n_threads; ++started)
        if (pthread_create(&th[started], NULL, bench_thread, &jobs[started]))
            break;
    if (started < n_threads) { /* cannot happen quietly: the others would wait at the gate for ever */
        abort();
    }
    for (int t = 0; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    double first = jobs[0].t0, last = jobs[0].t1, work = 0;
    for (int t = 0; t < n_threads; ++t) {
        if (jobs[t].t0 < first)
            first = jobs[t].t0;
        if (jobs[t].t1 > last)
            last = jobs[t].t1;
        if (thread_seconds)
            thread_seconds[t] = jobs[t].t1 - jobs[t].t0;
        work += jobs[t].work;
    }
    if (work_out)
        *work_out = work;
    pthread_barrier_destroy(&gate);
    free(th);
    free(jobs);
    return last - first;
}
