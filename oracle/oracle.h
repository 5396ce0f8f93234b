/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's hot path: mini-mcmc v0.8.3 (Rust, /root/reference), which
 * cannot be compiled in this image (no cargo/rustc, no vendored crates).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product library
 * (libmmcmc.so) never links or calls it and has no CPU fallback.
 *
 * Pinning: the RNG stack (rand_compat.c) and the NUTS restatement reproduce the reference's seeded
 * known-answer vectors nuts.rs:1123-1222 and the RNG-free ones nuts.rs:1050-1121; the statistics restate
 * stats.rs and reproduce stats.rs:704-808.  MH and HMC have no value-level vectors in the reference
 * ("parity unpinned" at value level): they are pinned statistically by the bands of
 * metropolis_hastings.rs:379-380, tests/metrohast_2d_gaussian_test.rs:46-61 and hmc.rs:764-786.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "orng.h"
#include "rand_compat.h"
#include "targets.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- core.rs:394-435  init / init_det / init_with_seed : n*d StandardNormal f64 draws, row-major ---- */
void o_init_with_seed(size_t n, size_t d, uint64_t seed, double *out);

/* ---- Metropolis-Hastings: metropolis_hastings.rs:149-193, 248-256, 303-315; core.rs:55-73, 176-186 ---- */
typedef struct o_mh o_mh;
/* MetropolisHastings::new(target, IsotropicGaussian::new(std), init) ; is_f32 selects T */
o_mh *o_mh_create(const ot_target *target, double proposal_std, const double *init, int n_chains, int is_f32);
void o_mh_destroy(o_mh *s);
/* IsotropicGaussian::set_seed (distributions.rs:388-391) applied BEFORE ::new: every chain then owns an
 * identical clone of the proposal generator (quirk Q1). */
void o_mh_proposal_seed(o_mh *s, uint64_t seed);
/* MetropolisHastings::seed (metropolis_hastings.rs:187-193): chain i accept-rng <- 1 + seed + i */
void o_mh_seed(o_mh *s, uint64_t seed);
/* run every chain on the GPU engine's counter-based stream instead (independent proposal noise per chain,
 * no extra normal per proposal, symmetric q-terms kept as the reference computes them) */
/* 0: IsotropicGaussian; 1: the asymmetric custom proposal (multiplicative log-normal random walk, mh_hmc.c) */
void o_mh_set_proposal(o_mh *s, int kind);
void o_mh_use_engine_stream(o_mh *s, uint64_t seed, uint64_t chain_offset);
/* ChainRunner::run (core.rs:176-186).  out [n_chains, n_collect, dim] (double = exact widening of T);
 * accept_counts[n_chains] (may be NULL) counts accepted proposals over all n_discard + n_collect steps. */
void o_mh_run(o_mh *s, size_t n_collect, size_t n_discard, int n_threads, double *out, uint64_t *accept_counts);
/* current states [n_chains, dim] */
void o_mh_state(o_mh *s, double *out);

/* ---- HMC: hmc.rs:87-158, 304-431 ---- */
typedef struct o_hmc o_hmc;
o_hmc *o_hmc_create(const ot_target *target, const double *init, int n_chains, double step_size, int n_leapfrog,
                    int is_f32);
void o_hmc_destroy(o_hmc *s);
/* momenta/uniforms from ONE global rand_compat stream, row-major over [chains, dim] then [chains]
 * (stands in for burn-ndarray's unseeded global generator, hmc.rs:309-313,359-363; unpinned) */
void o_hmc_seed_global(o_hmc *s, uint64_t seed);
/* one rand_compat stream per thread-block of chains (CPU-baseline timing; same transition arithmetic) */
void o_hmc_seed_blocked(o_hmc *s, uint64_t seed);
void o_hmc_use_engine_stream(o_hmc *s, uint64_t seed, uint64_t chain_offset);
void o_hmc_run(o_hmc *s, size_t n_collect, size_t n_discard, int n_threads, double *out, uint64_t *accept_counts);
void o_hmc_state(o_hmc *s, double *out);

/* ---- NUTS: nuts.rs (see nuts_impl.inc) ---- */
typedef struct o_nuts o_nuts;
o_nuts *o_nuts_create(const ot_target *target, const double *init, int n_chains, double target_accept_p,
                      int mode);
void o_nuts_destroy(o_nuts *s);
void o_nuts_set_chain_seed(o_nuts *s, int chain, uint64_t seed);
void o_nuts_set_seed(o_nuts *s, uint64_t seed);
void o_nuts_use_engine_stream(o_nuts *s, uint64_t seed, uint64_t chain_offset);
/* optional depth cap, 0 = unbounded like the reference (nuts.rs:578) */
void o_nuts_set_max_depth(o_nuts *s, int max_depth);
void o_nuts_run(o_nuts *s, size_t n_collect, size_t n_discard, int progress, int n_threads, double *out);
void o_nuts_chain_state(o_nuts *s, int chain, double out[7]);
void o_nuts_depth_hist(o_nuts *s, unsigned long long out[16], double *eps, double *eps_bar);
double o_nuts_find_reasonable_epsilon(const ot_target *target, const double *position, const double *mom,
                                      int mode);
void o_nuts_build_tree(const ot_target *target, const double *position, const double *mom, const double *grad,
                       double logu, int v, int j, double epsilon, double joint_0, uint64_t rng_seed, int mode,
                       double *vec_out, double *scal_out);

/* ---- diagnostics: stats.rs ---- */
/* split_rhat_mean_ess stats.rs:416-423 on sample[c, n, p] (f32): rhat[p] (= sqrt(W/var+), quirk Q7), ess[p] */
void o_split_rhat_mean_ess(const float *sample, size_t c, size_t n, size_t p, float *rhat, float *ess);
/* autocov_bf stats.rs:632-654 / autocov_fft stats.rs:576-620 on data[n, d] -> out[n, d] */
void o_autocov_bf(const float *data, size_t n, size_t d, float *out);
void o_autocov_fft(const float *data, size_t n, size_t d, float *out);
/* basic_stats stats.rs:310-336 : out = min, median, max, mean, std(ddof=1) */
void o_basic_stats(const float *data, size_t len, float out[5]);
/* MultiChainTracker stats.rs:189-306 : feed `steps` states [steps, chains, params]; returns rhat[params],
 * and the EMA accept estimate */
void o_multichain_tracker(const float *states, size_t steps, size_t chains, size_t params, float *rhat,
                          float *p_accept);
/* ChainTracker stats.rs:26-141 + collect_rhat stats.rs:150-178 : per-chain trackers fed [chains][steps, params]
 * with initial states init[chains, params]; returns rhat[params] and p_accept[chains] */
void o_chain_trackers_rhat(const float *init, const float *states, size_t chains, size_t steps, size_t params,
                           float *rhat, float *p_accept);

/* ess_from_chainstats stats.rs:668-671 (un-split ESS with the trackers' within / var, :155-178): per-chain trackers built on
 * init[chains, params] and fed tracked[chains, steps, params]; sample[chains, n, params] */
void o_ess_from_chainstats(const float *sample, size_t chains, size_t n, size_t params, const float *init,
                           const float *tracked, size_t steps, float *ess);

/* ---- cpu_bench.c: the timing harness of bench.py's cpu_baseline leg (one spawn per measurement, private samplers) ---- */
double o_cpu_bench(int kind, const ot_target *target, const double *init, int chains_per_thread, int n_threads,
                   double scale, int n_leapfrog, int is_f32, size_t n_collect, size_t n_discard, int reps,
                   uint64_t seed, int nuts_mode, int nuts_max_depth, double *thread_seconds, double *work_out);

#ifdef __cplusplus
}
#endif
#endif
