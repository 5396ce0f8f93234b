/*
 * oracle/orng.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The oracle samplers draw their randomness through this small vtable so the same restated sampler
 * code can run on
 *   (a) the reference's own stream (rand_compat: xoshiro256++ + ziggurat), which is what the
 *       reference's seeded known-answer tests pin, and
 *   (b) the GPU engine's counter-based stream (an independent plain-C statement of the Philox4x32-10
 *       draw schedule in DESIGN.md), so GPU results can be compared against the reference-ordered
 *       arithmetic on identical noise.
 */
#ifndef ORACLE_ORNG_H
#define ORACLE_ORNG_H

#include <stddef.h>

#include "rand_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o_rng o_rng;
struct o_rng {
    double (*normal_f64)(o_rng *);     /* rand_distr::StandardNormal -> f64 (f32 callers cast) */
    double (*exp1_f64)(o_rng *);       /* rand_distr::Exp1 -> f64 */
    double (*uniform_f64)(o_rng *);    /* StandardUniform f64 */
    float (*uniform_f32)(o_rng *);     /* StandardUniform f32 */
    double (*accept_uniform)(o_rng *); /* the MH / HMC accept draw `rng.random::<T>()`, T per is_f32 */
    void (*begin_iter)(o_rng *, uint64_t iteration); /* counter-based streams re-key here; no-op for (a) */
    int is_f32;                                       /* element type the sampler runs in */
    rc_rng rc;                                        /* state of backend (a) */
    uint64_t seed, chain, iteration;                  /* state of backend (b) */
    uint32_t draw, n_normal;                          /* running indices within the iteration, (b) */
    int mh_paired;                                    /* (b): the MH sampler's f32 stream at dim <= 2 (o_rng_engine_mh_paired) */
};

/* (a): SmallRng::seed_from_u64(seed).  Keeps r->is_f32. */
void o_rng_init_rand_compat(o_rng *r, uint64_t seed);
/* (b): engine stream of global chain id `chain` under `seed`.  Keeps r->is_f32. */
void o_rng_init_engine_stream(o_rng *r, uint64_t seed, uint64_t chain);
/* (b) for the Metropolis-Hastings sampler in f32 at dim <= 2: the engine's paired stream (csrc/mm_rng.h: one Philox block
 * per TWO transitions -- iteration t takes words 2 (t & 1), 2 (t & 1) + 1 of block (chain, t >> 1, 0x20000000) for its normals
 * and the high 16 bits of its accept uniform, the low 8 bits from word 0 of block (chain, t >> 1, 0x20000001)).  Call after
 * o_rng_init_engine_stream; no effect for f64. */
void o_rng_engine_mh_paired(o_rng *r, int dim);
float o_engine_mhp_normal_f32(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i);
float o_engine_mhp_accept_f32(uint64_t seed, uint64_t chain, uint32_t iteration);

/* raw pieces of (b), exported for the known-answer tests and for noise injection */
void o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void o_engine_block(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t block, uint32_t out[4]);
float o_engine_normal_f32(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i);
float o_engine_icdf24(uint32_t w); /* the f32 normal of one Philox word */
void o_engine_icdf24_words(const uint32_t *w, size_t n, float *out);
/* the same normal without the product's table: f64 inverse normal CDF (Halley on libm's erfc) rounded to f32 */
double o_ndtri(double p);
float o_engine_icdf24_independent(uint32_t w);
void o_engine_icdf24_independent_words(const uint32_t *w, size_t n, float *out);
double o_engine_normal_f64(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i);
double o_engine_normal_f64_libm(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i);
float o_engine_accept_f32(uint64_t seed, uint64_t chain, uint32_t iteration);
double o_engine_aux_u53(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
