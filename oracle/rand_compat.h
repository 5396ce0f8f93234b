/*
 * oracle/rand_compat.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the third-party RNG stack the reference's hot path depends on:
 *   rand 0.9.4   SmallRng (= xoshiro256++ on 64-bit), seed_from_u64 (SplitMix64), StandardUniform
 *   rand_distr 0.5.1  StandardNormal / Exp1 (256-layer ziggurat), Normal{mu,sigma}, Open01
 * Neither crate is under /root/reference (Cargo.lock pins the versions); the published algorithms
 * are restated here and pinned by the reference's own seeded known-answer tests
 * (nuts.rs:1123-1222), which only reproduce if every piece of this stream is right.
 *
 * Reference call sites: core.rs:398,417,429  metropolis_hastings.rs:190,253,310
 *                       distributions.rs:355,365-368,389  nuts.rs:350,414,444,535-538,554-557,563,579,660,910
 */
#ifndef ORACLE_RAND_COMPAT_H
#define ORACLE_RAND_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t s[4];
} rc_rng; /* rand::rngs::SmallRng on 64-bit targets: xoshiro256++ */

/* SmallRng::seed_from_u64 : four SplitMix64 outputs fill the state. */
void rc_seed_from_u64(rc_rng *r, uint64_t seed);
uint64_t rc_next_u64(rc_rng *r);
/* xoshiro256++ next_u32 = upper half of next_u64 */
uint32_t rc_next_u32(rc_rng *r);

/* StandardUniform */
double rc_uniform_f64(rc_rng *r); /* (next_u64 >> 11) * 2^-53, in [0,1) */
float rc_uniform_f32(rc_rng *r);  /* (next_u32 >> 8)  * 2^-24, in [0,1) */
double rc_open01_f64(rc_rng *r);  /* Open01: (0,1) */

/* rand_distr::StandardNormal / Exp1 for f64; the f32 impls sample f64 and cast. */
double rc_standard_normal(rc_rng *r);
double rc_exp1(rc_rng *r);

/* table access for tests (257 entries each) */
const double *rc_zig_norm_x(void);
const double *rc_zig_norm_f(void);
const double *rc_zig_exp_x(void);
const double *rc_zig_exp_f(void);

#ifdef __cplusplus
}
#endif
#endif
