/*
 * oracle/orng.c -- TEST INFRASTRUCTURE ONLY. Random-stream backends for the oracle samplers (see orng.h).
 *
 * Backend (b), the engine stream, is an INDEPENDENT plain-C statement of the Philox4x32-10 generator
 * (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; pinned in
 * tests/test_oracle_rng.py by the Random123 known-answer vectors) and of the draw schedule written
 * down in DESIGN.md / mini_mcmc_amd/csrc/mm_rng.h.  The NOISE of the engine stream shares the elementary functions
 * (mm_log, mm_sincos2pi) and the f32 normal's coefficient table (mm_icdf_table.h, data) with the product, because those
 * DEFINE the engine's noise: tests that compare trajectories on identical noise need the very same numbers.  Beside them
 * stand statements that share nothing -- o_ndtri / o_engine_icdf24_independent (f64 inverse normal CDF by Halley's
 * iteration on libm's erfc) and o_engine_normal_f64_libm (libm log / sin / cos) -- and tests/test_engine_stream.py
 * compares the product's numbers with those over the whole 24-bit lattice / a million draws.  The samplers' own
 * arithmetic (accept logarithms, targets) is libm's everywhere (mh_hmc.c, nuts.c, targets.c).
 */
#include "orng.h"

#include "../mini_mcmc_amd/csrc/mm_math.h"
#include "../mini_mcmc_amd/csrc/mm_icdf_table.h"

#include <math.h>

/* ---------------- backend (a): the reference's stream ---------------- */

static double rcb_normal(o_rng *r) { return rc_standard_normal(&r->rc); }
static double rcb_exp1(o_rng *r) { return rc_exp1(&r->rc); }
static double rcb_uniform_f64(o_rng *r) { return rc_uniform_f64(&r->rc); }
static float rcb_uniform_f32(o_rng *r) { return rc_uniform_f32(&r->rc); }
static void rcb_begin_iter(o_rng *r, uint64_t it)
{
    (void)r;
    (void)it;
}
static double rcb_accept_uniform(o_rng *r)
{
    return r->is_f32 ? (double)rc_uniform_f32(&r->rc) : rc_uniform_f64(&r->rc);
}

void o_rng_init_rand_compat(o_rng *r, uint64_t seed)
{
    int is_f32 = r->is_f32;
    memset(r, 0, sizeof *r);
    r->is_f32 = is_f32;
    r->normal_f64 = rcb_normal;
    r->exp1_f64 = rcb_exp1;
    r->uniform_f64 = rcb_uniform_f64;
    r->uniform_f32 = rcb_uniform_f32;
    r->accept_uniform = rcb_accept_uniform;
    r->begin_iter = rcb_begin_iter;
    rc_seed_from_u64(&r->rc, seed);
}

/* ---------------- backend (b): the engine's counter-based stream ---------------- */

static void philox_round(uint32_t c[4], uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n[4];
    n[0] = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    n[1] = (uint32_t)p1;
    n[2] = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    n[3] = (uint32_t)p0;
    c[0] = n[0];
    c[1] = n[1];
    c[2] = n[2];
    c[3] = n[3];
}

void o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}

void o_engine_block(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t block, uint32_t out[4])
{
    uint32_t ctr[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), iteration, block};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    o_philox4x32_10(ctr, key, out);
}

static float u24(uint32_t w) { return (float)((w >> 8) + 1u) * 0x1.0p-24f; }
static double u53(uint32_t hi, uint32_t lo)
{
    uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)(m + 1ull) * 0x1.0p-53;
}

/* normal #i of (chain, iteration) in the f32 schedule: word i % 4 of block i / 4, inverted through the piecewise
 * cubic of -Phi^-1 (mm_rng.h "icdf24"; the coefficient table is shared DATA, the evaluation below is stated on its
 * own: frexpf for the binade, arithmetic for segment and offset, where the product manipulates the float's bits) */
static const float icdf_tab[MM_ICDF_ROWS * 4] = MM_ICDF_COEFFS;

float o_engine_icdf24(uint32_t w)
{
    uint32_t n = (w >> 8) | 1u; /* odd 24-bit integer; the normal is sign * -Phi^-1(n 2^-25) */
    int e;
    float fr = frexpf((float)n, &e);  /* n = fr 2^e, fr in [1/2, 1) */
    float m = 2.0f * fr - 1.0f;       /* n = (1 + m) 2^(e-1), m in [0, 1), exact */
    int j = (int)(m * 16.0f);         /* segment of the binade */
    float t = (m - (float)j * 0.0625f); /* offset inside the segment, in [0, 1/16), exact */
    const float *c = icdf_tab + 4 * (((((e - 1) + 127) & 31) << 4) | j);
    float mag = fabsf(fmaf(fmaf(fmaf(c[3], t, c[2]), t, c[1]), t, c[0]));
    return ((w >> 8) & 1u) ? -mag : mag;
}

/* The same normal stated WITHOUT the product's table: -Phi^-1(n 2^-25) in f64 -- Halley's iteration on libm's erfc from the
 * Abramowitz-Stegun 26.2.22 starting point -- rounded to f32.  The product's table-driven value is an approximation of
 * this number; tests/test_engine_stream.py compares the two over the whole 2^24 lattice (data against data). */
double o_ndtri(double p)
{
    if (!(p > 0.0 && p < 1.0))
        return p == 0.0 ? -INFINITY : (p == 1.0 ? INFINITY : NAN);
    const int upper = p > 0.5;
    const double q = upper ? 1.0 - p : p; /* lower-tail probability, (0, 1/2] */
    const double t = sqrt(-2.0 * log(q));
    double x = -(t - (2.30753 + 0.27061 * t) / (1.0 + t * (0.99229 + 0.04481 * t))); /* |error| < 3e-3 */
    for (int it = 0; it < 8; ++it) {
        const double cdf = 0.5 * erfc(-x * 0.70710678118654752440);      /* Phi(x), accurate in the lower tail */
        const double pdf = 0.39894228040143267794 * exp(-0.5 * x * x);
        const double f = cdf - q;
        const double step = f / (pdf * (1.0 + 0.5 * x * f / pdf));        /* Halley: f'' / f' = -x */
        x -= step;
        if (fabs(step) <= 1e-16 * fmax(1.0, fabs(x)))
            break;
    }
    return upper ? -x : x;
}

float o_engine_icdf24_independent(uint32_t w)
{
    const uint32_t n = (w >> 8) | 1u;
    const float mag = (float)(-o_ndtri((double)n * 0x1.0p-25));
    return ((w >> 8) & 1u) ? -mag : mag;
}

void o_engine_icdf24_independent_words(const uint32_t *w, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = o_engine_icdf24_independent(w[i]);
}

void o_engine_icdf24_words(const uint32_t *w, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = o_engine_icdf24(w[i]);
}

float o_engine_normal_f32(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration, i / 4, w);
    return o_engine_icdf24(w[i % 4]);
}

double o_engine_normal_f64(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration, i / 2, w);
    double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    double r = sqrt(-2.0 * mm_log(u1));
    double s, c;
    mm_sincos2pi(u2, &s, &c);
    return (i % 2) ? r * s : r * c;
}

/* the same Box-Muller pair from libm's log / sincos instead of the engine's (mm_math.h): what the engine's elementary
 * functions are compared with, as data, in tests/test_engine_stream.py */
double o_engine_normal_f64_libm(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration, i / 2, w);
    double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    double r = sqrt(-2.0 * log(u1));
    double a = 6.28318530717958647692 * u2;
    return (i % 2) ? r * sin(a) : r * cos(a);
}

/* MH / HMC accept uniform of (chain, iteration) */
float o_engine_accept_f32(uint64_t seed, uint64_t chain, uint32_t iteration)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration, 0, w);
    uint32_t s = (w[0] & 255u) | ((w[1] & 255u) << 8) | ((w[2] & 255u) << 16);
    return (float)(s + 1u) * 0x1.0p-24f;
}

/* the MH sampler's paired f32 stream (csrc/mm_rng.h, round 5) */
float o_engine_mhp_normal_f32(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t i)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration >> 1, 0x20000000u, w);
    return o_engine_icdf24(w[2u * (iteration & 1u) + i]);
}
float o_engine_mhp_accept_f32(uint64_t seed, uint64_t chain, uint32_t iteration)
{
    uint32_t w[4], a[4];
    o_engine_block(seed, chain, iteration >> 1, 0x20000000u, w);
    o_engine_block(seed, chain, iteration >> 1, 0x20000001u, a);
    const uint32_t h = iteration & 1u;
    const uint32_t s16 = (w[2u * h] & 255u) | ((w[2u * h + 1u] & 255u) << 8);
    const uint32_t s8 = (a[0] >> (8u * h)) & 255u;
    return (float)(s16 * 256u + s8 + 1u) * 0x1.0p-24f;
}

double o_engine_aux_u53(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t k)
{
    uint32_t w[4];
    o_engine_block(seed, chain, iteration, 0x40000000u + (k >> 1), w);
    return (k & 1u) ? u53(w[2], w[3]) : u53(w[0], w[1]);
}

static double eng_normal(o_rng *r)
{
    uint32_t i = r->n_normal++;
    if (r->mh_paired)
        return (double)o_engine_mhp_normal_f32(r->seed, r->chain, (uint32_t)r->iteration, i);
    return r->is_f32 ? (double)o_engine_normal_f32(r->seed, r->chain, (uint32_t)r->iteration, i)
                     : o_engine_normal_f64(r->seed, r->chain, (uint32_t)r->iteration, i);
}
static double eng_aux(o_rng *r) { return o_engine_aux_u53(r->seed, r->chain, (uint32_t)r->iteration, r->draw++); }
static double eng_exp1(o_rng *r)
{
    double u = eng_aux(r);
    return -mm_log(u);
}
static double eng_uniform_f64(o_rng *r) { return eng_aux(r); }
static float eng_uniform_f32(o_rng *r) { return (float)eng_aux(r); }
static double eng_accept_uniform(o_rng *r)
{
    if (r->mh_paired)
        return (double)o_engine_mhp_accept_f32(r->seed, r->chain, (uint32_t)r->iteration);
    return r->is_f32 ? (double)o_engine_accept_f32(r->seed, r->chain, (uint32_t)r->iteration)
                     : o_engine_aux_u53(r->seed, r->chain, (uint32_t)r->iteration, 0);
}
static void eng_begin_iter(o_rng *r, uint64_t it)
{
    r->iteration = it;
    r->draw = 0;
    r->n_normal = 0;
}

void o_rng_init_engine_stream(o_rng *r, uint64_t seed, uint64_t chain)
{
    int is_f32 = r->is_f32;
    memset(r, 0, sizeof *r);
    r->is_f32 = is_f32;
    r->normal_f64 = eng_normal;
    r->exp1_f64 = eng_exp1;
    r->uniform_f64 = eng_uniform_f64;
    r->uniform_f32 = eng_uniform_f32;
    r->accept_uniform = eng_accept_uniform;
    r->begin_iter = eng_begin_iter;
    r->seed = seed;
    r->chain = chain;
}

void o_rng_engine_mh_paired(o_rng *r, int dim) { r->mh_paired = r->is_f32 && dim <= 2; }
