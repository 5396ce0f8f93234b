/*
 * oracle/rand_compat.c -- TEST INFRASTRUCTURE ONLY. See rand_compat.h.
 *
 * Algorithms restated (not copied; the crates are not on disk):
 *   - xoshiro256++ (Blackman & Vigna) as used by rand 0.9 SmallRng on 64-bit targets
 *   - SplitMix64 seeding (rand_xoshiro / rand 0.9 `seed_from_u64` for xoshiro)
 *   - Marsaglia & Tsang ziggurat, 256 layers, one u64 per attempt, as in rand_distr 0.5 `utils::ziggurat`
 * The ziggurat tables are regenerated from the published recurrence
 *   X[0] = V / f(R), X[1] = R, X[k] = f^-1(V / X[k-1] + f(X[k-1])), X[256] = 0, F[k] = f(X[k])
 * (rand_distr ships the same numbers as literals, produced by the same recurrence in double precision).
 */
#include "rand_compat.h"

#include <math.h>
#include <string.h>

#define ZIG_N 256

static const double ZIG_NORM_R = 3.654152885361008796;
static const double ZIG_NORM_V = 0.00492867323399;
static const double ZIG_EXP_R = 7.69711747013104972;
static const double ZIG_EXP_V = 0.0039496598225815571993;

static double g_norm_x[ZIG_N + 1], g_norm_f[ZIG_N + 1];
static double g_exp_x[ZIG_N + 1], g_exp_f[ZIG_N + 1];

static double norm_pdf(double x) { return exp(-x * x / 2.0); }
static double norm_pdf_inv(double y) { return sqrt(-2.0 * log(y)); }
static double exp_pdf(double x) { return exp(-x); }
static double exp_pdf_inv(double y) { return -log(y); }

static void build_table(double r, double v, double (*f)(double), double (*finv)(double), double *x,
                        double *fx)
{
    x[0] = v / f(r);
    x[1] = r;
    for (int i = 2; i < ZIG_N; ++i) {
        double last = x[i - 1];
        x[i] = finv(v / last + f(last));
    }
    x[ZIG_N] = 0.0;
    for (int i = 0; i <= ZIG_N; ++i)
        fx[i] = f(x[i]);
}

__attribute__((constructor)) static void rc_init_tables(void)
{
    build_table(ZIG_NORM_R, ZIG_NORM_V, norm_pdf, norm_pdf_inv, g_norm_x, g_norm_f);
    build_table(ZIG_EXP_R, ZIG_EXP_V, exp_pdf, exp_pdf_inv, g_exp_x, g_exp_f);
}

const double *rc_zig_norm_x(void) { return g_norm_x; }
const double *rc_zig_norm_f(void) { return g_norm_f; }
const double *rc_zig_exp_x(void) { return g_exp_x; }
const double *rc_zig_exp_f(void) { return g_exp_f; }

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void rc_seed_from_u64(rc_rng *r, uint64_t state)
{
    for (int i = 0; i < 4; ++i) {
        state += 0x9e3779b97f4a7c15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        r->s[i] = z ^ (z >> 31);
    }
}

uint64_t rc_next_u64(rc_rng *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

uint32_t rc_next_u32(rc_rng *r) { return (uint32_t)(rc_next_u64(r) >> 32); }

double rc_uniform_f64(rc_rng *r) { return (double)(rc_next_u64(r) >> 11) * 0x1.0p-53; }

float rc_uniform_f32(rc_rng *r) { return (float)(rc_next_u32(r) >> 8) * 0x1.0p-24f; }

static inline double f64_with_exponent(uint64_t mantissa52, int exponent)
{
    uint64_t bits = mantissa52 | ((uint64_t)(1023 + exponent) << 52);
    double d;
    memcpy(&d, &bits, sizeof d);
    return d;
}

double rc_open01_f64(rc_rng *r)
{
    /* Open01 for f64: 52 random mantissa bits in [1,2), minus (1 - eps/2) -> (0,1) */
    uint64_t v = rc_next_u64(r);
    return f64_with_exponent(v >> 12, 0) - (1.0 - 0x1.0p-52 / 2.0);
}

double rc_standard_normal(rc_rng *r)
{
    for (;;) {
        uint64_t bits = rc_next_u64(r);
        unsigned i = (unsigned)(bits & 0xff);
        /* symmetric: [2,4) - 3 -> [-1,1) */
        double u = f64_with_exponent(bits >> 12, 1) - 3.0;
        double x = u * g_norm_x[i];
        if (fabs(x) < g_norm_x[i + 1])
            return x;
        if (i == 0) {
            /* tail */
            double tx = 1.0, ty = 0.0;
            while (-2.0 * ty < tx * tx) {
                double x_ = rc_open01_f64(r);
                double y_ = rc_open01_f64(r);
                tx = log(x_) / ZIG_NORM_R;
                ty = log(y_);
            }
            return (u < 0.0) ? tx - ZIG_NORM_R : ZIG_NORM_R - tx;
        }
        if (g_norm_f[i + 1] + (g_norm_f[i] - g_norm_f[i + 1]) * rc_uniform_f64(r) < norm_pdf(x))
            return x;
    }
}

double rc_exp1(rc_rng *r)
{
    for (;;) {
        uint64_t bits = rc_next_u64(r);
        unsigned i = (unsigned)(bits & 0xff);
        double u = f64_with_exponent(bits >> 12, 0) - (1.0 - 0x1.0p-52 / 2.0);
        double x = u * g_exp_x[i];
        if (x < g_exp_x[i + 1])
            return x;
        if (i == 0)
            return ZIG_EXP_R - log(rc_uniform_f64(r));
        if (g_exp_f[i + 1] + (g_exp_f[i] - g_exp_f[i + 1]) * rc_uniform_f64(r) < exp_pdf(x))
            return x;
    }
}
