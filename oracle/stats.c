/*
 * oracle/stats.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Faithful f32 restatement of the reference's diagnostics (stats.rs):
 *   splitcat :396-402   split_rhat_mean_ess :416-423   rhat :425-427 (sqrt(W/var+), quirk Q7)
 *   withinvar :429-477  ess :496-546   autocov :548-554   autocov_fft :576-620   autocov_bf :632-654
 *   basic_stats :310-336   ChainTracker :26-141   collect_rhat / withinvar_from_cs :150-178
 *   MultiChainTracker :189-306
 * Summation order follows ndarray 0.16.1 where it matters for f32 rounding: contiguous 1-D sums use the
 * 8-accumulator `unrolled_fold`, strided lanes and iterator `.sum()` are sequential.  rustfft 6.4.1 is
 * replaced by a plain radix-2 complex FFT (same transform, different rounding; only tolerance-level
 * agreement with the reference is claimed for n > 100).
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ndarray numeric_util::unrolled_fold for f32 addition */
static float nd_sum_contig(const float *xs, size_t n)
{
    float acc = 0.0f;
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (n >= 8) {
        for (int k = 0; k < 8; ++k)
            p[k] = p[k] + xs[k];
        xs += 8;
        n -= 8;
    }
    acc = acc + (p[0] + p[4]);
    acc = acc + (p[1] + p[5]);
    acc = acc + (p[2] + p[6]);
    acc = acc + (p[3] + p[7]);
    for (size_t i = 0; i < n; ++i)
        acc = acc + xs[i];
    return acc;
}

/* ArrayBase::sum of a 1-D lane: contiguous -> unrolled, strided -> sequential fold */
static float nd_sum(const float *xs, size_t n, size_t stride)
{
    if (stride == 1)
        return nd_sum_contig(xs, n);
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i)
        acc = acc + xs[i * stride];
    return acc;
}

/* stats.rs:632-654 */
void o_autocov_bf(const float *data, size_t n, size_t d, float *out)
{
    float *col = (float *)malloc(sizeof(float) * (n ? n : 1));
    for (size_t c = 0; c < d; ++c) {
        float mean = nd_sum(data + c, n, d) / (float)n;
        for (size_t t = 0; t < n; ++t)
            col[t] = data[t * d + c] - mean;
        for (size_t lag = 0; lag < n; ++lag) {
            float sum_lag = 0.0f;
            for (size_t t = 0; t < n - lag; ++t)
                sum_lag += col[t] * col[t + lag];
            out[lag * d + c] = sum_lag / (float)n;
        }
    }
    free(col);
}

/* twiddles of the radix-2 FFT below: for every stage length len the factors (float)cos(ang k), (float)sin(ang k),
 * k < len / 2, ang = sign 2 pi / len -- the values the loop used to evaluate in place, tabulated once per transform
 * length (the table for length n holds all its stages back to back: n - 1 entries per sign). */
typedef struct fft_tab {
    size_t n;
    float *wr[2], *wi[2]; /* [0] forward (sign -1), [1] inverse (sign +1) */
    struct fft_tab *next;
} fft_tab;
static fft_tab *g_fft_tabs = NULL;
static pthread_mutex_t g_fft_mu = PTHREAD_MUTEX_INITIALIZER;

static const fft_tab *fft_table(size_t n)
{
    pthread_mutex_lock(&g_fft_mu);
    fft_tab *t = g_fft_tabs;
    while (t && t->n != n)
        t = t->next;
    if (!t) {
        t = (fft_tab *)malloc(sizeof(fft_tab));
        t->n = n;
        for (int sg = 0; sg < 2; ++sg) {
            t->wr[sg] = (float *)malloc(sizeof(float) * (n ? n : 1));
            t->wi[sg] = (float *)malloc(sizeof(float) * (n ? n : 1));
            size_t o = 0;
            for (size_t len = 2; len <= n; len <<= 1) {
                double ang = (sg ? 1.0 : -1.0) * 2.0 * M_PI / (double)len;
                for (size_t k = 0; k < len / 2; ++k, ++o) {
                    t->wr[sg][o] = (float)cos(ang * (double)k);
                    t->wi[sg][o] = (float)sin(ang * (double)k);
                }
            }
        }
        t->next = g_fft_tabs;
        g_fft_tabs = t;
    }
    pthread_mutex_unlock(&g_fft_mu);
    return t;
}

/* in-place iterative radix-2 FFT on interleaved complex f32; sign = -1 forward, +1 inverse (unnormalised) */
static void fft_radix2(float *re, float *im, size_t n, int sign)
{
    const fft_tab *tab = fft_table(n);
    const float *twr = tab->wr[sign > 0], *twi = tab->wi[sign > 0];
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i];
            re[i] = re[j];
            re[j] = t;
            t = im[i];
            im[i] = im[j];
            im[j] = t;
        }
    }
    size_t o = 0;
    for (size_t len = 2; len <= n; o += len / 2, len <<= 1) {
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                float wr = twr[o + k], wi = twi[o + k];
                size_t a = i + k, b = i + k + len / 2;
                float xr = re[b] * wr - im[b] * wi;
                float xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr;
                im[b] = im[a] - xi;
                re[a] = re[a] + xr;
                im[a] = im[a] + xi;
            }
        }
    }
}

/* stats.rs:576-620 */
void o_autocov_fft(const float *data, size_t n, size_t d, float *out)
{
    size_t n_padded = 1;
    while (n_padded < 2 * n - 1)
        n_padded <<= 1;
    float *re = (float *)malloc(sizeof(float) * n_padded);
    float *im = (float *)malloc(sizeof(float) * n_padded);
    for (size_t c = 0; c < d; ++c) {
        float mean = nd_sum(data + c, n, d) / (float)n;
        for (size_t t = 0; t < n; ++t) {
            re[t] = data[t * d + c] - mean;
            im[t] = 0.0f;
        }
        for (size_t t = n; t < n_padded; ++t)
            re[t] = im[t] = 0.0f;
        fft_radix2(re, im, n_padded, -1);
        for (size_t t = 0; t < n_padded; ++t) { /* x *= conj(x) */
            re[t] = re[t] * re[t] + im[t] * im[t];
            im[t] = 0.0f;
        }
        fft_radix2(re, im, n_padded, +1);
        for (size_t t = 0; t < n; ++t)
            out[t * d + c] = re[t] / (float)n_padded / (float)n;
    }
    free(re);
    free(im);
}

/* stats.rs:548-554 */
static void autocov(const float *data, size_t n, size_t d, float *out)
{
    if (n <= 100)
        o_autocov_bf(data, n, d, out);
    else
        o_autocov_fft(data, n, d, out);
}

/* worker threads for the per-chain autocovariances of o_split_rhat_mean_ess (0 / 1: the plain serial loop); the result
 * does not depend on it */
static int g_stats_threads = 1;
void o_stats_set_threads(int n) { g_stats_threads = n < 1 ? 1 : n; }

typedef struct {
    const float *sp;
    float *out;
    size_t half, p, lo, hi;
} acov_job;
static void *acov_worker(void *arg)
{
    acov_job *j = (acov_job *)arg;
    for (size_t ch = j->lo; ch < j->hi; ++ch)
        autocov(j->sp + ch * j->half * j->p, j->half, j->p, j->out + ch * j->half * j->p);
    return NULL;
}

/* stats.rs:416-546 */
void o_split_rhat_mean_ess(const float *sample, size_t c, size_t n, size_t p, float *rhat, float *ess)
{
    /* splitcat :396-402 -> [2c, half, p]; an odd n drops the middle draw */
    size_t half = n / 2;
    size_t C = 2 * c;
    float *sp = (float *)malloc(sizeof(float) * (C * half * p + 1));
    for (size_t ch = 0; ch < c; ++ch) {
        memcpy(sp + ch * half * p, sample + ch * n * p, sizeof(float) * half * p);
        memcpy(sp + (c + ch) * half * p, sample + (ch * n + (n - half)) * p, sizeof(float) * half * p);
    }
    float *within = (float *)malloc(sizeof(float) * p);
    float *var = (float *)malloc(sizeof(float) * p);
    float *chain_means = (float *)malloc(sizeof(float) * C);
    float *tmp = (float *)malloc(sizeof(float) * C);
    /* withinvar :429-477  (here `c` of the reference = C = 2*chains, `n` = half) */
    for (size_t k = 0; k < p; ++k) {
        for (size_t ch = 0; ch < C; ++ch)
            chain_means[ch] = nd_sum(sp + ch * half * p + k, half, p) / (float)half;
        float overall_mean = nd_sum_contig(chain_means, C) / (float)C;
        for (size_t ch = 0; ch < C; ++ch) {
            float df = chain_means[ch] - overall_mean;
            tmp[ch] = df * df;
        }
        float b = nd_sum_contig(tmp, C) * ((float)half / (float)(C - 1));
        for (size_t ch = 0; ch < C; ++ch) {
            float cm = chain_means[ch];
            float sq = 0.0f;
            for (size_t t = 0; t < half; ++t) {
                float v = sp[(ch * half + t) * p + k];
                sq += (v - cm) * (v - cm);
            }
            tmp[ch] = sq / (float)half;
        }
        float w = nd_sum_contig(tmp, C) / (float)C;
        float v = (((float)half - 1.0f) / (float)half) * w + b / (float)half;
        within[k] = w;
        var[k] = v;
        rhat[k] = sqrtf(w / v); /* :425-427 */
    }
    /* ess :496-546 */
    float *avg_rho = (float *)calloc(half * p + 1, sizeof(float));
    float *rho_c = (float *)malloc(sizeof(float) * (half * p + 1));
    if (g_stats_threads <= 1) {
        for (size_t ch = 0; ch < C; ++ch) {
            autocov(sp + ch * half * p, half, p, rho_c);
            for (size_t i = 0; i < half * p; ++i)
                avg_rho[i] = avg_rho[i] + rho_c[i]; /* mean_axis(Axis(0)): sequential over chains */
        }
    } else {
        /* the same numbers with the per-chain autocovariances of a batch computed on several threads; they are added
         * to the running sum one chain after the other, in chain order, as above: bit-identical */
        const size_t batch = 64 * (size_t)g_stats_threads;
        float *buf = (float *)malloc(sizeof(float) * batch * half * p);
        for (size_t c0 = 0; c0 < C; c0 += batch) {
            size_t nb = C - c0 < batch ? C - c0 : batch;
            acov_job jobs[64];
            pthread_t th[64];
            int nt = g_stats_threads > 64 ? 64 : g_stats_threads;
            for (int t = 0; t < nt; ++t) {
                jobs[t].sp = sp + c0 * half * p;
                jobs[t].out = buf;
                jobs[t].half = half;
                jobs[t].p = p;
                jobs[t].lo = nb * (size_t)t / (size_t)nt;
                jobs[t].hi = nb * (size_t)(t + 1) / (size_t)nt;
                pthread_create(&th[t], NULL, acov_worker, &jobs[t]);
            }
            for (int t = 0; t < nt; ++t)
                pthread_join(th[t], NULL);
            for (size_t ch = 0; ch < nb; ++ch)
                for (size_t i = 0; i < half * p; ++i)
                    avg_rho[i] = avg_rho[i] + buf[ch * half * p + i];
        }
        free(buf);
    }
    for (size_t i = 0; i < half * p; ++i)
        avg_rho[i] = avg_rho[i] / (float)C;
    for (size_t k = 0; k < p; ++k) {
        /* rho = -( (-avg_rho + within) / var ) + 1 */
        float *rho = rho_c; /* reuse */
        for (size_t t = 0; t < half; ++t) {
            float diff = -avg_rho[t * p + k] + within[k];
            rho[t] = -(diff / var[k]) + 1.0f;
        }
        float min = (half >= 2) ? rho[0] + rho[1] : 0.0f;
        float out = 0.0f;
        for (size_t t = 0; t + 1 < half; t += 2) { /* windows_with_stride(2, 2) */
            float p_t = rho[t] + rho[t + 1];
            if (p_t <= 0.0f)
                break;
            if (p_t > min)
                p_t = min;
            min = p_t;
            out += p_t;
        }
        float tau = -1.0f + 2.0f * out;
        ess[k] = (1.0f / tau) * (float)C * (float)half;
    }
    free(sp);
    free(within);
    free(var);
    free(chain_means);
    free(tmp);
    free(avg_rho);
    free(rho_c);
}

static int cmp_desc(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    if (y < x)
        return -1;
    if (y > x)
        return 1;
    return 0;
}

/* stats.rs:310-336 : sorted descending; min = last, median = [len/2], max = first, std ddof = 1 */
void o_basic_stats(const float *data, size_t len, float out[5])
{
    float *s = (float *)malloc(sizeof(float) * len);
    memcpy(s, data, sizeof(float) * len);
    qsort(s, len, sizeof(float), cmp_desc);
    out[0] = s[len - 1];
    out[1] = s[len / 2];
    out[2] = s[0];
    float mean = nd_sum_contig(s, len) / (float)len;
    out[3] = mean;
    /* ndarray std(1.0): Welford */
    float m = 0.0f, sum_sq = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        float count = (float)(i + 1);
        float delta = s[i] - m;
        m += delta / count;
        sum_sq = (s[i] - m) * delta + sum_sq;
    }
    out[4] = sqrtf(sum_sq / ((float)len - 1.0f));
    free(s);
}

/* stats.rs:189-306 */
void o_multichain_tracker(const float *states, size_t steps, size_t chains, size_t params, float *rhat,
                          float *p_accept_out)
{
    const float ALPHA = 0.01f;
    size_t cp = chains * params;
    float *mean = (float *)calloc(cp, sizeof(float));
    float *mean_sq = (float *)calloc(cp, sizeof(float));
    float *last = (float *)calloc(cp, sizeof(float));
    float p_accept = 0.0f;
    size_t n_i = 0;
    for (size_t st = 0; st < steps; ++st) {
        const float *x = states + st * cp;
        n_i += 1;
        float n = (float)n_i;
        for (size_t i = 0; i < cp; ++i) {
            mean[i] = (mean[i] * (n - 1.0f) + x[i]) / n;
            if (n_i == 1)
                mean_sq[i] = x[i] * x[i];
            else
                mean_sq[i] = (mean_sq[i] * (n - 1.0f) + x[i] * x[i]) / n;
        }
        for (size_t c = 0; c < chains; ++c) {
            int ne = 0;
            for (size_t k = 0; k < params; ++k)
                if (x[c * params + k] != last[c * params + k])
                    ne = 1;
            p_accept = (1.0f - ALPHA) * p_accept + ALPHA * (float)ne;
        }
        memcpy(last, x, sizeof(float) * cp);
    }
    /* within_and_var :288-306 */
    float n = (float)n_i, nch = (float)chains;
    for (size_t k = 0; k < params; ++k) {
        float msum = 0.0f;
        for (size_t c = 0; c < chains; ++c)
            msum = msum + mean[c * params + k];
        float mean_chain = msum / nch;
        float fac = n / (nch - 1.0f);
        float between = 0.0f;
        for (size_t c = 0; c < chains; ++c) {
            float df = mean[c * params + k] - mean_chain;
            between = between + df * df;
        }
        between = between * fac;
        float wsum = 0.0f;
        for (size_t c = 0; c < chains; ++c) {
            float m = mean[c * params + k];
            float sm2 = (mean_sq[c * params + k] - m * m) * n / (n - 1.0f);
            wsum = wsum + sm2;
        }
        float within = wsum / nch;
        float var = within * ((n - 1.0f) / n) + between * (1.0f / n);
        rhat[k] = sqrtf(var / within);
    }
    if (p_accept_out)
        *p_accept_out = p_accept;
    free(mean);
    free(mean_sq);
    free(last);
}

static void chain_trackers_core(const float *init, const float *states, size_t chains, size_t steps, size_t params,
                                float *rhat, float *p_accept_out, float *within_out, float *var_out);

/* stats.rs:26-141 (ChainTracker) + :150-178 (collect_rhat) */
void o_chain_trackers_rhat(const float *init, const float *states, size_t chains, size_t steps, size_t params,
                           float *rhat, float *p_accept_out)
{
    chain_trackers_core(init, states, chains, steps, params, rhat, p_accept_out, NULL, NULL);
}

/* ess_from_chainstats stats.rs:668-671: ess(sample, within, var) (:496-546, NOT split) with (within, var) =
 * withinvar_from_cs (:155-178) of per-chain ChainTrackers that were constructed on init[chains, params] and fed
 * tracked[chains, steps, params]; sample [chains, n, params] (the collected part of the run) */
void o_ess_from_chainstats(const float *sample, size_t chains, size_t n, size_t params, const float *init,
                           const float *tracked, size_t steps, float *ess)
{
    float *within = (float *)malloc(sizeof(float) * params);
    float *var = (float *)malloc(sizeof(float) * params);
    float *rhat = (float *)malloc(sizeof(float) * params);
    chain_trackers_core(init, tracked, chains, steps, params, rhat, NULL, within, var);
    float *avg_rho = (float *)calloc(n * params + 1, sizeof(float));
    float *rho_c = (float *)malloc(sizeof(float) * (n * params + 1));
    for (size_t ch = 0; ch < chains; ++ch) {
        autocov(sample + ch * n * params, n, params, rho_c);
        for (size_t i = 0; i < n * params; ++i)
            avg_rho[i] = avg_rho[i] + rho_c[i];
    }
    for (size_t i = 0; i < n * params; ++i)
        avg_rho[i] = avg_rho[i] / (float)chains;
    for (size_t k = 0; k < params; ++k) {
        float *rho = rho_c;
        for (size_t t = 0; t < n; ++t) {
            float diff = -avg_rho[t * params + k] + within[k];
            rho[t] = -(diff / var[k]) + 1.0f;
        }
        float min = (n >= 2) ? rho[0] + rho[1] : 0.0f;
        float out = 0.0f;
        for (size_t t = 0; t + 1 < n; t += 2) {
            float p_t = rho[t] + rho[t + 1];
            if (p_t <= 0.0f)
                break;
            if (p_t > min)
                p_t = min;
            min = p_t;
            out += p_t;
        }
        float tau = -1.0f + 2.0f * out;
        ess[k] = (1.0f / tau) * (float)chains * (float)n;
    }
    free(within);
    free(var);
    free(rhat);
    free(avg_rho);
    free(rho_c);
}

static void chain_trackers_core(const float *init, const float *states, size_t chains, size_t steps, size_t params,
                                float *rhat, float *p_accept_out, float *within_out, float *var_out)
{
    const float ALPHA = 0.01f;
    float *means = (float *)calloc(chains * params, sizeof(float));
    float *sm2s = (float *)calloc(chains * params, sizeof(float));
    float *mean_sq = (float *)malloc(sizeof(float) * params);
    float *last = (float *)malloc(sizeof(float) * params);
    for (size_t c = 0; c < chains; ++c) {
        float *mean = means + c * params;
        memset(mean_sq, 0, sizeof(float) * params);
        memcpy(last, init + c * params, sizeof(float) * params);
        float p_accept = -1.0f;
        for (size_t st = 0; st < steps; ++st) {
            const float *x = states + (c * steps + st) * params;
            float n = (float)(st + 1);
            for (size_t k = 0; k < params; ++k) {
                mean[k] = (mean[k] * (n - 1.0f) + x[k]) / n;
                if (st == 0)
                    mean_sq[k] = x[k] * x[k];
                else
                    mean_sq[k] = (mean_sq[k] * (n - 1.0f) + x[k] * x[k]) / n;
            }
            /* first step compares coordinate 0 only (quirk Q12); a 1-D array has a single "row" */
            float p_start = (p_accept >= 0.0f) ? p_accept : (float)(x[0] != last[0]);
            int ne = 0;
            for (size_t k = 0; k < params; ++k)
                if (x[k] != last[k])
                    ne = 1;
            p_accept = (1.0f - ALPHA) * p_start + ALPHA * (float)ne;
            memcpy(last, x, sizeof(float) * params);
        }
        float n = (float)steps;
        for (size_t k = 0; k < params; ++k)
            sm2s[c * params + k] = (mean_sq[k] - mean[k] * mean[k]) * n / (n - 1.0f);
        if (p_accept_out)
            p_accept_out[c] = p_accept;
    }
    /* withinvar_from_cs :155-178 : between divides by chains*params - 1 (quirk Q9) */
    float nch = (float)chains;
    float nmean = (float)steps; /* every chain has n = steps */
    for (size_t k = 0; k < params; ++k) {
        float w = 0.0f, gm = 0.0f;
        for (size_t c = 0; c < chains; ++c) {
            w = w + sm2s[c * params + k];
            gm = gm + means[c * params + k];
        }
        float within = w / nch;
        float global_mean = gm / nch;
        float ss = 0.0f;
        for (size_t c = 0; c < chains; ++c) {
            float df = means[c * params + k] - global_mean;
            ss = ss + df * df;
        }
        float between = ss / (float)(chains * params - 1);
        float var = between + within * ((nmean - 1.0f) / nmean);
        rhat[k] = sqrtf(var / within);
        if (within_out)
            within_out[k] = within;
        if (var_out)
            var_out[k] = var;
    }
    free(means);
    free(sm2s);
    free(mean_sq);
    free(last);
}
