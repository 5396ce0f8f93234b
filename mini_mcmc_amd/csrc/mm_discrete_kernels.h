/*
 * mm_discrete_kernels.h -- integer-state Metropolis-Hastings around a model of the caller's own (mm_rtc.hip:
 * mmcmc_discrete_register_source).
 *
 * MHMarkovChain<S = i32> is generic over `Target<i32, f64>` and `Proposal<i32, f64>` (metropolis_hastings.rs:28-48); the
 * reference's own integration tests implement both traits by hand (tests/metrohast_poisson_test.rs:18-85, :150-212).
 * The GPU analogue is one functor,
 *     struct mmcmc_user_discrete {
 *         MM_HD static double logp(const double *params, int k);              // Target::unnorm_logp
 *         MM_HD static int sample(const double *params, int current, int step); // Proposal::sample; step = +1 / -1, each with probability 1/2
 *         MM_HD static double logq(const double *params, int from, int to);   // Proposal::logp = log q(to | from)
 *     };
 * compiled into the skeleton below: one chain per lane, the whole run in one launch, samples staged per wave in LDS and
 * written as 16-byte pieces (as mm_discrete_kernel).  The transition is MHMarkovChain::step word for word
 * (metropolis_hastings.rs:303-315), on the stream of the built-in models (mm_discrete.h: direction = top bit of word 0 of
 * block 0, accept uniform = u53 of words 2, 3), so a user restatement of a built-in model reproduces it bit for bit.
 */
#ifndef MM_DISCRETE_KERNELS_H
#define MM_DISCRETE_KERNELS_H

#include "mm_kernels.h" /* first: it brings hip_runtime.h (MM_HD needs __forceinline__) when the unit is built by hipcc */
#include "mm_discrete.h"

struct mm_discrete_user_args {
    double params[8];
    int32_t *state;                 /* [C] */
    int32_t *out;                   /* [C, n_collect] or NULL */
    unsigned long long *accept;     /* [C] running totals */
    unsigned long long n_chains, seed, chain_offset;
    unsigned int iter0, n_discard, n_collect;
};

template <class Model>
MM_HD int mm_discrete_user_step(const double *params, int32_t *x, const mm_u32x4 &b)
{
    const int step = (b.w[0] >> 31) ? 1 : -1;
    const int cur = *x;
    const int prop = Model::sample(params, cur, step);
    const double current_lp = Model::logp(params, cur);
    const double proposed_lp = Model::logp(params, prop);
    const double qf = Model::logq(params, cur, prop);
    const double qb = Model::logq(params, prop, cur);
    const double log_accept_ratio = (proposed_lp + qb) - (current_lp + qf);
    const double u = mm_u53(b.w[2], b.w[3]);
    if (mm_ratio_exceeds_ln_u(log_accept_ratio, u)) { /* log_accept_ratio > ln u (metropolis_hastings.rs:311) */
        *x = prop;
        return 1;
    }
    return 0;
}

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
template <class Model>
__device__ __forceinline__ void mm_discrete_user_body(const mm_discrete_user_args &a)
{
    using Tile = mm_tile<int32_t, 1>;
    constexpr unsigned int TILE = Tile::tile_t;
    constexpr int STRIDE = Tile::stride;
    __shared__ __attribute__((aligned(16))) int32_t tile[64 * STRIDE];
    const int lane = threadIdx.x;
    const unsigned long long c0 = (unsigned long long)blockIdx.x * 64, c = c0 + lane;
    const bool active = c < a.n_chains;
    int32_t x = active ? a.state[c] : 0;
    unsigned long long n_acc = 0;
    const unsigned int total = a.n_discard + a.n_collect;
    unsigned int col = 0, t0 = 0;
    for (unsigned int t = 0; t < total; ++t) {
        /* inactive lanes of a ragged last wave take no step: a model need not be defined at state 0 */
        if (active)
            n_acc += (unsigned long long)mm_discrete_user_step<Model>(a.params, &x, mm_block(a.seed, a.chain_offset + c, a.iter0 + t, 0u));
        if (t >= a.n_discard && a.out) {
            tile[lane * STRIDE + col++] = x;
            if (col == TILE || t + 1 == total) {
                mm_flush_tile_raw<int32_t, 1>(a.out, a.n_collect, a.n_chains, tile, lane, c0, t0, col);
                t0 += col;
                col = 0;
            }
        }
    }
    if (active) {
        a.state[c] = x;
        a.accept[c] += n_acc;
    }
}
#endif

#endif /* MM_DISCRETE_KERNELS_H */
