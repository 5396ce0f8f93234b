/*
 * oracle/engine_host.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Host (g++) build of the engine's per-chain transition functions -- the very headers the HIP kernels are
 * compiled from (mini_mcmc_amd/csrc/mm_samplers.h, mm_targets.h, mm_rng.h, mm_math.h) -- driven by a plain loop
 * over chains and iterations.  Because those headers use only IEEE +,-,*,/,sqrt and explicit fma(), and both sides
 * are built with -ffp-contract=off, this library must reproduce the GPU's samples, accept masks and accept counts
 * BIT FOR BIT.  It is what the `-m gpu` parity tests compare the device results with at every size; the
 * independent, reference-ordered restatement (mh_hmc.c, nuts.c, pinned by the reference's known-answer tests) is
 * compared with it on the CPU in tests/test_step_parity.py.  Nothing in the product links or loads this file.
 */
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../mini_mcmc_amd/csrc/mm_params.h"
#include "../mini_mcmc_amd/csrc/mm_nuts.h"
#include "../mini_mcmc_amd/csrc/mm_nuts_dims.h"
#include "../mini_mcmc_amd/csrc/mm_nuts_generic.h"
#include "../mini_mcmc_amd/csrc/mm_samplers.h"
#include "../mini_mcmc_amd/csrc/mm_generic.h"

namespace {

template <class T, class Tgt, int HMC>
void run_chains(const mm_tparams<T> &P, T scale, int n_leapfrog, T *state, size_t n_chains, uint64_t seed,
                uint64_t chain_offset, uint32_t iter0, size_t n_collect, size_t n_discard, T *out, uint64_t *accept,
                int n_threads)
{
    constexpr int D = Tgt::dim;
    auto work = [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
            T x[D], g[D], lp;
            for (int i = 0; i < D; ++i)
                x[i] = state[c * D + i];
            lp = HMC ? Tgt::logp_grad(P, x, g) : Tgt::logp(P, x);
            uint64_t n_acc = 0;
            uint32_t it = iter0;
            for (size_t t = 0; t < n_discard + n_collect; ++t, ++it) {
                int acc = HMC ? mm_hmc_step<T, Tgt>(P, scale, n_leapfrog, x, &lp, g, seed, chain_offset + c, it)
                              : mm_mh_step<T, Tgt>(P, scale, x, &lp, seed, chain_offset + c, it);
                n_acc += (uint64_t)acc;
                if (t >= n_discard && out)
                    for (int i = 0; i < D; ++i)
                        out[(c * n_collect + (t - n_discard)) * D + i] = x[i];
            }
            for (int i = 0; i < D; ++i)
                state[c * D + i] = x[i];
            if (accept)
                accept[c] = n_acc;
        }
    };
    if (n_threads <= 1 || n_chains < 2) {
        work(0, n_chains);
        return;
    }
    std::vector<std::thread> th;
    size_t nt = std::min<size_t>((size_t)n_threads, n_chains);
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back(work, n_chains * t / nt, n_chains * (t + 1) / nt);
    for (auto &t : th)
        t.join();
}

/* the run-time-dimension path (mm_generic.h) on the host: the store is a plain array per chain */
template <class T>
int run_generic(int hmc, int kind, int dim, const mm_tparams<T> &P, T scale, int L, T *state, size_t n, uint64_t seed,
                uint64_t off, uint32_t it0, size_t nc, size_t nd, T *out, uint64_t *acc, int nth)
{
    if (!mm_generic_kind_ok(kind) || dim < 1)
        return -2;
    auto work = [&](size_t lo, size_t hi) {
        std::vector<T> buf((size_t)MM_GV_HMC * dim);
        mm_gstore<T, T *> s;
        s.base = buf.data();
        s.stride = 1;
        s.dim = dim;
        for (size_t c = lo; c < hi; ++c) {
            for (int i = 0; i < dim; ++i)
                s.st(MM_GV_X, i, state[c * dim + i]);
            T lp = mm_gen_logp_grad<T>(kind, P, s, MM_GV_X, hmc ? MM_GV_G : -1);
            uint64_t n_acc = 0;
            uint32_t it = it0;
            for (size_t t = 0; t < nd + nc; ++t, ++it) {
                const int a = hmc ? mm_gen_hmc_step<T>(kind, P, scale, L, s, &lp, seed, off + c, it)
                                  : mm_gen_mh_step<T>(kind, P, scale, s, &lp, seed, off + c, it);
                n_acc += (uint64_t)a;
                if (t >= nd && out)
                    for (int i = 0; i < dim; ++i)
                        out[(c * nc + (t - nd)) * dim + i] = s.ld(MM_GV_X, i);
            }
            for (int i = 0; i < dim; ++i)
                state[c * dim + i] = s.ld(MM_GV_X, i);
            if (acc)
                acc[c] = n_acc;
        }
    };
    if (nth <= 1 || n < 2) {
        work(0, n);
        return 0;
    }
    std::vector<std::thread> th;
    size_t nt = std::min<size_t>((size_t)nth, n);
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th)
        t.join();
    return 0;
}

template <class T, int KIND, int D>
int run_kd(int hmc, const mm_tparams<T> &P, T scale, int L, T *state, size_t n, uint64_t seed, uint64_t off,
           uint32_t it0, size_t nc, size_t nd, T *out, uint64_t *acc, int nth)
{
    using Tgt = mm_target<T, KIND, D>;
    if (hmc)
        run_chains<T, Tgt, 1>(P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);
    else
        run_chains<T, Tgt, 0>(P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);
    return 0;
}

#define ND_CASES(KIND)                                                                                            \
    switch (dim) {                                                                                                \
    case 1: return run_kd<T, KIND, 1>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 2: return run_kd<T, KIND, 2>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 3: return run_kd<T, KIND, 3>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 4: return run_kd<T, KIND, 4>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 5: return run_kd<T, KIND, 5>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 6: return run_kd<T, KIND, 6>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 7: return run_kd<T, KIND, 7>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 8: return run_kd<T, KIND, 8>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);          \
    case 16: return run_kd<T, KIND, 16>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);        \
    case 32: return run_kd<T, KIND, 32>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);        \
    default: return run_generic<T>(hmc, KIND, dim, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);   \
    }

template <class T>
int run_t(int hmc, int kind, int dim, const double params[8], const double *matrix, double scale_d, int L, T *state,
          size_t n, uint64_t seed, uint64_t off, uint32_t it0, size_t nc, size_t nd, T *out, uint64_t *acc, int nth)
{
    mm_tparams<T> P;
    if (mm_fill_params<T>(kind, params, &P) != 0)
        return -1;
    std::vector<T> mat;
    if (kind == MM_GAUSSIAN_ND) {
        if (!matrix)
            return -1;
        mat.resize((size_t)dim * dim);
        for (size_t i = 0; i < mat.size(); ++i)
            mat[i] = (T)matrix[i];
        P.mat = mat.data();
    }
    T scale = (T)scale_d;
    if (hmc >= 2) /* samplers 2 / 3: MH / HMC through the run-time-dimension path whatever the dimension */
        return run_generic<T>(hmc - 2, kind, dim, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth);
    switch (kind) {
    case MM_GAUSSIAN2D:
        return dim == 2 ? run_kd<T, MM_GAUSSIAN2D, 2>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth) : -3;
    case MM_DIFFABLE_GAUSSIAN2D:
        return dim == 2 ? run_kd<T, MM_DIFFABLE_GAUSSIAN2D, 2>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth) : -3;
    case MM_ROSENBROCK2D:
        return dim == 2 ? run_kd<T, MM_ROSENBROCK2D, 2>(hmc, P, scale, L, state, n, seed, off, it0, nc, nd, out, acc, nth) : -3;
    case MM_ISOTROPIC_GAUSSIAN:
        ND_CASES(MM_ISOTROPIC_GAUSSIAN)
    case MM_ROSENBROCK_ND:
        ND_CASES(MM_ROSENBROCK_ND)
    case MM_STANDARD_NORMAL:
        ND_CASES(MM_STANDARD_NORMAL)
    case MM_GAUSSIAN_ND:
        ND_CASES(MM_GAUSSIAN_ND)
    default:
        return -2;
    }
}

} // namespace

extern "C" {

/* Run n_chains chains for n_discard + n_collect transitions on the host with the engine's own arithmetic.
 * sampler: 0 = MH (scale = proposal std), 1 = HMC (scale = step size); 2 / 3 = the same through the run-time-dimension
 * path (mm_generic.h), which dimensions without a fixed-size instance take anyway.  dtype: 0 = f32, 1 = f64.
 * state [n, dim] in/out; out [n, n_collect, dim] or NULL; accept [n] or NULL.  Returns 0, or <0 if unsupported. */
int eh_run(int sampler, int dtype, int kind, int dim, const double params[8], const double *matrix, double scale,
           int n_leapfrog, void *state, size_t n_chains, uint64_t seed, uint64_t chain_offset, uint32_t iter0,
           size_t n_collect, size_t n_discard, void *out, uint64_t *accept, int n_threads)
{
    if (dtype == 0)
        return run_t<float>(sampler, kind, dim, params, matrix, scale, n_leapfrog, (float *)state, n_chains, seed,
                            chain_offset, iter0, n_collect, n_discard, (float *)out, accept, n_threads);
    return run_t<double>(sampler, kind, dim, params, matrix, scale, n_leapfrog, (double *)state, n_chains, seed,
                         chain_offset, iter0, n_collect, n_discard, (double *)out, accept, n_threads);
}

} /* extern "C" */

/* NUTS with the engine's arithmetic: the same call sequence as the GPU path (mm_nuts_api.hip / mm_nuts_kernels.h):
 * init_chain, then n_pre unrecorded and n_rec recorded transitions, row 0 = initial position for run() with
 * n_discard == 0.  positions [n, dim] doubles in/out (converted to the tensor type on entry like the C ABI does),
 * adapt [n, 4] doubles in/out (epsilon, epsilon_bar, h_bar, mu), out [n, n_collect, dim] of the tensor type. */
/* eh_nuts_set_pairs: walk the trees leaf pair by leaf pair (mm_nuts_step_pairs, the order of mm_nuts_pair_kernel) */
static int g_nuts_pairs = 0;
extern "C" void eh_nuts_set_pairs(int on) { g_nuts_pairs = on; }

template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
static void nuts_chains(const mm_tparams<TT> &P, double *positions, size_t n, double tap, uint64_t seed, uint64_t off,
                        uint32_t m0, size_t n_collect, size_t n_discard, int progress, int max_depth, double *adapt,
                        TT *out, uint64_t *nlf, int nth)
{
    constexpr int D = Tgt::dim;
    const ST eps_tol = sizeof(ST) == 4 ? (ST)1.1920929e-7 : (ST)2.220446049250313e-16;
    unsigned n_pre, n_rec, write_initial;
    const size_t total = n_collect + n_discard;
    if (progress) { write_initial = 0; n_pre = (unsigned)n_discard; n_rec = (unsigned)n_collect; }
    else if (total == 0) { write_initial = 0; n_pre = n_rec = 0; }
    else if (n_discard == 0) { write_initial = n_collect > 0; n_pre = 0; n_rec = (unsigned)(n_collect > 0 ? n_collect - 1 : 0); }
    else { write_initial = 0; n_pre = (unsigned)(n_discard - 1); n_rec = (unsigned)n_collect; }
    auto work = [&](size_t lo, size_t hi) {
        std::vector<TT> vec(mm_nuts_stack<TT, ST, D>::vec_slots);
        std::vector<ST> al(mm_nuts_stack<TT, ST, D>::alpha_slots);
        std::vector<uint32_t> cn(mm_nuts_stack<TT, ST, D>::cnt_slots);
        mm_nuts_stack<TT, ST, D> stk;
        stk.vec = vec.data(); stk.alpha = al.data(); stk.cnt = cn.data(); stk.stride = 1;
        for (size_t c = lo; c < hi; ++c) {
            TT x[D];
            for (int i = 0; i < D; ++i) x[i] = (TT)positions[c * D + i];
            mm_nuts_adapt<ST> ad;
            ad.epsilon = (ST)adapt[4 * c]; ad.epsilon_bar = (ST)adapt[4 * c + 1];
            ad.h_bar = (ST)adapt[4 * c + 2]; ad.mu = (ST)adapt[4 * c + 3];
            mm_nuts_init_chain<TT, ST, Tgt, Red>(P, x, &ad, eps_tol, seed, off + c);
            size_t row = 0;
            uint64_t lf = 0;
            uint32_t m = m0;
            auto rec = [&]() { if (out) { for (int i = 0; i < D; ++i) out[(c * n_collect + row) * D + i] = x[i]; } ++row; };
            if (write_initial) rec();
            for (unsigned t = 0; t < n_pre + n_rec; ++t) {
                ++m;
                mm_nuts_info inf = g_nuts_pairs
                                       ? mm_nuts_step_pairs<TT, ST, Tgt, Red>(P, x, &ad, m, (uint32_t)n_discard, (ST)tap, max_depth, seed, off + c, stk)
                                       : mm_nuts_step<TT, ST, Tgt, Red>(P, x, &ad, m, (uint32_t)n_discard, (ST)tap, max_depth, seed, off + c, stk);
                lf += inf.n_leapfrog;
                if (t >= n_pre) rec();
            }
            for (int i = 0; i < D; ++i) positions[c * D + i] = (double)x[i];
            adapt[4 * c] = (double)ad.epsilon; adapt[4 * c + 1] = (double)ad.epsilon_bar;
            adapt[4 * c + 2] = (double)ad.h_bar; adapt[4 * c + 3] = (double)ad.mu;
            if (nlf) nlf[c] = lf;
        }
    };
    if (nth <= 1 || n < 2) { work(0, n); return; }
    std::vector<std::thread> th;
    size_t nt = std::min<size_t>((size_t)nth, n);
    for (size_t t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
}

/* the same at run-time dimension (mm_nuts_generic.h): the twin of the device's generic NUTS kernel */
template <class TT, class ST>
static void nuts_chains_generic(int kind, int D, const mm_tparams<TT> &P, double *positions, size_t n, double tap, uint64_t seed,
                                uint64_t off, uint32_t m0, size_t n_collect, size_t n_discard, int progress, int max_depth,
                                double *adapt, TT *out, uint64_t *nlf, int nth)
{
    const ST eps_tol = sizeof(ST) == 4 ? (ST)1.1920929e-7 : (ST)2.220446049250313e-16;
    unsigned n_pre, n_rec, write_initial;
    const size_t total = n_collect + n_discard;
    if (progress) { write_initial = 0; n_pre = (unsigned)n_discard; n_rec = (unsigned)n_collect; }
    else if (total == 0) { write_initial = 0; n_pre = n_rec = 0; }
    else if (n_discard == 0) { write_initial = n_collect > 0; n_pre = 0; n_rec = (unsigned)(n_collect > 0 ? n_collect - 1 : 0); }
    else { write_initial = 0; n_pre = (unsigned)(n_discard - 1); n_rec = (unsigned)n_collect; }
    auto work = [&](size_t lo, size_t hi) {
        std::vector<TT> store((size_t)mm_gen_nuts_vectors(max_depth) * D);
        mm_gstore<TT, TT *> s;
        s.base = store.data(); s.stride = 1; s.dim = D;
        for (size_t c = lo; c < hi; ++c) {
            for (int i = 0; i < D; ++i) s.st(MM_NV_X, i, (TT)positions[c * D + i]);
            mm_nuts_adapt<ST> ad;
            ad.epsilon = (ST)adapt[4 * c]; ad.epsilon_bar = (ST)adapt[4 * c + 1];
            ad.h_bar = (ST)adapt[4 * c + 2]; ad.mu = (ST)adapt[4 * c + 3];
            mm_gen_nuts_init_chain<TT, ST>(kind, P, s, &ad, eps_tol, seed, off + c);
            size_t row = 0;
            uint64_t lf = 0;
            uint32_t m = m0;
            auto rec = [&]() { if (out) { for (int i = 0; i < D; ++i) out[(c * n_collect + row) * D + i] = s.ld(MM_NV_X, i); } ++row; };
            if (write_initial) rec();
            for (unsigned t = 0; t < n_pre + n_rec; ++t) {
                ++m;
                mm_nuts_info inf = mm_gen_nuts_step<TT, ST>(kind, P, s, &ad, m, (uint32_t)n_discard, (ST)tap, max_depth, seed, off + c);
                lf += inf.n_leapfrog;
                if (t >= n_pre) rec();
            }
            for (int i = 0; i < D; ++i) positions[c * D + i] = (double)s.ld(MM_NV_X, i);
            adapt[4 * c] = (double)ad.epsilon; adapt[4 * c + 1] = (double)ad.epsilon_bar;
            adapt[4 * c + 2] = (double)ad.h_bar; adapt[4 * c + 3] = (double)ad.mu;
            if (nlf) nlf[c] = lf;
        }
    };
    if (nth <= 1 || n < 2) { work(0, n); return; }
    std::vector<std::thread> th;
    size_t nt = std::min<size_t>((size_t)nth, n);
    for (size_t t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
}

template <class TT, class ST>
static int nuts_t(int kind, int dim, const double params[8], const double *matrix, double *positions, size_t n, double tap,
                  uint64_t seed, uint64_t off, uint32_t m0, size_t nc, size_t nd, int progress, int max_depth,
                  double *adapt, void *out, uint64_t *nlf, int nth, bool force_generic = false)
{
    mm_tparams<TT> P;
    if (mm_fill_params<TT>(kind, params, &P) != 0) return -1;
    std::vector<TT> mat;
    if (kind == MM_GAUSSIAN_ND) {
        if (!matrix) return -1;
        mat.resize((size_t)dim * dim);
        for (size_t i = 0; i < mat.size(); ++i) mat[i] = (TT)matrix[i];
        P.mat = mat.data();
    }
#define NT(KIND, DD)                                                                                              \
    if (kind == KIND && dim == DD) {                                                                              \
        nuts_chains<TT, ST, mm_target<TT, KIND, DD>>(P, positions, n, tap, seed, off, m0, nc, nd, progress, max_depth, \
                                                     adapt, (TT *)out, nlf, nth);                                 \
        return 0;                                                                                                 \
    }
    if (!force_generic) {
        MM_NUTS_INSTANCES(NT)
    }
#undef NT
    if (!mm_generic_kind_ok(kind) || dim < 1 || max_depth > MM_NUTS_JMAX)
        return -2;
    nuts_chains_generic<TT, ST>(kind, dim, P, positions, n, tap, seed, off, m0, nc, nd, progress, max_depth, adapt, (TT *)out, nlf,
                                nth);
    return 0;
}

extern "C" {

int eh_nuts_run(int mode, int kind, int dim, const double params[8], const double *matrix, double *positions, size_t n,
                double target_accept_p, uint64_t seed, uint64_t chain_offset, uint32_t m0, size_t n_collect,
                size_t n_discard, int progress, int max_depth, double *adapt, void *out, uint64_t *nlf, int n_threads)
{
    if (mode == 3) {
        /* host twin of the lane-group / MFMA kernel (mm_nuts_lg.h): f64, GaussianND, grouped dot-product order */
        if (kind != MM_GAUSSIAN_ND || !matrix || (dim != 16 && dim != 32))
            return -2;
        mm_tparams<double> P;
        mm_fill_params<double>(kind, params, &P);
        P.mat = matrix;
        if (dim == 32)
            nuts_chains<double, double, mm_target_gnd_grp4<double, 32>, mm_red_grp4<double, 32>>(
                P, positions, n, target_accept_p, seed, chain_offset, m0, n_collect, n_discard, progress, max_depth, adapt,
                (double *)out, nlf, n_threads);
        else
            nuts_chains<double, double, mm_target_gnd_grp4<double, 16>, mm_red_grp4<double, 16>>(
                P, positions, n, target_accept_p, seed, chain_offset, m0, n_collect, n_discard, progress, max_depth, adapt,
                (double *)out, nlf, n_threads);
        return 0;
    }
    /* modes 4, 5, 6: type modes 0, 1, 2 through the run-time-dimension path (also the fallback of 0, 1, 2 for a
     * (kind, dim) without a fixed-D instance) */
    const bool force_generic = mode >= 4 && mode <= 6;
    if (force_generic)
        mode -= 4;
    if (mode == 0)
        return nuts_t<float, double>(kind, dim, params, matrix, positions, n, target_accept_p, seed, chain_offset, m0,
                                     n_collect, n_discard, progress, max_depth, adapt, out, nlf, n_threads, force_generic);
    if (mode == 1)
        return nuts_t<float, float>(kind, dim, params, matrix, positions, n, target_accept_p, seed, chain_offset, m0,
                                    n_collect, n_discard, progress, max_depth, adapt, out, nlf, n_threads, force_generic);
    return nuts_t<double, double>(kind, dim, params, matrix, positions, n, target_accept_p, seed, chain_offset, m0,
                                  n_collect, n_discard, progress, max_depth, adapt, out, nlf, n_threads, force_generic);
}

/* the product's f32 normal of one Philox word (mm_rng.h: mm_icdf_f32), word by word */
int eh_icdf24_words(const uint32_t *w, size_t n, float *out)
{
    const mm_icdf_global tab;
    for (size_t i = 0; i < n; ++i)
        out[i] = mm_icdf_f32(w[i], tab);
    return 0;
}

/* mm_rng.h: mm_ratio_exceeds_ln_u against the plain comparison, pair by pair: the number of disagreements */
size_t eh_ratio_filter_disagreements(const double *ratio, const double *u, size_t n)
{
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i)
        bad += mm_ratio_exceeds_ln_u(ratio[i], u[i]) != (ratio[i] > mm_log(u[i]));
    return bad;
}

/* the product's logarithm of the f32 accept uniform (mm_rng.h: mm_lnu_f32), value by value */
int eh_lnu_f32(const float *u, size_t n, float *out)
{
    const mm_icdf_global tab;
    for (size_t i = 0; i < n; ++i)
        out[i] = mm_lnu_f32(u[i], tab);
    return 0;
}

/* the NUTS leaf's acceptance statistic as the product evaluates it (mm_nuts.h: mm_accept_stat, a branch that skips the
 * exponential for d >= 0) beside the form it replaced, min(1, exp(d)): out_new, out_old [n] (tests/test_step_parity.py) */
int eh_accept_stat(int dtype, const void *d, size_t n, void *out_new, void *out_old)
{
    for (size_t i = 0; i < n; ++i) {
        if (dtype == 0) {
            const float x = ((const float *)d)[i];
            ((float *)out_new)[i] = mm_accept_stat<float>(x);
            ((float *)out_old)[i] = mm_minT(1.0f, mm_exp_hotT(x));
        } else {
            const double x = ((const double *)d)[i];
            ((double *)out_new)[i] = mm_accept_stat<double>(x);
            ((double *)out_old)[i] = mm_minT(1.0, mm_exp_hotT(x));
        }
    }
    return 0;
}

/* noise of (chain, iteration) as the engine draws it: z [n, dim], u [n] */
int eh_noise(int dtype, uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n, int dim, void *z, void *u)
{
#define NOISE_CASE(DD)                                                                                            \
    case DD:                                                                                                      \
        for (size_t c = 0; c < n; ++c) {                                                                          \
            if (dtype == 0)                                                                                       \
                mm_draw_noise<DD>(seed, chain_offset + c, iteration, (float *)z + c * DD, (float *)u + c);        \
            else                                                                                                  \
                mm_draw_noise<DD>(seed, chain_offset + c, iteration, (double *)z + c * DD, (double *)u + c);      \
        }                                                                                                         \
        return 0;
    switch (dim) {
        NOISE_CASE(1)
        NOISE_CASE(2)
        NOISE_CASE(3)
        NOISE_CASE(4)
        NOISE_CASE(5)
        NOISE_CASE(6)
        NOISE_CASE(7)
        NOISE_CASE(8)
        NOISE_CASE(16)
        NOISE_CASE(32)
    default:
        break;
    }
    if (dim < 1)
        return -2;
    for (size_t i = 0; i < n; ++i) {
        if (dtype == 0) {
            float *zz = (float *)z + i * dim;
            mm_gen_noise(seed, chain_offset + i, iteration, dim, 0.0f, [&](int k, float v) { zz[k] = v; });
            ((float *)u)[i] = mm_spare_u24(mm_block(seed, chain_offset + i, iteration, 0u));
        } else {
            double *zz = (double *)z + i * dim;
            mm_gen_noise(seed, chain_offset + i, iteration, dim, 0.0, [&](int k, double v) { zz[k] = v; });
            ((double *)u)[i] = mm_aux_u53(seed, chain_offset + i, iteration, 0);
        }
    }
    return 0;
}

/* log-density (and gradient if g != NULL) of n rows with the engine's arithmetic */
int eh_logp_grad(int dtype, int kind, int dim, const double params[8], const double *matrix, const void *x, size_t n,
                 void *logp, void *grad)
{
    /* reuse the sampler dispatch: run HMC with zero transitions is not enough, so evaluate directly */
    auto eval = [&](auto tag) -> int {
        using T = decltype(tag);
        mm_tparams<T> P;
        if (mm_fill_params<T>(kind, params, &P) != 0)
            return -1;
        std::vector<T> mat;
        if (kind == MM_GAUSSIAN_ND) {
            mat.resize((size_t)dim * dim);
            for (size_t i = 0; i < mat.size(); ++i)
                mat[i] = (T)matrix[i];
            P.mat = mat.data();
        }
        const T *xx = (const T *)x;
        T *lp = (T *)logp, *gg = (T *)grad;
#define EV(KIND, DD)                                                                                              \
    if (kind == KIND && dim == DD) {                                                                              \
        for (size_t i = 0; i < n; ++i) {                                                                          \
            T gv[DD];                                                                                             \
            lp[i] = gg ? mm_target<T, KIND, DD>::logp_grad(P, xx + i * DD, gv)                                    \
                       : mm_target<T, KIND, DD>::logp(P, xx + i * DD);                                            \
            if (gg)                                                                                               \
                for (int k = 0; k < DD; ++k)                                                                      \
                    gg[i * DD + k] = gv[k];                                                                       \
        }                                                                                                         \
        return 0;                                                                                                 \
    }
#define EV_ND(KIND) EV(KIND, 1) EV(KIND, 2) EV(KIND, 3) EV(KIND, 4) EV(KIND, 5) EV(KIND, 6) EV(KIND, 7) EV(KIND, 8) EV(KIND, 16) EV(KIND, 32)
        EV(MM_GAUSSIAN2D, 2)
        EV(MM_DIFFABLE_GAUSSIAN2D, 2)
        EV(MM_ROSENBROCK2D, 2)
        EV_ND(MM_ISOTROPIC_GAUSSIAN)
        EV_ND(MM_ROSENBROCK_ND)
        EV_ND(MM_STANDARD_NORMAL)
        EV_ND(MM_GAUSSIAN_ND)
        if (mm_generic_kind_ok(kind) && dim >= 1) {
            std::vector<T> buf((size_t)2 * dim);
            mm_gstore<T, T *> s;
            s.base = buf.data();
            s.stride = 1;
            s.dim = dim;
            for (size_t i = 0; i < n; ++i) {
                for (int k = 0; k < dim; ++k)
                    s.st(0, k, xx[i * dim + k]);
                lp[i] = mm_gen_logp_grad<T>(kind, P, s, 0, gg ? 1 : -1);
                if (gg)
                    for (int k = 0; k < dim; ++k)
                        gg[i * dim + k] = s.ld(1, k);
            }
            return 0;
        }
        return -2;
    };
    return dtype == 0 ? eval(float{}) : eval(double{});
}

} /* extern "C" */

/* ---- integer-state MH: the product's mm_discrete.h compiled for the host (the GPU must match it bit for bit) ---- */
#include "../mini_mcmc_amd/csrc/mm_discrete.h"

extern "C" int eh_discrete_run(int kind, const double *params, const int32_t *init, size_t n_chains, uint64_t seed,
                               uint64_t chain_offset, uint32_t iter0, size_t n_collect, size_t n_discard, int32_t *out,
                               int32_t *state_out, uint64_t *accept_counts)
{
    const int cap = kind == MM_BINOMIAL_CLAMP ? (int)params[0] + 1 : MM_DISCRETE_POISSON_TABLE;
    std::vector<double> tab((size_t)cap);
    mm_discrete_params P;
    P.kind = kind;
    P.n = kind == MM_BINOMIAL_CLAMP ? (int)params[0] : 0;
    P.table_len = mm_discrete_fill_table(kind, params, tab.data(), cap);
    P.logp = tab.data();
    P.ln_half = mm_log(0.5);
    for (size_t c = 0; c < n_chains; ++c) {
        int32_t x = init[c];
        uint64_t acc = 0;
        for (size_t t = 0; t < n_collect + n_discard; ++t) {
            acc += (uint64_t)mm_discrete_step(P, &x, seed, chain_offset + c, iter0 + (uint32_t)t);
            if (t >= n_discard && out)
                out[c * n_collect + (t - n_discard)] = x;
        }
        if (state_out)
            state_out[c] = x;
        if (accept_counts)
            accept_counts[c] = acc;
    }
    return 0;
}

/* ---- Gibbs with the built-in mixture conditional: the product's mm_gibbs.h compiled for the host ---- */
#include "../mini_mcmc_amd/csrc/mm_gibbs.h"

extern "C" int eh_gibbs_mixture_run(const double *params, const double *init, size_t n_chains, uint64_t seed,
                                    uint64_t chain_offset, uint32_t iter0, size_t n_collect, size_t n_discard, double *out,
                                    double *state_out)
{
    const mm_mixture_params P{params[0], params[1], params[2], params[3], params[4]};
    for (size_t c = 0; c < n_chains; ++c) {
        double s[2] = {init[2 * c], init[2 * c + 1]};
        for (size_t t = 0; t < n_collect + n_discard; ++t) {
            mm_gibbs_mixture_step(P, s, seed, chain_offset + c, iter0 + (uint32_t)t);
            if (t >= n_discard && out) {
                out[(c * n_collect + (t - n_discard)) * 2] = s[0];
                out[(c * n_collect + (t - n_discard)) * 2 + 1] = s[1];
            }
        }
        if (state_out) {
            state_out[2 * c] = s[0];
            state_out[2 * c + 1] = s[1];
        }
    }
    return 0;
}

/* ---- HMC on the dense Gaussian with the grouped reduction order: the host twin of the lane-group kernel
 *      (mm_hmc_lg.h).  state [n, D] f64 in/out; out [n, n_collect, D] or NULL; accept [n] or NULL. ---- */
template <int D>
static void hmc_grouped(const double *mat, double eps, int L, double *state, size_t n, uint64_t seed, uint64_t off,
                        uint32_t it0, size_t nc, size_t nd, double *out, uint64_t *acc)
{
    using Tgt = mm_target_gnd_grp4<double, D>;
    using Red = mm_red_grp4<double, D>;
    mm_tparams<double> P;
    P.mat = mat;
    for (size_t c = 0; c < n; ++c) {
        double x[D], g[D], p[D], u;
        for (int i = 0; i < D; ++i)
            x[i] = state[c * D + i];
        double lp = Tgt::logp_grad(P, x, g);
        uint64_t a = 0;
        for (size_t t = 0; t < nc + nd; ++t) {
            mm_draw_noise<D>(seed, off + c, it0 + (uint32_t)t, p, &u);
            a += (uint64_t)mm_hmc_step_noise<double, Tgt, 0, Red>(P, eps, L, x, &lp, g, p, mm_log(u));
            if (t >= nd && out)
                for (int i = 0; i < D; ++i)
                    out[(c * nc + (t - nd)) * D + i] = x[i];
        }
        for (int i = 0; i < D; ++i)
            state[c * D + i] = x[i];
        if (acc)
            acc[c] = a;
    }
}

/* the f32 twin (mm_hmc_lg32_kernel): blocked ownership of the coordinates, see mm_targets.h */
template <int D>
static void hmc_grouped_f32(const float *mat, float eps, int L, float *state, size_t n, uint64_t seed, uint64_t off,
                            uint32_t it0, size_t nc, size_t nd, float *out, uint64_t *acc)
{
    using Tgt = mm_target_gnd_blk4<float, D>;
    using Red = mm_red_blk4<float, D>;
    mm_tparams<float> P;
    P.mat = mat;
    for (size_t c = 0; c < n; ++c) {
        float x[D], g[D], p[D], u;
        for (int i = 0; i < D; ++i)
            x[i] = state[c * D + i];
        float lp = Tgt::logp_grad(P, x, g);
        uint64_t a = 0;
        for (size_t t = 0; t < nc + nd; ++t) {
            mm_draw_noise<D>(seed, off + c, it0 + (uint32_t)t, p, &u);
            a += (uint64_t)mm_hmc_step_noise<float, Tgt, 0, Red>(P, eps, L, x, &lp, g, p, mm_lnu_f32(u, mm_icdf_global()));
            if (t >= nd && out)
                for (int i = 0; i < D; ++i)
                    out[(c * nc + (t - nd)) * D + i] = x[i];
        }
        for (int i = 0; i < D; ++i)
            state[c * D + i] = x[i];
        if (acc)
            acc[c] = a;
    }
}

extern "C" int eh_hmc_grouped_run_f32(int dim, const float *matrix, float eps, int n_leapfrog, float *state, size_t n_chains,
                                      uint64_t seed, uint64_t chain_offset, uint32_t iter0, size_t n_collect,
                                      size_t n_discard, float *out, uint64_t *accept)
{
    if (dim == 16)
        hmc_grouped_f32<16>(matrix, eps, n_leapfrog, state, n_chains, seed, chain_offset, iter0, n_collect, n_discard, out, accept);
    else if (dim == 32)
        hmc_grouped_f32<32>(matrix, eps, n_leapfrog, state, n_chains, seed, chain_offset, iter0, n_collect, n_discard, out, accept);
    else
        return -2;
    return 0;
}

extern "C" int eh_hmc_grouped_run(int dim, const double *matrix, double eps, int n_leapfrog, double *state, size_t n_chains,
                                  uint64_t seed, uint64_t chain_offset, uint32_t iter0, size_t n_collect, size_t n_discard,
                                  double *out, uint64_t *accept)
{
    if (dim == 16)
        hmc_grouped<16>(matrix, eps, n_leapfrog, state, n_chains, seed, chain_offset, iter0, n_collect, n_discard, out, accept);
    else if (dim == 32)
        hmc_grouped<32>(matrix, eps, n_leapfrog, state, n_chains, seed, chain_offset, iter0, n_collect, n_discard, out, accept);
    else
        return -2;
    return 0;
}

/* ---- the statistics kernel's wave-level FFT (mm_stats_fft.h), its 64 lanes run one after another, phase by phase ---- */
#include "../mini_mcmc_amd/csrc/mm_stats_fft.h"

namespace {
template <int R1>
void fft_power_wave(const float *a, const float *b, int m, float *S_out)
{
    using pl = mm_fft_plan<R1>;
    std::vector<mm_cx> lds(pl::LDS_CX, mm_cx{0.f, 0.f});
    float S[64][pl::J][8];
    memset(S, 0, sizeof S);
    const double two_pi = 6.283185307179586476925286766559;
    for (int lane = 0; lane < 64; ++lane) {
        mm_cx z[pl::H], tw1[R1];
        for (int s = 0; s < pl::H; ++s) {
            const int t = 64 * s + lane;
            z[s] = t < m ? mm_cx{a[t], b[t]} : mm_cx{0.f, 0.f};
        }
        for (int r = 0; r < R1; ++r) {
            const double th = two_pi * (double)((lane * r) % pl::N) / (double)pl::N;
            tw1[r] = mm_cx{(float)std::cos(th), (float)-std::sin(th)};
        }
        mm_fft_pass1<R1>(z, [&](int r) { return tw1[r]; }, lds.data(), lane);
    }
    std::vector<mm_cx> held((size_t)64 * pl::J * 8);
    for (int lane = 0; lane < 64; ++lane)
        mm_fft_pass2_load<R1>(*reinterpret_cast<mm_cx(*)[pl::J][8]>(&held[(size_t)lane * pl::J * 8]), lds.data(), lane);
    for (int lane = 0; lane < 64; ++lane) {
        mm_cx tw2[8];
        for (int g = 0; g < 8; ++g) {
            const double th = two_pi * (double)((lane & 7) * g) / 64.0;
            tw2[g] = mm_cx{(float)std::cos(th), (float)-std::sin(th)};
        }
        mm_fft_pass2_store<R1>(*reinterpret_cast<mm_cx(*)[pl::J][8]>(&held[(size_t)lane * pl::J * 8]), [&](int g) { return tw2[g]; }, lds.data(), lane);
    }
    for (int lane = 0; lane < 64; ++lane)
        mm_fft_pass3<R1>(lds.data(), lane, S[lane]);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < pl::J; ++j)
            for (int h = 0; h < 8; ++h)
                S_out[mm_fft_bin<R1>(lane, j, h)] = S[lane][j][h];
}
} // namespace

/* |FFT_N(a + i b)|^2, N = 64 r1 bins in natural order, through the kernel's own per-lane code */
extern "C" int eh_fft_power(int r1, const float *a, const float *b, int m, float *S_out)
{
    if (m < 1 || 2 * m > 64 * r1)
        return -1;
    if (r1 == 8)
        fft_power_wave<8>(a, b, m, S_out);
    else if (r1 == 16)
        fft_power_wave<16>(a, b, m, S_out);
    else if (r1 == 32)
        fft_power_wave<32>(a, b, m, S_out);
    else
        return -1;
    return 0;
}
