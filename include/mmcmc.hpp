/*
 * mmcmc.hpp -- header-only C++17 facade over the C ABI (mmcmc.h), shaped like the reference crate's surface.
 *
 * The reference is Rust; with no Rust toolchain in this image the host side above the C ABI is C++ (this file) and
 * Python (mini_mcmc_amd/).  Names, argument meaning and error behaviour follow the crate:
 *     mini_mcmc::core::{init_det, init_with_seed}              core.rs:394-435
 *     mini_mcmc::distributions::{Gaussian2D, DiffableGaussian2D, IsotropicGaussian, Rosenbrock2D, RosenbrockND}
 *     mini_mcmc::metropolis_hastings::MetropolisHastings::{new, seed} + ChainRunner::run   (core.rs:176-186)
 *     mini_mcmc::hmc::HMC::{new, set_seed, run, step}          hmc.rs:87-158, 304-377
 *     mini_mcmc::nuts::NUTS::{new, set_seed, run}              nuts.rs:123-170, 347-353
 *     mini_mcmc::stats::{split_rhat_mean_ess, RunStats}        stats.rs:339-423
 * Samples come back as a flat row-major std::vector with shape [n_chains, n_collect, dim] (Array3 / Tensor<B,3>).
 * Rust's `Result<_, ShapeError>` / panics become the exception mmcmc::Error carrying the C status.
 */
#ifndef MMCMC_HPP
#define MMCMC_HPP

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <memory>
#include <vector>

#include "mmcmc.h"

namespace mmcmc {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &where)
        : std::runtime_error(where + ": " + mmcmc_status_string(st) + " (" + std::to_string(st) + ")"), status(st)
    {
    }
};
inline void check(int st, const char *where)
{
    if (st != MMCMC_OK)
        throw Error(st, where);
}

template <class T> constexpr int dtype_of()
{
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "f32 or f64");
    return std::is_same<T, float>::value ? MMCMC_F32 : MMCMC_F64;
}

/* core.rs:413-419 / :404-409 : row-major [n, d] */
template <class T> std::vector<T> init_with_seed(size_t n, size_t d, uint64_t seed)
{
    std::vector<double> tmp(n * d);
    check(mmcmc_init_with_seed(n, d, seed, tmp.data()), "mmcmc_init_with_seed");
    return std::vector<T>(tmp.begin(), tmp.end());
}
template <class T> std::vector<T> init_det(size_t n, size_t d) { return init_with_seed<T>(n, d, 42); }

/* ---- distributions ---- */
struct Target {
    mmcmc_target_desc d{};
    std::vector<double> matrix;
    const mmcmc_target_desc *desc()
    {
        d.matrix = matrix.empty() ? nullptr : matrix.data();
        return &d;
    }
};
inline Target Gaussian2D(std::array<double, 2> mean, std::array<std::array<double, 2>, 2> cov, bool diffable = false)
{
    Target t;
    t.d.kind = diffable ? MMCMC_DIFFABLE_GAUSSIAN2D : MMCMC_GAUSSIAN2D;
    t.d.dim = 2;
    t.d.params[0] = mean[0];
    t.d.params[1] = mean[1];
    t.d.params[2] = cov[0][0];
    t.d.params[3] = cov[0][1];
    t.d.params[4] = cov[1][0];
    t.d.params[5] = cov[1][1];
    return t;
}
inline Target DiffableGaussian2D(std::array<double, 2> mean, std::array<std::array<double, 2>, 2> cov)
{
    return Gaussian2D(mean, cov, true);
}
inline Target Rosenbrock2D(double a, double b)
{
    Target t;
    t.d.kind = MMCMC_ROSENBROCK2D;
    t.d.dim = 2;
    t.d.params[0] = a;
    t.d.params[1] = b;
    return t;
}
inline Target RosenbrockND(int dim)
{
    Target t;
    t.d.kind = MMCMC_ROSENBROCK_ND;
    t.d.dim = dim;
    return t;
}
inline Target StandardNormal(int dim)
{
    Target t;
    t.d.kind = MMCMC_STANDARD_NORMAL;
    t.d.dim = dim;
    return t;
}
inline Target GaussianND(int dim, std::vector<double> precision)
{
    Target t;
    t.d.kind = MMCMC_GAUSSIAN_ND;
    t.d.dim = dim;
    t.matrix = std::move(precision);
    return t;
}
struct IsotropicGaussian {
    double std;
    explicit IsotropicGaussian(double s) : std(s) {}
    IsotropicGaussian set_seed(uint64_t) const { return *this; } /* the engine's stream is keyed by the sampler */
    Target as_target(int dim) const
    {
        Target t;
        t.d.kind = MMCMC_ISOTROPIC_GAUSSIAN;
        t.d.dim = dim;
        t.d.params[0] = std;
        return t;
    }
};

/* ---- stats ---- */
struct RunStats {
    mmcmc_run_stats s{};
};
inline std::pair<std::vector<float>, std::vector<float>> split_rhat_mean_ess(const std::vector<float> &sample,
                                                                              size_t chains, size_t n, size_t dim,
                                                                              int device = 0)
{
    std::vector<float> rhat(dim), ess(dim);
    check(mmcmc_split_rhat_mean_ess(sample.data(), 0, MMCMC_F32, chains, n, dim, rhat.data(), ess.data(), device,
                                    nullptr),
          "mmcmc_split_rhat_mean_ess");
    return {rhat, ess};
}

/* MultiChainTracker (stats.rs:189-306): step() takes host states [n_chains, k, dim] */
class MultiChainTracker {
    mmcmc_tracker *h_ = nullptr;
    size_t n_chains_, dim_;

  public:
    MultiChainTracker(size_t n_chains, size_t n_params, int device = 0) : n_chains_(n_chains), dim_(n_params)
    {
        check(mmcmc_tracker_create(&h_, n_chains, n_params, device), "mmcmc_tracker_create");
    }
    /* own a tracker the library handed back (mmcmc_*_run_progress' tracker_out) */
    explicit MultiChainTracker(mmcmc_tracker *adopt) : h_(adopt), n_chains_(0), dim_(0)
    {
        check(mmcmc_tracker_shape(h_, &n_chains_, &dim_, nullptr), "mmcmc_tracker_shape");
    }
    MultiChainTracker(const MultiChainTracker &) = delete;
    MultiChainTracker &operator=(const MultiChainTracker &) = delete;
    ~MultiChainTracker() { mmcmc_tracker_destroy(h_); }
    template <class T> void step(const std::vector<T> &states)
    {
        const size_t k = states.size() / (n_chains_ * dim_);
        check(mmcmc_tracker_steps(h_, states.data(), 0, dtype_of<T>(), k, 0, k, nullptr), "mmcmc_tracker_steps");
    }
    std::vector<float> rhat()
    {
        std::vector<float> r(dim_);
        check(mmcmc_tracker_stats(h_, r.data(), nullptr, nullptr, nullptr), "mmcmc_tracker_stats");
        return r;
    }
    float max_rhat()
    {
        float m = 0.f;
        check(mmcmc_tracker_stats(h_, nullptr, &m, nullptr, nullptr), "mmcmc_tracker_stats");
        return m;
    }
    float p_accept()
    {
        float p = 0.f;
        check(mmcmc_tracker_stats(h_, nullptr, nullptr, &p, nullptr), "mmcmc_tracker_stats");
        return p;
    }
    /* the per-chain flavour (ChainTracker + collect_rhat, stats.rs:26-178): rhat [dim]; max_rhat / avg_p_accept may be NULL */
    std::vector<float> chain_rhat(float *max_rhat = nullptr, float *avg_p_accept = nullptr)
    {
        std::vector<float> r(dim_);
        check(mmcmc_tracker_chain_stats(h_, r.data(), max_rhat, avg_p_accept, nullptr), "mmcmc_tracker_chain_stats");
        return r;
    }
    uint64_t n() const
    {
        uint64_t v = 0;
        check(mmcmc_tracker_n(h_, &v), "mmcmc_tracker_n");
        return v;
    }
    mmcmc_tracker *handle() const { return h_; }
};

/* stats::ess_from_chainstats (stats.rs:668-671): un-split ESS of sample [chains, n, dim] with the per-chain trackers' within / var */
inline std::vector<float> ess_from_chainstats(const std::vector<float> &sample, size_t chains, size_t n, size_t dim,
                                              const MultiChainTracker &tracker, int device = 0)
{
    std::vector<float> ess(dim);
    check(mmcmc_ess_from_chainstats(sample.data(), 0, MMCMC_F32, chains, n, dim, tracker.handle(), ess.data(), device, nullptr),
          "mmcmc_ess_from_chainstats");
    return ess;
}

/* what run_progress returns: (sample [n_chains, n_collect, dim], RunStats) as the reference, plus the tracker that was fed */
template <class T> struct ProgressResult {
    std::vector<T> sample;
    RunStats stats;
    std::unique_ptr<MultiChainTracker> tracker;
};



/* ---- MetropolisHastings<T> ---- */
template <class T> class MetropolisHastings {
    mmcmc_mh *h_ = nullptr;
    size_t n_chains_, dim_;

  public:
    /* MetropolisHastings::new(target, proposal, initial_states); initial_states row-major [n_chains, dim] */
    MetropolisHastings(Target target, IsotropicGaussian proposal, const std::vector<T> &initial_states, size_t n_chains,
                       int device = 0)
        : n_chains_(n_chains), dim_(initial_states.size() / n_chains)
    {
        mmcmc_proposal_desc p{MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN, 0, proposal.std};
        check(mmcmc_mh_create(&h_, target.desc(), &p, initial_states.data(), n_chains, dtype_of<T>(), device),
              "mmcmc_mh_create");
    }
    MetropolisHastings(const MetropolisHastings &) = delete;
    MetropolisHastings &operator=(const MetropolisHastings &) = delete;
    ~MetropolisHastings() { mmcmc_mh_destroy(h_); }
    MetropolisHastings &seed(uint64_t s)
    {
        check(mmcmc_mh_seed(h_, s), "mmcmc_mh_seed");
        return *this;
    }
    /* ChainRunner::run -> [n_chains, n_collect, dim] */
    std::vector<T> run(size_t n_collect, size_t n_discard, std::vector<uint64_t> *accept_counts = nullptr)
    {
        std::vector<T> out(n_chains_ * n_collect * dim_);
        if (accept_counts)
            accept_counts->assign(n_chains_, 0);
        check(mmcmc_mh_run(h_, n_collect, n_discard, out.data(), 0, accept_counts ? accept_counts->data() : nullptr,
                           nullptr),
              "mmcmc_mh_run");
        check(mmcmc_mh_sync(h_), "mmcmc_mh_sync");
        return out;
    }
    size_t n_chains() const { return n_chains_; }
    size_t dim() const { return dim_; }
    /* ChainRunner::run_progress (core.rs:208-360): cb(user, done, total, p_accept, max_rhat) after every `every` transitions */
    ProgressResult<T> run_progress(size_t n_collect, size_t n_discard, size_t every = 0, mmcmc_progress_fn cb = nullptr,
                                   void *user = nullptr)
    {
        ProgressResult<T> r;
        r.sample.resize(n_chains_ * n_collect * dim_);
        mmcmc_tracker *t = nullptr;
        check(mmcmc_mh_run_progress(h_, n_collect, n_discard, every, cb, user, r.sample.data(), 0, &r.stats.s, &t, nullptr),
              "mmcmc_mh_run_progress");
        r.tracker.reset(new MultiChainTracker(t));
        return r;
    }
};

/* ---- HMC<T> ---- */
template <class T> class HMC {
    mmcmc_hmc *h_ = nullptr;
    size_t n_chains_, dim_;

  public:
    HMC(Target target, const std::vector<T> &initial_positions, size_t n_chains, T step_size, int n_leapfrog,
        int device = 0)
        : n_chains_(n_chains), dim_(initial_positions.size() / n_chains)
    {
        check(mmcmc_hmc_create(&h_, target.desc(), initial_positions.data(), n_chains, (double)step_size, n_leapfrog,
                               dtype_of<T>(), device),
              "mmcmc_hmc_create");
    }
    HMC(const HMC &) = delete;
    HMC &operator=(const HMC &) = delete;
    ~HMC() { mmcmc_hmc_destroy(h_); }
    HMC &set_seed(uint64_t s)
    {
        check(mmcmc_hmc_seed(h_, s), "mmcmc_hmc_seed");
        return *this;
    }
    std::vector<T> run(size_t n_collect, size_t n_discard)
    {
        std::vector<T> out(n_chains_ * n_collect * dim_);
        check(mmcmc_hmc_run(h_, n_collect, n_discard, out.data(), 0, nullptr, nullptr), "mmcmc_hmc_run");
        check(mmcmc_hmc_sync(h_), "mmcmc_hmc_sync");
        return out;
    }
    /* HMC::run_progress (hmc.rs:222-294) */
    ProgressResult<T> run_progress(size_t n_collect, size_t n_discard, size_t every = 0, mmcmc_progress_fn cb = nullptr,
                                   void *user = nullptr)
    {
        ProgressResult<T> r;
        r.sample.resize(n_chains_ * n_collect * dim_);
        mmcmc_tracker *t = nullptr;
        check(mmcmc_hmc_run_progress(h_, n_collect, n_discard, every, cb, user, r.sample.data(), 0, &r.stats.s, &t, nullptr),
              "mmcmc_hmc_run_progress");
        r.tracker.reset(new MultiChainTracker(t));
        return r;
    }
    void step()
    {
        check(mmcmc_hmc_step(h_, nullptr), "mmcmc_hmc_step");
        check(mmcmc_hmc_sync(h_), "mmcmc_hmc_sync");
    }
    std::vector<T> positions()
    {
        std::vector<T> out(n_chains_ * dim_);
        check(mmcmc_hmc_state(h_, out.data()), "mmcmc_hmc_state");
        return out;
    }
};

/* ---- a target of the user's own: the GPU analogue of `impl GradientTarget for MyDensity` (distributions.rs:65-108).
 * `hip_source` defines `template <class T> struct mmcmc_user_target` (dim, logp, logp_grad; include/mmcmc.h); it is
 * compiled at run time and the returned Target is accepted wherever a built-in one is (MH / HMC, dim <= 32). ---- */
inline Target UserTarget(const std::string &name, int dim, const std::string &hip_source, const std::vector<double> &params = {})
{
    int kind = 0;
    std::string log(1 << 16, '\0');
    const int st = mmcmc_target_register_source(name.c_str(), dim, hip_source.c_str(), &kind, &log[0], log.size());
    if (st != MMCMC_OK)
        throw Error(st, "mmcmc_target_register_source: " + std::string(log.c_str()));
    Target t;
    t.d.kind = kind;
    t.d.dim = dim;
    for (size_t i = 0; i < params.size() && i < 8; ++i)
        t.d.params[i] = params[i];
    return t;
}

/* ---- HMC over several GPUs from one call: run() executes every chain (ChainRunner::run, core.rs:176-186) ---- */
template <class T> class HMCGroup {
    mmcmc_hmc_group *g_ = nullptr;
    size_t n_chains_, dim_;

  public:
    HMCGroup(Target target, const std::vector<T> &initial_positions, size_t n_chains, T step_size, int n_leapfrog,
             const std::vector<int> &devices)
        : n_chains_(n_chains), dim_(initial_positions.size() / n_chains)
    {
        check(mmcmc_hmc_group_create(&g_, target.desc(), initial_positions.data(), n_chains, (double)step_size, n_leapfrog,
                                     dtype_of<T>(), devices.data(), (int)devices.size()),
              "mmcmc_hmc_group_create");
    }
    HMCGroup(const HMCGroup &) = delete;
    HMCGroup &operator=(const HMCGroup &) = delete;
    ~HMCGroup() { mmcmc_hmc_group_destroy(g_); }
    HMCGroup &set_seed(uint64_t s)
    {
        check(mmcmc_hmc_group_seed(g_, s), "mmcmc_hmc_group_seed");
        return *this;
    }
    std::vector<T> run(size_t n_collect, size_t n_discard)
    {
        std::vector<T> out(n_chains_ * n_collect * dim_);
        check(mmcmc_hmc_group_run(g_, n_collect, n_discard, out.data(), nullptr), "mmcmc_hmc_group_run");
        return out;
    }
    /* the same run with the sample left on the devices: only ENQUEUED on the shards' streams (returns at once; several in a
     * row keep every device busy); sync() waits, split_rhat_mean_ess() orders itself behind them */
    void run_on_devices(size_t n_collect, size_t n_discard)
    {
        check(mmcmc_hmc_group_run_async(g_, n_collect, n_discard), "mmcmc_hmc_group_run_async");
    }
    void sync() { check(mmcmc_hmc_group_sync(g_), "mmcmc_hmc_group_sync"); }
    /* how the diagnostics will travel (1 RCCL, 0 host by design, < 0 host as a fallback), known from construction on */
    int exchange() const
    {
        int status = 0;
        check(mmcmc_hmc_group_exchange(g_, &status, nullptr), "mmcmc_hmc_group_exchange");
        return status;
    }
    /* stats::split_rhat_mean_ess of the last run over the chains of all devices (RCCL inside the library) */
    std::pair<std::vector<float>, std::vector<float>> split_rhat_mean_ess()
    {
        std::vector<float> rhat(dim_), ess(dim_);
        check(mmcmc_hmc_group_split_rhat_mean_ess(g_, rhat.data(), ess.data(), nullptr), "mmcmc_hmc_group_split_rhat_mean_ess");
        return {rhat, ess};
    }
};

/* ---- NUTS: tensors f32, scalars T (mode 0 for T = double, 1 for T = float), like the reference backend ---- */
template <class T> class NUTS {
    mmcmc_nuts *h_ = nullptr;
    size_t n_chains_, dim_;

  public:
    NUTS(Target target, const std::vector<T> &initial_positions, size_t n_chains, T target_accept_p, int device = 0)
        : n_chains_(n_chains), dim_(initial_positions.size() / n_chains)
    {
        std::vector<double> init(initial_positions.begin(), initial_positions.end());
        check(mmcmc_nuts_create(&h_, target.desc(), init.data(), n_chains, (double)target_accept_p,
                                std::is_same<T, double>::value ? 0 : 1, device),
              "mmcmc_nuts_create");
    }
    NUTS(const NUTS &) = delete;
    NUTS &operator=(const NUTS &) = delete;
    ~NUTS() { mmcmc_nuts_destroy(h_); }
    NUTS &set_seed(uint64_t s)
    {
        check(mmcmc_nuts_seed(h_, s), "mmcmc_nuts_seed");
        return *this;
    }
    /* NUTS::run -> [n_chains, n_collect, dim] (f32, the backend's element type) */
    std::vector<float> run(size_t n_collect, size_t n_discard)
    {
        std::vector<float> out(n_chains_ * n_collect * dim_);
        check(mmcmc_nuts_run(h_, n_collect, n_discard, out.data(), 0, 0, nullptr), "mmcmc_nuts_run");
        check(mmcmc_nuts_sync(h_), "mmcmc_nuts_sync");
        return out;
    }
    /* NUTS::run_progress (nuts.rs:172-345): all N transitions; the per-chain trackers saw the initial position and every state */
    ProgressResult<float> run_progress(size_t n_collect, size_t n_discard, size_t every = 0, mmcmc_progress_fn cb = nullptr,
                                   void *user = nullptr)
    {
        ProgressResult<float> r;
        r.sample.resize(n_chains_ * n_collect * dim_);
        mmcmc_tracker *t = nullptr;
        check(mmcmc_nuts_run_progress(h_, n_collect, n_discard, every, cb, user, r.sample.data(), 0, &r.stats.s, &t, nullptr),
              "mmcmc_nuts_run_progress");
        r.tracker.reset(new MultiChainTracker(t));
        return r;
    }
};

/* NUTS over several GPUs from one call (mmcmc_nuts_group_*): tensors f32, scalars T, like NUTS<T> */
template <class T> class NUTSGroup {
    mmcmc_nuts_group *g_ = nullptr;
    size_t n_chains_, dim_;

  public:
    NUTSGroup(Target target, const std::vector<T> &initial_positions, size_t n_chains, T target_accept_p, const std::vector<int> &devices)
        : n_chains_(n_chains), dim_(initial_positions.size() / n_chains)
    {
        std::vector<double> init(initial_positions.begin(), initial_positions.end());
        check(mmcmc_nuts_group_create(&g_, target.desc(), init.data(), n_chains, (double)target_accept_p,
                                      std::is_same<T, double>::value ? 0 : 1, devices.data(), (int)devices.size()),
              "mmcmc_nuts_group_create");
    }
    NUTSGroup(const NUTSGroup &) = delete;
    NUTSGroup &operator=(const NUTSGroup &) = delete;
    ~NUTSGroup() { mmcmc_nuts_group_destroy(g_); }
    NUTSGroup &set_seed(uint64_t s)
    {
        check(mmcmc_nuts_group_seed(g_, s), "mmcmc_nuts_group_seed");
        return *this;
    }
    std::vector<float> run(size_t n_collect, size_t n_discard)
    {
        std::vector<float> out(n_chains_ * n_collect * dim_);
        check(mmcmc_nuts_group_run(g_, n_collect, n_discard, out.data(), 0), "mmcmc_nuts_group_run");
        return out;
    }
    std::pair<std::vector<float>, std::vector<float>> split_rhat_mean_ess()
    {
        std::vector<float> rhat(dim_), ess(dim_);
        check(mmcmc_nuts_group_split_rhat_mean_ess(g_, rhat.data(), ess.data(), nullptr), "mmcmc_nuts_group_split_rhat_mean_ess");
        return {rhat, ess};
    }
};

/* NUTS in f64 tensors / f64 scalars (mode 2 of the C ABI; BASELINE.json config 5): for mmcmc::GaussianND of dim 16
 * or 32 this runs the lane-group / MFMA kernel with tree-depth compaction (include/mmcmc.h, kernel variants). */
class NUTS64 {
    mmcmc_nuts *h_ = nullptr;
    size_t n_chains_, dim_;

  public:
    NUTS64(Target target, const std::vector<double> &initial_positions, size_t n_chains, double target_accept_p,
           int device = 0)
        : n_chains_(n_chains), dim_(initial_positions.size() / n_chains)
    {
        check(mmcmc_nuts_create(&h_, target.desc(), initial_positions.data(), n_chains, target_accept_p, 2, device),
              "mmcmc_nuts_create");
    }
    NUTS64(const NUTS64 &) = delete;
    NUTS64 &operator=(const NUTS64 &) = delete;
    ~NUTS64() { mmcmc_nuts_destroy(h_); }
    NUTS64 &set_seed(uint64_t s)
    {
        check(mmcmc_nuts_seed(h_, s), "mmcmc_nuts_seed");
        return *this;
    }
    NUTS64 &set_max_depth(int d)
    {
        check(mmcmc_nuts_set_max_depth(h_, d), "mmcmc_nuts_set_max_depth");
        return *this;
    }
    NUTS64 &set_kernel_variant(int v)
    {
        check(mmcmc_nuts_set_kernel_variant(h_, v), "mmcmc_nuts_set_kernel_variant");
        return *this;
    }
    int kernel_variant() const { return mmcmc_nuts_kernel_variant(h_); }
    std::vector<double> run(size_t n_collect, size_t n_discard)
    {
        std::vector<double> out(n_chains_ * n_collect * dim_);
        check(mmcmc_nuts_run(h_, n_collect, n_discard, out.data(), 0, 0, nullptr), "mmcmc_nuts_run");
        check(mmcmc_nuts_sync(h_), "mmcmc_nuts_sync");
        return out;
    }
    /* NUTS::run_progress (nuts.rs:172-345): all N transitions; the per-chain trackers saw the initial position and every state */
    ProgressResult<double> run_progress(size_t n_collect, size_t n_discard, size_t every = 0, mmcmc_progress_fn cb = nullptr,
                                   void *user = nullptr)
    {
        ProgressResult<double> r;
        r.sample.resize(n_chains_ * n_collect * dim_);
        mmcmc_tracker *t = nullptr;
        check(mmcmc_nuts_run_progress(h_, n_collect, n_discard, every, cb, user, r.sample.data(), 0, &r.stats.s, &t, nullptr),
              "mmcmc_nuts_run_progress");
        r.tracker.reset(new MultiChainTracker(t));
        return r;
    }
};

/* MetropolisHastings<i32, f64, ...> over the reference's discrete test models (include/mmcmc.h: MMCMC_POISSON_REFLECT,
 * MMCMC_BINOMIAL_CLAMP, MMCMC_POISSON_NONNEG); run() -> [n_chains, n_collect] */
class DiscreteMetropolisHastings {
    mmcmc_mh_discrete *h_ = nullptr;
    size_t n_chains_;

  public:
    DiscreteMetropolisHastings(int kind, const std::vector<double> &params, const std::vector<int32_t> &initial_states,
                               int device = 0)
        : n_chains_(initial_states.size())
    {
        check(mmcmc_mh_discrete_create(&h_, kind, params.data(), initial_states.data(), n_chains_, device),
              "mmcmc_mh_discrete_create");
    }
    DiscreteMetropolisHastings(const DiscreteMetropolisHastings &) = delete;
    DiscreteMetropolisHastings &operator=(const DiscreteMetropolisHastings &) = delete;
    ~DiscreteMetropolisHastings() { mmcmc_mh_discrete_destroy(h_); }
    DiscreteMetropolisHastings &seed(uint64_t s)
    {
        check(mmcmc_mh_discrete_seed(h_, s), "mmcmc_mh_discrete_seed");
        return *this;
    }
    std::vector<int32_t> run(size_t n_collect, size_t n_discard)
    {
        std::vector<int32_t> out(n_chains_ * n_collect);
        check(mmcmc_mh_discrete_run(h_, n_collect, n_discard, out.data(), 0, nullptr), "mmcmc_mh_discrete_run");
        return out;
    }
    std::vector<uint64_t> accept_counts()
    {
        std::vector<uint64_t> a(n_chains_);
        check(mmcmc_mh_discrete_accept_counts(h_, a.data()), "mmcmc_mh_discrete_accept_counts");
        return a;
    }
};

/* GibbsSampler<f64, MixtureConditional> (gibbs.rs:139-203, 231-285); run() -> [n_chains, n_collect, 2] */
class GibbsMixtureSampler {
    mmcmc_gibbs_mixture *h_ = nullptr;
    size_t n_chains_;

  public:
    GibbsMixtureSampler(double mu0, double sigma0, double mu1, double sigma1, double pi0,
                        const std::vector<double> &initial_states, int device = 0)
        : n_chains_(initial_states.size() / 2)
    {
        const double p[5] = {mu0, sigma0, mu1, sigma1, pi0};
        check(mmcmc_gibbs_mixture_create(&h_, p, initial_states.data(), n_chains_, device), "mmcmc_gibbs_mixture_create");
    }
    GibbsMixtureSampler(const GibbsMixtureSampler &) = delete;
    GibbsMixtureSampler &operator=(const GibbsMixtureSampler &) = delete;
    ~GibbsMixtureSampler() { mmcmc_gibbs_mixture_destroy(h_); }
    GibbsMixtureSampler &set_seed(uint64_t s)
    {
        check(mmcmc_gibbs_mixture_seed(h_, s), "mmcmc_gibbs_mixture_seed");
        return *this;
    }
    std::vector<double> run(size_t n_collect, size_t n_discard)
    {
        std::vector<double> out(n_chains_ * n_collect * 2);
        check(mmcmc_gibbs_mixture_run(h_, n_collect, n_discard, out.data(), 0, nullptr), "mmcmc_gibbs_mixture_run");
        return out;
    }
};

} // namespace mmcmc
#endif /* MMCMC_HPP */
