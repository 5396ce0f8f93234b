/* run_progress behind the C ABI: the reference's three progress runners, their tracker feeding orders written ONCE here
 * (every binding -- Python, C++, Rust -- calls these instead of re-deriving the sequences):
 *
 *   HMC   (hmc.rs:222-294)             burn-in with no tracker; ONE MultiChainTracker (stats.rs:189-306) fed the positions
 *                                      the sampling starts from, then every collected state; RunStats from the sample.
 *   MH    (core.rs:90-140, 208-360)    one ChainTracker per chain (stats.rs:26-141), constructed with the chain's state
 *                                      before the first step and stepped with EVERY state, burn-in included; the bar shows
 *                                      the average p_accept and collect_rhat's maximum (stats.rs:150-178).
 *   NUTS  (nuts.rs:172-345, 473-526)   as MH: per-chain trackers from the initial position through all
 *                                      n_discard + n_collect states; init_chain runs once, so the whole run is ONE launch
 *                                      (mmcmc_nuts_run with every state recorded) and the trackers are fed afterwards.
 *
 * Instead of a terminal bar redrawn by a UI thread, `cb(user, transitions done, transitions total, p_accept, max_rhat)` is
 * called after every `every` transitions (0 = ten times per run) with the two numbers the reference's bar shows.
 * Host-only code over the public C ABI (mmcmc_*_run_rows, mmcmc_tracker_*, mmcmc_run_stats_from): no kernels here. */
#include "../../include/mmcmc.h"
#include "mm_hostcopy.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace {

struct DevGuard {
    int prev = 0;
    explicit DevGuard(int d)
    {
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(d);
    }
    ~DevGuard() { (void)hipSetDevice(prev); }
};

struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p)
            (void)hipFree(p);
    }
};

/* a caller that passes no stream gets one of the call's own: the samplers' NULL means "the handle's stream", the
 * trackers' and the diagnostics' NULL the device's default stream -- one explicit stream orders all of them */
struct StreamOwner {
    hipStream_t s = nullptr;
    bool own = false;
    int init(void *given)
    {
        if (given) {
            s = (hipStream_t)given;
            return MMCMC_OK;
        }
        hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        own = e == hipSuccess;
        return e == hipSuccess ? MMCMC_OK : (int)e;
    }
    ~StreamOwner()
    {
        if (own) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
};

struct TrackerOwner {
    mmcmc_tracker *t = nullptr;
    ~TrackerOwner()
    {
        if (t)
            (void)mmcmc_tracker_destroy(t);
    }
};

#define MMP(x)                                                                                                      \
    do {                                                                                                            \
        int _s = (x);                                                                                               \
        if (_s != MMCMC_OK)                                                                                         \
            return _s;                                                                                              \
    } while (0)
#define MMP_HIP(x)                                                                                                  \
    do {                                                                                                            \
        hipError_t _e = (x);                                                                                        \
        if (_e != hipSuccess)                                                                                       \
            return (int)_e;                                                                                         \
    } while (0)

/* what differs between the fixed-length samplers */
struct FixedOps {
    int (*run_rows)(void *h, size_t n_rows, size_t n_discard, void *out_device, size_t n_total_rows, size_t row0, void *stream);
    int (*state)(void *h, void *out_host);
    bool per_chain_trackers; /* MH: ChainTrackers through the burn-in; HMC: one MultiChainTracker after it */
};

int report(mmcmc_tracker *tr, bool per_chain, mmcmc_progress_fn cb, void *user, uint64_t done, uint64_t total, void *stream)
{
    if (!cb)
        return MMCMC_OK;
    uint64_t n = 0;
    MMP(mmcmc_tracker_n(tr, &n));
    if (n < 2)
        return MMCMC_OK; /* n / (n - 1): nothing to show yet, like the reference's `valid.len() >= 2` guard */
    float mx = 0.f, p = 0.f;
    if (per_chain)
        MMP(mmcmc_tracker_chain_stats(tr, nullptr, &mx, &p, stream));
    else
        MMP(mmcmc_tracker_stats(tr, nullptr, &mx, &p, stream));
    cb(user, done, total, p, mx);
    return MMCMC_OK;
}

int finish(const void *d_out, void *out, int out_is_device, int dtype, size_t n_chains, size_t n_collect, size_t dim,
           mmcmc_run_stats *stats, int device, void *stream, TrackerOwner &tr, mmcmc_tracker **tracker_out)
{
    if (stats && n_collect >= 2)
        MMP(mmcmc_run_stats_from(d_out, 1, dtype, n_chains, n_collect, dim, stats, device, stream));
    if (out && !out_is_device && n_collect)
        MMP_HIP(mm_copy_to_host(out, d_out, n_chains * n_collect * dim * (dtype == MMCMC_F32 ? 4 : 8), device, (hipStream_t)stream));
    MMP_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (tracker_out) {
        *tracker_out = tr.t;
        tr.t = nullptr;
    }
    return MMCMC_OK;
}

int fixed_run_progress(void *h, const FixedOps &ops, size_t n_chains, int dim_i, int dtype, int device, size_t n_collect,
                       size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user, void *out, int out_is_device,
                       mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream)
{
    if (tracker_out)
        *tracker_out = nullptr;
    if (n_collect == 0 || (stats && n_collect < 2))
        return MMCMC_ERR_INVALID_ARG;
    const size_t dim = (size_t)dim_i, esz = dtype == MMCMC_F32 ? 4 : 8, total = n_collect + n_discard;
    DevGuard g(device);
    StreamOwner so;
    MMP(so.init(stream));
    stream = (void *)so.s;
    TrackerOwner tr;
    MMP(mmcmc_tracker_create(&tr.t, n_chains, dim, device));
    std::vector<unsigned char> st0(n_chains * dim * esz);
    if (every == 0)
        every = std::max<size_t>(1, (n_collect + 9) / 10);
    DevBuf sample, burn;
    void *d_out = out;
    if (!out || !out_is_device) {
        MMP_HIP(hipMalloc(&sample.p, n_chains * n_collect * dim * esz));
        d_out = sample.p;
    }
    if (ops.per_chain_trackers) {
        /* ChainTracker::new(n_params, chain.current_state()) (core.rs:103), then every state of the burn-in too */
        MMP(ops.state(h, st0.data()));
        MMP(mmcmc_tracker_init_last(tr.t, st0.data(), 0, dtype, stream));
        if (n_discard) {
            const size_t chunk = std::min(every, n_discard);
            MMP_HIP(hipMalloc(&burn.p, n_chains * chunk * dim * esz));
            for (size_t done = 0; done < n_discard;) {
                const size_t k = std::min(chunk, n_discard - done);
                MMP(ops.run_rows(h, k, 0, burn.p, k, 0, stream));
                MMP(mmcmc_tracker_steps(tr.t, burn.p, 1, dtype, k, 0, k, stream));
                done += k;
                MMP(report(tr.t, true, cb, user, done, total, stream));
            }
        }
    } else {
        /* (0..n_discard).for_each(|_| self.step()); tracker.step(positions) (hmc.rs:229-247) */
        if (n_discard)
            MMP(ops.run_rows(h, 0, n_discard, nullptr, 0, 0, stream));
        MMP(ops.state(h, st0.data()));
        MMP(mmcmc_tracker_steps(tr.t, st0.data(), 0, dtype, 1, 0, 1, stream));
    }
    for (size_t done = 0; done < n_collect;) {
        const size_t k = std::min(every, n_collect - done);
        MMP(ops.run_rows(h, k, 0, d_out, n_collect, done, stream));
        MMP(mmcmc_tracker_steps(tr.t, d_out, 1, dtype, n_collect, done, k, stream));
        done += k;
        MMP(report(tr.t, ops.per_chain_trackers, cb, user, n_discard + done, total, stream));
    }
    return finish(d_out, out, out_is_device, dtype, n_chains, n_collect, dim, stats, device, stream, tr, tracker_out);
}

int mh_rows(void *h, size_t a, size_t b, void *o, size_t c, size_t d, void *s) { return mmcmc_mh_run_rows((mmcmc_mh *)h, a, b, o, c, d, s); }
int mh_state(void *h, void *o) { return mmcmc_mh_state((mmcmc_mh *)h, o); }
int hmc_rows(void *h, size_t a, size_t b, void *o, size_t c, size_t d, void *s) { return mmcmc_hmc_run_rows((mmcmc_hmc *)h, a, b, o, c, d, s); }
int hmc_state(void *h, void *o) { return mmcmc_hmc_state((mmcmc_hmc *)h, o); }

} // namespace

extern "C" {

int mmcmc_mh_run_progress(mmcmc_mh *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                          void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream)
{
    size_t c = 0;
    int dim = 0, dtype = 0, device = 0;
    MMP(mmcmc_mh_shape(h, &c, &dim, &dtype, &device));
    const FixedOps ops{mh_rows, mh_state, true};
    return fixed_run_progress(h, ops, c, dim, dtype, device, n_collect, n_discard, every, cb, user, out, out_is_device, stats,
                              tracker_out, stream);
}

int mmcmc_hmc_run_progress(mmcmc_hmc *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                           void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream)
{
    size_t c = 0;
    int dim = 0, dtype = 0, device = 0;
    MMP(mmcmc_hmc_shape(h, &c, &dim, &dtype, &device));
    const FixedOps ops{hmc_rows, hmc_state, false};
    return fixed_run_progress(h, ops, c, dim, dtype, device, n_collect, n_discard, every, cb, user, out, out_is_device, stats,
                              tracker_out, stream);
}

int mmcmc_nuts_run_progress(mmcmc_nuts *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                            void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream)
{
    if (tracker_out)
        *tracker_out = nullptr;
    size_t n_chains = 0;
    int dim_i = 0, mode = 0, device = 0;
    MMP(mmcmc_nuts_shape(h, &n_chains, &dim_i, &mode, &device));
    if (n_collect == 0 || (stats && n_collect < 2))
        return MMCMC_ERR_INVALID_ARG;
    const int dtype = mode == 2 ? MMCMC_F64 : MMCMC_F32;
    const size_t dim = (size_t)dim_i, esz = dtype == MMCMC_F32 ? 4 : 8, total = n_collect + n_discard;
    DevGuard g(device);
    StreamOwner so;
    MMP(so.init(stream));
    stream = (void *)so.s;
    TrackerOwner tr;
    MMP(mmcmc_tracker_create(&tr.t, n_chains, dim, device));
    /* pos_0 = the position the chain starts from (init_chain does not move it, nuts.rs:478-486) */
    std::vector<unsigned char> st0(n_chains * dim * esz);
    MMP(mmcmc_nuts_state(h, st0.data()));
    MMP(mmcmc_tracker_init_last(tr.t, st0.data(), 0, dtype, stream));
    /* all n_discard + n_collect states of every chain, then the trackers in the order the chains produced them */
    DevBuf all, sample;
    MMP_HIP(hipMalloc(&all.p, n_chains * total * dim * esz));
    MMP(mmcmc_nuts_run(h, n_collect, n_discard, all.p, 1, 2, stream));
    if (every == 0)
        every = std::max<size_t>(1, (total + 9) / 10);
    for (size_t done = 0; done < total;) {
        const size_t k = std::min(every, total - done);
        MMP(mmcmc_tracker_steps(tr.t, all.p, 1, dtype, total, done, k, stream));
        done += k;
        MMP(report(tr.t, true, cb, user, done, total, stream));
    }
    void *d_out = out;
    if (!out || !out_is_device) {
        MMP_HIP(hipMalloc(&sample.p, n_chains * n_collect * dim * esz));
        d_out = sample.p;
    }
    /* the sample = the last n_collect rows of every chain (nuts.rs:516-521) */
    MMP_HIP(hipMemcpy2DAsync(d_out, n_collect * dim * esz, (const char *)all.p + n_discard * dim * esz, total * dim * esz,
                             n_collect * dim * esz, n_chains, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return finish(d_out, out, out_is_device, dtype, n_chains, n_collect, dim, stats, device, stream, tr, tracker_out);
}

} /* extern "C" */
