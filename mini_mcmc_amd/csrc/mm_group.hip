/*
 * mm_group.hip -- device groups behind the C ABI: ONE call runs every chain on N GPUs of the process.
 *
 * The reference's contract is "run() executes all chains" (ChainRunner::run core.rs:176-186, HMC::run hmc.rs:137-158);
 * with one sampler handle per device the caller of the C ABI would have to write the sharding and the exchange of the
 * diagnostics itself.  A group owns one handle per device -- device i holds the contiguous block of global chains
 * [first_i, first_i + n_i), keyed into the random stream by their GLOBAL index, so the samples do not depend on the
 * number of devices.  `mmcmc_*_group_run` BLOCKS (whatever its arguments).  The asynchronous spelling is its own entry
 * point, `mmcmc_{mh,hmc}_group_run_async`: the launches are ENQUEUED on every shard's stream by the calling thread and
 * the call returns at once -- like the single-device call on a caller's stream, so back-to-back runs keep every device
 * busy without a host round trip per run (mmcmc_*_group_sync waits; everything that reads results is ordered behind the
 * queued work by the shard's stream).  Calls that block (group_run, NUTS with its adaptation hand-offs, the
 * diagnostics) run from one host thread per device, each blocking only its own thread.  A run that fails after some
 * shards have advanced and others have not marks the group broken (MMCMC_ERR_GROUP_BROKEN from then on).  Chains never talk to each other while sampling: no data-path collective.  The one exchange is
 * the split-R-hat / ESS reduction (stats.rs:416-546): every device reduces its own sample to the per-half-chain
 * statistics (mmcmc_stats_partials), RCCL all-gathers means / sums of squares over xGMI (ncclAllGather) and
 * all-reduces the lag sums (ncclAllReduce), and the host finish (mmcmc_stats_finish) runs on the gathered statistics
 * in the single-GPU order.  RCCL is loaded at run time (librccl.so.1: the copy PyTorch has already mapped, if any);
 * without it, or when a device appears twice in the list (several shards on one GPU: what a one-GPU test box can
 * exercise), the few KB of statistics are exchanged through the host instead.
 */
#include "../../include/mmcmc.h"
#include "mm_hostcopy.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

/* ---- RCCL, bound at run time ---- */
typedef void *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && AllReduce; }
};
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
                break;
        }
        if (!r.lib)
            return;
        r.CommInitAll = (int (*)(ncclComm_t *, int, const int *))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))dlsym(r.lib, "ncclAllGather");
        r.AllReduce = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    });
    return r;
}

/* withinvar's cross-chain sums (stats.rs:449-465) over the GATHERED per-half-chain statistics, on the device (round 5: the
 * host used to copy all N x 2 x part floats back and walk them -- 1 ms at one device, and growing with the device count):
 * block (x = 4 d + quarter, y = rank i) sums a quarter of rank i's 2 n_i half-chains of parameter d in f64 -- sum of the means,
 * of their squares, of the biased variances ssq / m -- into out[((i D + d) 4 + quarter) 3 ..]; the host adds the N x 4 partial
 * sums of a parameter in index order and forms sum((mean - overall)^2) = Sq - Sd^2 / c2.  gathered: [N][means (part) | ssq (part)],
 * a rank's valid entries first (2 n_i D), zero padding behind. */
struct GroupCounts {
    unsigned int n2[64]; /* 2 n_i */
};
__global__ __launch_bounds__(256) void mm_group_cross_sums_kernel(const float *__restrict__ gathered, size_t part, unsigned int D,
                                                                  float nf, GroupCounts cnt, double *__restrict__ out)
{
    __shared__ double red[3][256];
    const unsigned int d = blockIdx.x >> 2, quarter = blockIdx.x & 3u, i = blockIdx.y, tid = threadIdx.x;
    const float *mi = gathered + (size_t)i * 2 * part, *qi = mi + part;
    const unsigned int c2 = cnt.n2[i], per = (c2 + 3u) / 4u, lo = quarter * per, hi = lo + per < c2 ? lo + per : c2;
    /* the means are taken relative to the mean of the FIRST half of global chain 0 -- rank 0's first entry, the very shift
     * mm_stats_tail_kernel uses on one device (means[d] of the splitcat order) -- and subtracted in f32 as there, so that
     * sum((mean - overall)^2) = Sq - Sd^2 / c2 loses no digits on a target far from the origin */
    const float shift = gathered[d];
    double sd = 0.0, sq = 0.0, ws = 0.0;
    for (unsigned int c = lo + tid; c < hi; c += 256u) {
        const double mu = (double)(mi[(size_t)c * D + d] - shift);
        sd += mu;
        sq += mu * mu;
        ws += (double)(qi[(size_t)c * D + d] / nf); /* biased per-chain variance (quirk Q8) */
    }
    red[0][tid] = sd;
    red[1][tid] = sq;
    red[2][tid] = ws;
    __syncthreads();
    for (unsigned int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            red[0][tid] += red[0][tid + st];
            red[1][tid] += red[1][tid + st];
            red[2][tid] += red[2][tid + st];
        }
        __syncthreads();
    }
    if (tid < 3)
        out[(((size_t)i * D + d) * 4 + quarter) * 3 + tid] = red[tid][0];
}

/* one host thread per device for the life of the group: calls into the per-device ABI block only their own thread,
 * and per-thread state of that ABI (the statistics' work buffers) stays with its device */
class Worker {
  public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job);
            has_job_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return done_; });
        return rc_;
    }

  private:
    void loop()
    {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return has_job_ || quit_; });
                if (quit_ && !has_job_)
                    return;
                job = std::move(job_);
                has_job_ = false;
            }
            const int rc = job();
            {
                std::lock_guard<std::mutex> l(m_);
                rc_ = rc;
                done_ = true;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool has_job_ = false, done_ = true, quit_ = false;
    int rc_ = 0;
    std::thread th_;
};

struct Shard {
    int device = 0;
    size_t first = 0, n = 0; /* global chains [first, first + n) */
    mmcmc_hmc *hmc = nullptr;
    mmcmc_mh *mh = nullptr;
    mmcmc_nuts *nuts = nullptr;
    void *d_sample = nullptr; /* [n, n_collect, dim] of the last run, on the device */
    size_t sample_cap = 0;
    hipStream_t stream = nullptr;
    float *d_stats = nullptr; /* means [2 cmax, dim] | ssq [2 cmax, dim] | acov [m, dim] | gathered [N][2][2 cmax dim] */
    size_t stats_cap = 0;
    ncclComm_t comm = nullptr;
    hipEvent_t tev0 = nullptr, tev1 = nullptr; /* the stream timer (group_stream_timer) */
    std::unique_ptr<Worker> worker;
};

struct Group {
    int sampler = 0; /* 0 MH, 1 HMC, 2 NUTS */
    int progress = 0; /* NUTS: run (N - 1 transitions) or run_progress stepping (nuts.rs:457-471 / 491-522) */
    int dtype = MMCMC_F32, dim = 0;
    size_t n_chains = 0, cmax = 0;
    size_t last_collect = 0;
    bool use_rccl = false, comm_ready = false;
    /* why the statistics travel as they do (reported through *used_rccl): 1 RCCL; 0 the host BY DESIGN (a device listed
     * twice: RCCL refuses two ranks on one GPU); -1 the host as a FALLBACK: libnccl / librccl not found; -2: ncclCommInitAll
     * failed (e.g. no peer access) -- a scaling run must not take the host path silently */
    int exchange_status = 0;
    /* an asynchronous run whose enqueue loop failed half way leaves the shards at different iterations: the group refuses
     * every later run / diagnostics call with MMCMC_ERR_GROUP_BROKEN instead of continuing on an inconsistent state */
    bool broken = false;
    double phase_ms[3] = {0.0, 0.0, 0.0}; /* last group_split_rhat_ess: local partials | exchange | finish (host wall clock) */
    uint64_t user_offset = 0;
    std::vector<Shard> sh;
    size_t esize() const { return dtype == MMCMC_F32 ? 4 : 8; }
};

template <class F> int for_each_shard(Group *g, F &&f)
{
    for (size_t i = 0; i < g->sh.size(); ++i) {
        if (!g->sh[i].worker)
            g->sh[i].worker.reset(new Worker());
        g->sh[i].worker->post([&f, g, i]() -> int { return f(g->sh[i], (int)i); });
    }
    int st = MMCMC_OK;
    for (size_t i = 0; i < g->sh.size(); ++i) {
        const int rc = g->sh[i].worker->wait();
        if (rc != MMCMC_OK && st == MMCMC_OK)
            st = rc;
    }
    return st;
}

int group_destroy(Group *g)
{
    if (!g)
        return MMCMC_ERR_INVALID_ARG;
    /* runs may still be queued on the shards' streams (the asynchronous group_run): nothing they use may be freed under them */
    for (Shard &s : g->sh)
        if (s.stream && hipSetDevice(s.device) == hipSuccess)
            (void)hipStreamSynchronize(s.stream);
    for (Shard &s : g->sh) {
        s.worker.reset();
        (void)hipSetDevice(s.device);
        if (s.comm && rccl().ok())
            (void)rccl().CommDestroy(s.comm);
        if (s.hmc)
            (void)mmcmc_hmc_destroy(s.hmc);
        if (s.mh)
            (void)mmcmc_mh_destroy(s.mh);
        if (s.nuts)
            (void)mmcmc_nuts_destroy(s.nuts);
        if (s.d_sample)
            (void)hipFree(s.d_sample);
        if (s.d_stats)
            (void)hipFree(s.d_stats);
        if (s.tev0)
            (void)hipEventDestroy(s.tev0);
        if (s.tev1)
            (void)hipEventDestroy(s.tev1);
        if (s.stream)
            (void)hipStreamDestroy(s.stream);
    }
    delete g;
    return MMCMC_OK;
}

int group_create(Group **out, int sampler, const mmcmc_target_desc *target, const mmcmc_proposal_desc *proposal, const void *init,
                 size_t n_chains, double scale, int n_leapfrog, int dtype, const int *devices, int n_devices)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!target || !init || !devices || n_devices < 1 || n_devices > 64 || n_chains < (size_t)n_devices ||
        (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return MMCMC_ERR_NO_DEVICE;
    bool dup = false;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= n_dev)
            return MMCMC_ERR_INVALID_ARG;
        for (int j = 0; j < i; ++j)
            dup = dup || devices[j] == devices[i];
    }
    Group *g = new (std::nothrow) Group();
    if (!g)
        return (int)hipErrorOutOfMemory;
    g->sampler = sampler;
    g->dtype = dtype;
    g->dim = target->dim;
    g->n_chains = n_chains;
    g->use_rccl = !dup && rccl().ok();
    g->exchange_status = g->use_rccl ? 1 : dup ? 0 : -1;
    g->sh.resize((size_t)n_devices);
    const size_t esz = g->esize(), base = n_chains / (size_t)n_devices, rem = n_chains % (size_t)n_devices;
    size_t first = 0;
    for (int i = 0; i < n_devices; ++i) {
        Shard &s = g->sh[(size_t)i];
        s.device = devices[i];
        s.first = first;
        s.n = base + ((size_t)i < rem ? 1 : 0);
        first += s.n;
        g->cmax = s.n > g->cmax ? s.n : g->cmax;
    }
    const int st = for_each_shard(g, [&](Shard &s, int) -> int {
        /* NUTS: init is [n_chains, dim] doubles whatever the mode (Vec<Vec<T>>, nuts.rs:123-129); scale = target_accept_p,
         * n_leapfrog = the type mode */
        const char *p = (const char *)init + s.first * (size_t)g->dim * (sampler == 2 ? sizeof(double) : esz);
        int rc = sampler == 2   ? mmcmc_nuts_create(&s.nuts, target, (const double *)p, s.n, scale, n_leapfrog, s.device)
                 : sampler == 1 ? mmcmc_hmc_create(&s.hmc, target, p, s.n, scale, n_leapfrog, dtype, s.device)
                                : mmcmc_mh_create(&s.mh, target, proposal, p, s.n, dtype, s.device);
        if (rc != MMCMC_OK)
            return rc;
        rc = sampler == 2 ? mmcmc_nuts_set_chain_offset(s.nuts, s.first)
                          : sampler ? mmcmc_hmc_set_chain_offset(s.hmc, s.first) : mmcmc_mh_set_chain_offset(s.mh, s.first);
        if (rc != MMCMC_OK)
            return rc;
        /* nobody reads the per-handle launch timing of a shard: no event packets between back-to-back launches */
        if (sampler == 1)
            (void)mmcmc_hmc_enable_timing(s.hmc, 0);
        else if (sampler == 0)
            (void)mmcmc_mh_enable_timing(s.mh, 0);
        if (hipSetDevice(s.device) != hipSuccess)
            return MMCMC_ERR_NO_DEVICE;
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        return e == hipSuccess ? MMCMC_OK : (int)e;
    });
    if (st != MMCMC_OK) {
        group_destroy(g);
        return st;
    }
    /* the communicators are made HERE, not inside the first diagnostics call: a timed reduction never pays
     * ncclCommInitAll, and a failure is known (mmcmc_*_group_exchange) before any sampling has been paid for */
    if (g->use_rccl) {
        const size_t N = g->sh.size();
        std::vector<ncclComm_t> comms(N);
        std::vector<int> devs(N);
        for (size_t i = 0; i < N; ++i)
            devs[i] = g->sh[i].device;
        if (rccl().CommInitAll(comms.data(), (int)N, devs.data()) != 0) {
            g->use_rccl = false; /* e.g. no peer access: the statistics travel through the host, and the status says so */
            g->exchange_status = -2;
        } else {
            for (size_t i = 0; i < N; ++i)
                g->sh[i].comm = comms[i];
            g->comm_ready = true;
        }
    }
    *out = g;
    return MMCMC_OK;
}

int group_seed(Group *g, uint64_t seed)
{
    if (!g)
        return MMCMC_ERR_INVALID_ARG;
    for (Shard &s : g->sh) {
        const int rc = g->sampler == 2 ? mmcmc_nuts_seed(s.nuts, seed) : g->sampler ? mmcmc_hmc_seed(s.hmc, seed) : mmcmc_mh_seed(s.mh, seed);
        if (rc != MMCMC_OK)
            return rc;
    }
    return MMCMC_OK;
}

int group_set_chain_offset(Group *g, uint64_t off)
{
    if (!g)
        return MMCMC_ERR_INVALID_ARG;
    g->user_offset = off;
    for (Shard &s : g->sh) {
        const int rc = g->sampler == 2 ? mmcmc_nuts_set_chain_offset(s.nuts, off + s.first)
                       : g->sampler    ? mmcmc_hmc_set_chain_offset(s.hmc, off + s.first)
                                       : mmcmc_mh_set_chain_offset(s.mh, off + s.first);
        if (rc != MMCMC_OK)
            return rc;
    }
    return MMCMC_OK;
}

static int shard_reserve_sample(Shard &s, size_t bytes)
{
    if (bytes <= s.sample_cap)
        return MMCMC_OK;
    /* work queued on the shard's stream may still write the old buffer */
    hipError_t e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess)
        return (int)e;
    if (s.d_sample)
        (void)hipFree(s.d_sample);
    s.d_sample = nullptr;
    s.sample_cap = 0;
    if ((e = hipMalloc(&s.d_sample, bytes)) != hipSuccess)
        return (int)e;
    s.sample_cap = bytes;
    return MMCMC_OK;
}

/* run(n_collect, n_discard) of every chain; the sample stays on the devices (one shard each) and, when out_host is
 * given, is also copied into the caller's [n_chains, n_collect, dim] array; accept_counts [n_chains] or NULL.  BLOCKING:
 * returns when every shard's results are there (also with out_host == NULL and accept_counts == NULL: a caller may read
 * mmcmc_*_group_shard's device pointer from any stream afterwards).  async (mmcmc_{mh,hmc}_group_run_async only):
 * enqueued on every shard's stream from this thread, returns without waiting (group_sync). */
int group_run(Group *g, size_t n_collect, size_t n_discard, void *out_host, uint64_t *accept_counts, bool async)
{
    if (!g)
        return MMCMC_ERR_INVALID_ARG;
    if (g->broken)
        return MMCMC_ERR_GROUP_BROKEN;
    const size_t esz = g->esize(), row = n_collect * (size_t)g->dim * esz;
    if (async) {
        if (g->sampler == 2)
            return MMCMC_ERR_UNSUPPORTED; /* NUTS hands adaptation state over between its launches: blocking only */
        int prev = -1;
        (void)hipGetDevice(&prev);
        int st = MMCMC_OK;
        /* every allocation first: a failure here has advanced no shard */
        for (Shard &s : g->sh) {
            if (hipSetDevice(s.device) != hipSuccess) {
                st = MMCMC_ERR_NO_DEVICE;
                break;
            }
            if ((st = shard_reserve_sample(s, s.n * row)) != MMCMC_OK)
                break;
        }
        size_t enqueued = 0;
        if (st == MMCMC_OK)
            for (Shard &s : g->sh) {
                if (hipSetDevice(s.device) != hipSuccess) {
                    st = MMCMC_ERR_NO_DEVICE;
                    break;
                }
                void *d_out = n_collect ? s.d_sample : nullptr;
                st = g->sampler ? mmcmc_hmc_run(s.hmc, n_collect, n_discard, d_out, 1, nullptr, s.stream)
                                : mmcmc_mh_run(s.mh, n_collect, n_discard, d_out, 1, nullptr, s.stream);
                if (st != MMCMC_OK)
                    break;
                ++enqueued;
            }
        if (st != MMCMC_OK && enqueued > 0) {
            /* shards 0 .. enqueued - 1 have advanced, the rest have not: wait for what was queued and refuse further use */
            for (size_t i = 0; i < enqueued; ++i)
                if (hipSetDevice(g->sh[i].device) == hipSuccess)
                    (void)hipStreamSynchronize(g->sh[i].stream);
            g->broken = true;
        }
        if (prev >= 0)
            (void)hipSetDevice(prev);
        if (st == MMCMC_OK)
            g->last_collect = n_collect;
        return st;
    }
    std::vector<char> advanced(g->sh.size(), 0); /* which shards' samplers have run (each written by its own worker) */
    const int st = for_each_shard(g, [&](Shard &s, int i) -> int {
        if (hipSetDevice(s.device) != hipSuccess)
            return MMCMC_ERR_NO_DEVICE;
        const size_t bytes = s.n * row;
        int rc = shard_reserve_sample(s, bytes);
        if (rc != MMCMC_OK)
            return rc;
        uint64_t *acc = accept_counts ? accept_counts + s.first : nullptr;
        void *d_out = n_collect ? s.d_sample : nullptr;
        rc = g->sampler == 2 ? mmcmc_nuts_run(s.nuts, n_collect, n_discard, d_out, 1, g->progress, s.stream)
             : g->sampler    ? mmcmc_hmc_run(s.hmc, n_collect, n_discard, d_out, 1, acc, s.stream)
                             : mmcmc_mh_run(s.mh, n_collect, n_discard, d_out, 1, acc, s.stream);
        if (rc != MMCMC_OK)
            return rc;
        advanced[(size_t)i] = 1;
        hipError_t e = hipSuccess;
        if (out_host && n_collect) /* through the device's own pinned bounce ring: every shard over its own link, side by side */
            e = mm_copy_to_host((char *)out_host + s.first * row, s.d_sample, bytes, s.device, s.stream);
        else
            e = hipStreamSynchronize(s.stream);
        return e == hipSuccess ? MMCMC_OK : (int)e;
    });
    if (st == MMCMC_OK)
        g->last_collect = n_collect;
    else {
        size_t n_adv = 0;
        for (char a : advanced)
            n_adv += a ? 1 : 0;
        if (n_adv > 0 && n_adv < g->sh.size())
            g->broken = true; /* some shards have run and others not: same rule as the asynchronous path */
    }
    return st;
}

/* wait for everything queued on the shards' streams */
int group_sync(Group *g)
{
    if (!g)
        return MMCMC_ERR_INVALID_ARG;
    int prev = -1;
    (void)hipGetDevice(&prev);
    int st = MMCMC_OK;
    for (Shard &s : g->sh) {
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess && st == MMCMC_OK)
            st = (int)e;
    }
    if (prev >= 0)
        (void)hipSetDevice(prev);
    return st;
}

/* Stream timer: stop = 0 records a start event on every shard's stream; stop = 1 records the end events, waits for them
 * and writes the elapsed milliseconds of every shard into ms [n_devices] -- the time the DEVICE spent on what was queued
 * in between (bench.py: kernel time per device of a scaling run, free of the host's launch loop). */
int group_stream_timer(Group *g, int stop, float *ms)
{
    if (!g || (stop && !ms))
        return MMCMC_ERR_INVALID_ARG;
    int prev = -1;
    (void)hipGetDevice(&prev);
    hipError_t e = hipSuccess;
    for (Shard &s : g->sh) {
        if ((e = hipSetDevice(s.device)) != hipSuccess)
            break;
        if (!s.tev0 && (e = hipEventCreate(&s.tev0)) != hipSuccess)
            break;
        if (!s.tev1 && (e = hipEventCreate(&s.tev1)) != hipSuccess)
            break;
        if ((e = hipEventRecord(stop ? s.tev1 : s.tev0, s.stream)) != hipSuccess)
            break;
    }
    if (e == hipSuccess && stop) {
        size_t i = 0;
        for (Shard &s : g->sh) {
            if ((e = hipSetDevice(s.device)) != hipSuccess || (e = hipEventSynchronize(s.tev1)) != hipSuccess ||
                (e = hipEventElapsedTime(&ms[i], s.tev0, s.tev1)) != hipSuccess)
                break;
            ++i;
        }
    }
    if (prev >= 0)
        (void)hipSetDevice(prev);
    return e == hipSuccess ? MMCMC_OK : (int)e;
}

/* how the diagnostics' statistics will travel (decided at creation): the values of *used_rccl */
int group_exchange(Group *g, int *status, int *n_rccl_ranks)
{
    if (!g || !status)
        return MMCMC_ERR_INVALID_ARG;
    *status = g->exchange_status;
    if (n_rccl_ranks)
        *n_rccl_ranks = g->comm_ready ? (int)g->sh.size() : 0;
    return MMCMC_OK;
}

int group_state(Group *g, void *out)
{
    if (!g || !out)
        return MMCMC_ERR_INVALID_ARG;
    /* a broken group's shards stand at different iterations; their states are still what each shard holds -- readable */
    const size_t esz = g->esize();
    return for_each_shard(g, [&](Shard &s, int) -> int {
        void *p = (char *)out + s.first * (size_t)g->dim * esz;
        return g->sampler == 2 ? mmcmc_nuts_state(s.nuts, p) : g->sampler ? mmcmc_hmc_state(s.hmc, p) : mmcmc_mh_state(s.mh, p);
    });
}

/* split-R-hat / ESS of the last run's sample (stats.rs:416-423) over ALL devices' chains */
int group_split_rhat_ess(Group *g, float *rhat, float *ess, int *used_rccl)
{
    if (!g || !rhat || !ess)
        return MMCMC_ERR_INVALID_ARG;
    if (g->broken)
        return MMCMC_ERR_GROUP_BROKEN;
    const size_t n = g->last_collect, m = n / 2, D = (size_t)g->dim, N = g->sh.size();
    if (m < 1)
        return MMCMC_ERR_SHAPE;
    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    clk::time_point t_local = t_begin, t_exch = t_begin;
    auto phases = [&](int rc) { /* local partials | exchange (collectives + cross sums / host copies) | finish */
        const clk::time_point t_end = clk::now();
        g->phase_ms[0] = std::chrono::duration<double, std::milli>(t_local - t_begin).count();
        g->phase_ms[1] = std::chrono::duration<double, std::milli>(t_exch - t_local).count();
        g->phase_ms[2] = std::chrono::duration<double, std::milli>(t_end - t_exch).count();
        return rc;
    };
    const size_t part = 2 * g->cmax * D;             /* per-device means (or ssq), padded to the largest shard */
    const size_t own = 2 * part + m * D;             /* means | ssq | acov */
    const size_t n_cs = N * D * 4 * 3;               /* partial cross-chain sums (f64) of mm_group_cross_sums_kernel */
    const size_t total = own + N * 2 * part + m * D + 2 * n_cs + 2; /* + gathered [N][means | ssq] + reduced acov + the partial sums */
    if (used_rccl)
        *used_rccl = g->exchange_status;
    std::vector<float> h_own(g->use_rccl ? 0 : N * own);
    /* phase 1: every shard's local statistics, finished and checked BEFORE any rank enters a collective -- a shard that
     * failed here would otherwise leave the others waiting in ncclAllGather for good */
    int st = for_each_shard(g, [&](Shard &s, int) -> int {
        if (hipSetDevice(s.device) != hipSuccess)
            return MMCMC_ERR_NO_DEVICE;
        hipError_t e;
        if (total > s.stats_cap) {
            if (s.d_stats)
                (void)hipFree(s.d_stats);
            s.d_stats = nullptr;
            s.stats_cap = 0;
            if ((e = hipMalloc((void **)&s.d_stats, total * sizeof(float))) != hipSuccess)
                return (int)e;
            s.stats_cap = total;
        }
        float *means = s.d_stats, *ssq = means + part, *acov = ssq + part;
        if ((e = hipMemsetAsync(s.d_stats, 0, own * sizeof(float), s.stream)) != hipSuccess)
            return (int)e;
        /* the local statistics are [2][n_i][D] (first halves of the local chains, then their second halves); every
         * rank sends a slot of 2 * cmax * D floats (all-gather wants equal counts), the tail past 2 n_i D stays zero */
        int rc = mmcmc_stats_partials(s.d_sample, g->dtype, s.n, n, D, means, ssq, acov, s.device, s.stream);
        if (rc != MMCMC_OK)
            return rc;
        e = hipStreamSynchronize(s.stream);
        return e == hipSuccess ? MMCMC_OK : (int)e;
    });
    if (st != MMCMC_OK)
        return st;
    t_local = t_exch = clk::now();
    /* phase 2: the exchange */
    st = for_each_shard(g, [&](Shard &s, int i) -> int {
        if (hipSetDevice(s.device) != hipSuccess)
            return MMCMC_ERR_NO_DEVICE;
        hipError_t e;
        float *means = s.d_stats, *ssq = means + part, *acov = ssq + part, *gathered = acov + m * D, *acov_all = gathered + N * 2 * part;
        (void)ssq;
        if (g->use_rccl) {
            /* means and ssq are adjacent: one all-gather of 2 * part floats per rank; lag sums: one all-reduce */
            if (rccl().AllGather(means, gathered, 2 * part, kNcclFloat32, s.comm, s.stream) != 0)
                return (int)hipErrorUnknown;
            if (rccl().AllReduce(acov, acov_all, m * D, kNcclFloat32, kNcclSum, s.comm, s.stream) != 0)
                return (int)hipErrorUnknown;
            if (i == 0 && N <= 64) { /* the cross-chain sums of ALL ranks' half-chains, from shard 0's copy of the gathered statistics */
                GroupCounts cnt;
                for (size_t r = 0; r < N; ++r)
                    cnt.n2[r] = (unsigned int)(2 * g->sh[r].n);
                double *cs = reinterpret_cast<double *>(s.d_stats + ((own + N * 2 * part + m * D + 1) / 2) * 2);
                hipLaunchKernelGGL(mm_group_cross_sums_kernel, dim3((unsigned int)D * 4u, (unsigned int)N), dim3(256), 0, s.stream, gathered,
                                   part, (unsigned int)D, (float)m, cnt, cs);
                if ((e = hipGetLastError()) != hipSuccess)
                    return (int)e;
            }
        } else {
            if ((e = hipMemcpyAsync(h_own.data() + (size_t)i * own, s.d_stats, own * sizeof(float), hipMemcpyDeviceToHost, s.stream)) != hipSuccess)
                return (int)e;
        }
        e = hipStreamSynchronize(s.stream);
        return e == hipSuccess ? MMCMC_OK : (int)e;
    });
    if (st != MMCMC_OK)
        return st;
    t_exch = clk::now();
    /* RCCL path: the finish needs the gathered statistics only through their cross-chain sums, which shard 0 has reduced on
     * the device.  Host-exchange path (a device listed twice, no RCCL): the host walks the statistics in splitcat order (first
     * halves of all chains, then second halves) */
    std::vector<float> acov;
    if (g->use_rccl && N <= 64) {
        /* a few KB come back: the reduced lag sums and the N x 4 partial cross-chain sums per parameter */
        Shard &s0 = g->sh[0];
        (void)hipSetDevice(s0.device);
        const float *acov_all = s0.d_stats + own + N * 2 * part;
        const double *cs = reinterpret_cast<const double *>(s0.d_stats + ((own + N * 2 * part + m * D + 1) / 2) * 2);
        std::vector<double> h_cs(n_cs);
        acov.resize(m * D);
        hipError_t e = hipMemcpy(acov.data(), acov_all, acov.size() * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(h_cs.data(), cs, n_cs * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return (int)e;
        const double c2 = 2.0 * (double)g->n_chains;
        std::vector<double> dsum(D), wsum(D);
        for (size_t d = 0; d < D; ++d) {
            double sd = 0.0, sq = 0.0, ws = 0.0;
            for (size_t r = 0; r < N; ++r)
                for (size_t qtr = 0; qtr < 4; ++qtr) {
                    const double *q3 = h_cs.data() + ((r * D + d) * 4 + qtr) * 3;
                    sd += q3[0];
                    sq += q3[1];
                    ws += q3[2];
                }
            const double ds = sq - sd * sd / c2; /* sum of (mean - overall mean)^2 */
            dsum[d] = ds > 0.0 ? ds : 0.0;
            wsum[d] = ws;
        }
        return phases(mmcmc_stats_finish_sums(dsum.data(), wsum.data(), acov.data(), 2 * g->n_chains, m, D, rhat, ess));
    }
    std::vector<float> g_all(N * 2 * part);
    acov.assign(m * D, 0.f);
    if (g->use_rccl) {
        Shard &s0 = g->sh[0];
        (void)hipSetDevice(s0.device);
        const float *gathered = s0.d_stats + own, *acov_all = gathered + N * 2 * part;
        hipError_t e = hipMemcpy(g_all.data(), gathered, g_all.size() * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(acov.data(), acov_all, acov.size() * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return (int)e;
    } else {
        for (size_t i = 0; i < N; ++i) {
            std::memcpy(g_all.data() + i * 2 * part, h_own.data() + i * own, 2 * part * sizeof(float));
            const float *a = h_own.data() + i * own + 2 * part;
            for (size_t k = 0; k < m * D; ++k)
                acov[k] += a[k];
        }
    }
    const size_t C = g->n_chains;
    std::vector<float> means(2 * C * D), ssq(2 * C * D);
    for (size_t i = 0; i < N; ++i) {
        const Shard &s = g->sh[i];
        const float *mi = g_all.data() + i * 2 * part, *qi = mi + part;
        for (int half = 0; half < 2; ++half) {
            std::memcpy(means.data() + ((size_t)half * C + s.first) * D, mi + (size_t)half * s.n * D, s.n * D * sizeof(float));
            std::memcpy(ssq.data() + ((size_t)half * C + s.first) * D, qi + (size_t)half * s.n * D, s.n * D * sizeof(float));
        }
    }
    return phases(mmcmc_stats_finish(means.data(), ssq.data(), acov.data(), 2 * C, m, D, rhat, ess));
}

int group_stats_phases(Group *g, double *ms3)
{
    if (!g || !ms3)
        return MMCMC_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i)
        ms3[i] = g->phase_ms[i];
    return MMCMC_OK;
}

} // namespace

struct mmcmc_hmc_group {
    Group *g;
};
struct mmcmc_mh_group {
    Group *g;
};
struct mmcmc_nuts_group {
    Group *g;
};

extern "C" {

int mmcmc_hmc_group_create(mmcmc_hmc_group **out, const mmcmc_target_desc *target, const void *init, size_t n_chains,
                           double step_size, int n_leapfrog, int dtype, const int *devices, int n_devices)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    Group *g = nullptr;
    const int st = group_create(&g, 1, target, nullptr, init, n_chains, step_size, n_leapfrog, dtype, devices, n_devices);
    if (st != MMCMC_OK)
        return st;
    *out = new (std::nothrow) mmcmc_hmc_group{g};
    if (!*out) {
        group_destroy(g);
        return (int)hipErrorOutOfMemory;
    }
    return MMCMC_OK;
}
int mmcmc_hmc_group_seed(mmcmc_hmc_group *h, uint64_t seed) { return h ? group_seed(h->g, seed) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_group_set_chain_offset(mmcmc_hmc_group *h, uint64_t off) { return h ? group_set_chain_offset(h->g, off) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_group_run(mmcmc_hmc_group *h, size_t n_collect, size_t n_discard, void *out_host, uint64_t *accept_counts)
{
    return h ? group_run(h->g, n_collect, n_discard, out_host, accept_counts, false) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_group_run_async(mmcmc_hmc_group *h, size_t n_collect, size_t n_discard)
{
    return h ? group_run(h->g, n_collect, n_discard, nullptr, nullptr, true) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_group_stats_phases(mmcmc_hmc_group *h, double *ms3) { return h ? group_stats_phases(h->g, ms3) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_group_state(mmcmc_hmc_group *h, void *out) { return h ? group_state(h->g, out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_group_split_rhat_mean_ess(mmcmc_hmc_group *h, float *rhat, float *ess, int *used_rccl)
{
    return h ? group_split_rhat_ess(h->g, rhat, ess, used_rccl) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_group_shard(mmcmc_hmc_group *h, int i, int *device, size_t *first_chain, size_t *n_chains, void **sample_dev)
{
    if (!h || i < 0 || (size_t)i >= h->g->sh.size())
        return MMCMC_ERR_INVALID_ARG;
    const Shard &s = h->g->sh[(size_t)i];
    if (device)
        *device = s.device;
    if (first_chain)
        *first_chain = s.first;
    if (n_chains)
        *n_chains = s.n;
    if (sample_dev)
        *sample_dev = s.d_sample;
    return MMCMC_OK;
}
int mmcmc_hmc_group_sync(mmcmc_hmc_group *h) { return h ? group_sync(h->g) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_group_stream_timer(mmcmc_hmc_group *h, int stop, float *ms_per_device)
{
    return h ? group_stream_timer(h->g, stop, ms_per_device) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_group_exchange(mmcmc_hmc_group *h, int *status, int *n_rccl_ranks)
{
    return h ? group_exchange(h->g, status, n_rccl_ranks) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_group_destroy(mmcmc_hmc_group *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    const int st = group_destroy(h->g);
    delete h;
    return st;
}

int mmcmc_mh_group_create(mmcmc_mh_group **out, const mmcmc_target_desc *target, const mmcmc_proposal_desc *proposal,
                          const void *init, size_t n_chains, int dtype, const int *devices, int n_devices)
{
    if (!out || !proposal)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    Group *g = nullptr;
    const int st = group_create(&g, 0, target, proposal, init, n_chains, proposal->std, 0, dtype, devices, n_devices);
    if (st != MMCMC_OK)
        return st;
    *out = new (std::nothrow) mmcmc_mh_group{g};
    if (!*out) {
        group_destroy(g);
        return (int)hipErrorOutOfMemory;
    }
    return MMCMC_OK;
}
int mmcmc_mh_group_seed(mmcmc_mh_group *h, uint64_t seed) { return h ? group_seed(h->g, seed) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_group_set_chain_offset(mmcmc_mh_group *h, uint64_t off) { return h ? group_set_chain_offset(h->g, off) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_group_run(mmcmc_mh_group *h, size_t n_collect, size_t n_discard, void *out_host, uint64_t *accept_counts)
{
    return h ? group_run(h->g, n_collect, n_discard, out_host, accept_counts, false) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_group_run_async(mmcmc_mh_group *h, size_t n_collect, size_t n_discard)
{
    return h ? group_run(h->g, n_collect, n_discard, nullptr, nullptr, true) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_group_stats_phases(mmcmc_mh_group *h, double *ms3) { return h ? group_stats_phases(h->g, ms3) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_group_state(mmcmc_mh_group *h, void *out) { return h ? group_state(h->g, out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_group_split_rhat_mean_ess(mmcmc_mh_group *h, float *rhat, float *ess, int *used_rccl)
{
    return h ? group_split_rhat_ess(h->g, rhat, ess, used_rccl) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_group_sync(mmcmc_mh_group *h) { return h ? group_sync(h->g) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_group_stream_timer(mmcmc_mh_group *h, int stop, float *ms_per_device)
{
    return h ? group_stream_timer(h->g, stop, ms_per_device) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_group_exchange(mmcmc_mh_group *h, int *status, int *n_rccl_ranks)
{
    return h ? group_exchange(h->g, status, n_rccl_ranks) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_group_destroy(mmcmc_mh_group *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    const int st = group_destroy(h->g);
    delete h;
    return st;
}

/* NUTS::run / run_progress (nuts.rs:163-170, 194-338) over N devices: `init` [n_chains, dim] doubles, mode as in
 * mmcmc_nuts_create; the sample is of the mode's tensor type (f32 for modes 0 and 1, f64 for mode 2) */
int mmcmc_nuts_group_create(mmcmc_nuts_group **out, const mmcmc_target_desc *target, const double *init, size_t n_chains,
                            double target_accept_p, int mode, const int *devices, int n_devices)
{
    if (!out || mode < 0 || mode > 2)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    Group *g = nullptr;
    const int st = group_create(&g, 2, target, nullptr, init, n_chains, target_accept_p, mode, mode == 2 ? MMCMC_F64 : MMCMC_F32, devices,
                                n_devices);
    if (st != MMCMC_OK)
        return st;
    *out = new (std::nothrow) mmcmc_nuts_group{g};
    if (!*out) {
        group_destroy(g);
        return (int)hipErrorOutOfMemory;
    }
    return MMCMC_OK;
}
int mmcmc_nuts_group_seed(mmcmc_nuts_group *h, uint64_t seed) { return h ? group_seed(h->g, seed) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_group_set_chain_offset(mmcmc_nuts_group *h, uint64_t off) { return h ? group_set_chain_offset(h->g, off) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_group_set_max_depth(mmcmc_nuts_group *h, int max_depth)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    for (Shard &s : h->g->sh) {
        const int rc = mmcmc_nuts_set_max_depth(s.nuts, max_depth);
        if (rc != MMCMC_OK)
            return rc;
    }
    return MMCMC_OK;
}
int mmcmc_nuts_group_run(mmcmc_nuts_group *h, size_t n_collect, size_t n_discard, void *out_host, int progress)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->g->progress = progress ? 1 : 0;
    return group_run(h->g, n_collect, n_discard, out_host, nullptr, false);
}
int mmcmc_nuts_group_stats_phases(mmcmc_nuts_group *h, double *ms3) { return h ? group_stats_phases(h->g, ms3) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_group_state(mmcmc_nuts_group *h, void *out) { return h ? group_state(h->g, out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_group_leapfrog_counts(mmcmc_nuts_group *h, uint64_t *out)
{
    if (!h || !out)
        return MMCMC_ERR_INVALID_ARG;
    return for_each_shard(h->g, [&](Shard &s, int) -> int { return mmcmc_nuts_leapfrog_counts(s.nuts, out + s.first); });
}
int mmcmc_nuts_group_split_rhat_mean_ess(mmcmc_nuts_group *h, float *rhat, float *ess, int *used_rccl)
{
    return h ? group_split_rhat_ess(h->g, rhat, ess, used_rccl) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_group_sync(mmcmc_nuts_group *h) { return h ? group_sync(h->g) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_group_stream_timer(mmcmc_nuts_group *h, int stop, float *ms_per_device)
{
    return h ? group_stream_timer(h->g, stop, ms_per_device) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_group_exchange(mmcmc_nuts_group *h, int *status, int *n_rccl_ranks)
{
    return h ? group_exchange(h->g, status, n_rccl_ranks) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_group_destroy(mmcmc_nuts_group *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    const int st = group_destroy(h->g);
    delete h;
    return st;
}

} /* extern "C" */
