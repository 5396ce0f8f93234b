/*
 * mm_nuts_api.hip -- C ABI of the NUTS sampler (include/mmcmc.h: mmcmc_nuts_*).  Host logic only.
 * Mirrors NUTS::{new, set_seed, run, run_progress} (nuts.rs:123-170, 194-353) over NUTSChain (nuts.rs:410-691).
 */
#include "../../include/mmcmc.h"
#include "mm_hostcopy.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <map>
#include <vector>

#include <type_traits>

#include "mm_nuts_kernels.h"
#include "mm_nuts_lg.h"
#include "mm_nuts_generic.h"
#include "mm_tuning.h"
#include "mm_rtc.h"
#include "mm_params.h"

#define MM_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return (int)_e;                                                                                       \
    } while (0)

namespace {

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int d)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        (void)hipSetDevice(d);
    }
    ~DevGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

/* 0: normal; 1: inside the self-test of a run-time compiled unit; 2: create without run-time compiled built-in units */
thread_local int g_rtc_create_mode = 0;

struct NutsBase {
    virtual ~NutsBase() {}
    virtual int run(size_t n_collect, size_t n_discard, void *out, int out_is_device, int progress, void *stream) = 0;
    virtual int state(void *out) = 0;
    virtual int adapt_state(double *out) = 0;
    virtual int leapfrog_counts(uint64_t *out) = 0;
    virtual int depth_histogram(uint32_t *out) = 0;
    virtual int set_variant(int v) = 0;
    int variant = 0; /* 0: one chain per lane, lanes in step; 4: one chain per lane, asynchronous lanes (default for
                      * dim <= 8); 1: lane-group / MFMA (mm_nuts_lg.h); 2: + tree-depth compaction, one
                        launch per level; 3: + compaction by a persistent scheduler */
    int compaction_start = 5; /* variants 2, 3: doublings below this run before the first compaction */
    bool compaction_auto = true; /* variant 3: choose it from the depths seen so far (until set_compaction is called) */
    int compaction_groups = 0; /* variant 2: chain groups with their own launch sequence (0 = choose) */
    int device = 0, mode = 0, kind = 0, dim = 0;
    const void *rtc_unit = nullptr; /* the run-time compiled unit whose kernels this handle launches (mm_rtc.hip), or NULL */
    int rtc_run_kernel = 2; /* which of the unit's run kernels: 2 asynchronous lanes, leaves in pairs (what handles launch);
                               0 the lanes in step (launched only by rtc_unit_verified, as the second opinion on a user target) */
    size_t n_chains = 0;
    uint64_t seed = 0, chain_offset = 0;
    uint32_t m = 0; /* self.m: transitions taken so far */
    int max_depth = 10;
    double target_accept_p = 0.8;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    mmcmc_timing timing{};
};

template <class TT, class ST> const mm_nuts_entry<TT, ST> *nuts_table(int *n);
template <> const mm_nuts_entry<float, double> *nuts_table<float, double>(int *n) { return mm_nuts_table_m0(n); }
template <> const mm_nuts_entry<float, float> *nuts_table<float, float>(int *n) { return mm_nuts_table_m1(n); }
template <> const mm_nuts_entry<double, double> *nuts_table<double, double>(int *n) { return mm_nuts_table_m2(n); }

inline hipError_t launch_generic(const mm_gen_nuts_args<float, double> &a, int init, hipStream_t st) { return mm_launch_nuts_generic_m0(a, init, st); }
inline hipError_t launch_generic(const mm_gen_nuts_args<float, float> &a, int init, hipStream_t st) { return mm_launch_nuts_generic_m1(a, init, st); }
inline hipError_t launch_generic(const mm_gen_nuts_args<double, double> &a, int init, hipStream_t st) { return mm_launch_nuts_generic_m2(a, init, st); }

template <class TT, class ST> struct Nuts : NutsBase {
    const mm_nuts_entry<TT, ST> *k = nullptr;
    /* run-time-dimension path (mm_nuts_generic.h, variant 6): the only one where the table has no (kind, dim) entry */
    bool generic_ok = false;
    const mm_user_target *user = nullptr; /* run-time compiled target (mm_rtc.hip, variant 7) */
    size_t user_stack_bytes = 0, user_lds = 0;
    TT *d_gstore = nullptr;
    int gstore_depth = 0;
    size_t g_pad = 0;
    mm_tparams<TT> P;
    TT *d_state = nullptr, *d_mat = nullptr;
    mm_nuts_adapt<ST> *d_adapt = nullptr;
    unsigned long long *d_nlf = nullptr;
    unsigned int *d_hist = nullptr;
    unsigned char *d_scratch = nullptr;
    bool stack_in_lds = true;
    const mm_nuts_lg_entry *lg = nullptr;
    double *d_lg_scratch = nullptr, *d_lg_rec = nullptr;
    unsigned int *d_lg_lists = nullptr, *d_lg_counts = nullptr;
    mm_lgq_ctrl *d_lg_ctrl = nullptr;
    unsigned int n_resident_waves = 1024;
    int lgq_occ = 1; /* waves per SIMD of the persistent scheduler (MMCMC_LGQ_OCC overrides) */
    unsigned int n_launches_run = 0; /* sampling launches of the current run() (reported by timing()) */
    int auto_level = -1;             /* first compaction level chosen from the depth histogram, once it is known */
    size_t c_pad = 0;
    static constexpr int kMaxGroups = 16;
    hipStream_t lg_streams[kMaxGroups] = {};
    hipEvent_t lg_fork = nullptr, lg_join[kMaxGroups] = {};

    ~Nuts() override
    {
        DevGuard g(device);
        if (stream)
            (void)hipStreamSynchronize(stream);
        (void)hipFree(d_state);
        (void)hipFree(d_mat);
        (void)hipFree(d_adapt);
        (void)hipFree(d_nlf);
        (void)hipFree(d_hist);
        (void)hipFree(d_scratch);
        (void)hipFree(d_gstore);
        (void)hipFree(d_lg_scratch);
        (void)hipFree(d_lg_rec);
        (void)hipFree(d_lg_lists);
        (void)hipFree(d_lg_counts);
        (void)hipFree(d_lg_ctrl);
        for (int i = 0; i < kMaxGroups; ++i) {
            if (lg_streams[i])
                (void)hipStreamDestroy(lg_streams[i]);
            if (lg_join[i])
                (void)hipEventDestroy(lg_join[i]);
        }
        if (lg_fork)
            (void)hipEventDestroy(lg_fork);
        if (ev0)
            (void)hipEventDestroy(ev0);
        if (ev1)
            (void)hipEventDestroy(ev1);
        if (stream)
            (void)hipStreamDestroy(stream);
    }

    int setup(const mmcmc_target_desc *t, const double *init)
    {
        int n = 0;
        const mm_nuts_entry<TT, ST> *tab = nuts_table<TT, ST>(&n);
        for (int i = 0; i < n; ++i)
            if (tab[i].kind == t->kind && tab[i].dim == t->dim)
                k = &tab[i];
        generic_ok = mm_generic_kind_ok(t->kind) && t->dim >= 1;
        bool builtin_user = false;
        const int create_mode = std::is_same<TT, double>::value ? 2 : (std::is_same<ST, double>::value ? 0 : 1);
        if (t->kind < MM_USER_KIND_BASE && !k && generic_ok && t->dim <= 32 && g_rtc_create_mode != 2) {
            /* a built-in target at a dimension without a compiled instance: its functor compiled into the one-chain-per-lane
             * kernels on first use (hipRTC, mm_rtc_builtin_nuts); NULL keeps the run-time-D kernel */
            DevGuard gb(device);
            user = mm_rtc_builtin_nuts(t->kind, t->dim);
            /* ... and only if the unit has this type mode's init and pair kernels (mm_rtc_nuts_usable) */
            if (user && !mm_rtc_nuts_usable(user, create_mode))
                user = nullptr;
            builtin_user = user != nullptr;
        }
        if (t->kind >= MM_USER_KIND_BASE || builtin_user) {
            if (!builtin_user)
                user = mm_rtc_find(t->kind);
            if (!user)
                return MMCMC_ERR_INVALID_ARG;
            if (mm_rtc_is_model(user) || mm_rtc_is_discrete(user) || (!builtin_user && mm_rtc_is_internal(user)))
                return MMCMC_ERR_UNSUPPORTED; /* a target + proposal model / an integer-state model has no NUTS kernels; a unit
                                                 the library built for itself is not a kind a caller may name */
            if (mm_rtc_dim(user) != t->dim)
                return MMCMC_ERR_SHAPE;
            if (!builtin_user) {
                DevGuard gu(device);
                if (!mm_rtc_nuts_usable(user, create_mode))
                    return MMCMC_ERR_UNSUPPORTED; /* this dimension and precision need more registers than a lane has */
            }
            /* mm_nuts_stack_layout<TT, ST, D>::bytes and the output tile of mm_tile<TT, D>, for a run-time D */
            const size_t D = (size_t)t->dim, a16 = 15;
            user_stack_bytes = (((size_t)MM_NUTS_JMAX * 3 * D * 64 * sizeof(TT) + a16) / 16 * 16) +
                               (((size_t)MM_NUTS_JMAX * 64 * sizeof(ST) + a16) / 16 * 16) +
                               (((size_t)MM_NUTS_JMAX * 3 * 64 * sizeof(uint32_t) + a16) / 16 * 16);
            const int target = sizeof(TT) == 4 ? 96 : 48;
            const int tile_t = (target / (int)D) >= 2 ? ((target / (int)D) & ~1) : 1;
            const int run_len = tile_t * (int)D, epl = (int)(16 / sizeof(TT));
            const int stride = (run_len % epl) == 0 ? run_len + epl : (run_len | 1);
            user_lds = ((size_t)64 * stride * sizeof(TT) + 15) / 16 * 16; /* must mirror mm_kernels.h (mm_tile_default) */
        } else if (!k && !generic_ok) {
            return MMCMC_ERR_UNSUPPORTED;
        }
        rtc_unit = user;
        if (user && !builtin_user) {
            for (int i = 0; i < 8; ++i) /* P.p[0..8) = the description's params (mm_rtc.hip) */
                P.p[i] = (TT)t->params[i];
            P.mat = nullptr;
        } else if (mm_fill_params<TT>(t->kind, t->params, &P) != 0) {
            return MMCMC_ERR_INVALID_ARG;
        }
        g_pad = (n_chains + 63) / 64 * 64;
        DevGuard g(device);
        const size_t cd = n_chains * (size_t)dim;
        if (t->kind == MMCMC_GAUSSIAN_ND || (user && t->matrix)) {
            std::vector<TT> h((size_t)dim * dim);
            for (size_t i = 0; i < h.size(); ++i)
                h[i] = (TT)t->matrix[i];
            MM_HIP(hipMalloc((void **)&d_mat, h.size() * sizeof(TT)));
            MM_HIP(hipMemcpy(d_mat, h.data(), h.size() * sizeof(TT), hipMemcpyHostToDevice));
            P.mat = d_mat;
        }
        std::vector<TT> h(cd);
        for (size_t i = 0; i < cd; ++i)
            h[i] = (TT)init[i]; /* Vec<Vec<T>> -> backend float tensor (nuts.rs:411-413) */
        MM_HIP(hipMalloc((void **)&d_state, cd * sizeof(TT)));
        MM_HIP(hipMemcpy(d_state, h.data(), cd * sizeof(TT), hipMemcpyHostToDevice));
        /* NUTSChain::new (nuts.rs:415-433): epsilon = -1 (sentinel), epsilon_bar = 1, h_bar = 0, mu = ln 10 */
        std::vector<mm_nuts_adapt<ST>> ad(n_chains);
        for (auto &a : ad) {
            a.epsilon = (ST)-1;
            a.epsilon_bar = (ST)1;
            a.h_bar = (ST)0;
            a.mu = (ST)std::log(10.0);
        }
        MM_HIP(hipMalloc((void **)&d_adapt, n_chains * sizeof(mm_nuts_adapt<ST>)));
        MM_HIP(hipMemcpy(d_adapt, ad.data(), n_chains * sizeof(mm_nuts_adapt<ST>), hipMemcpyHostToDevice));
        MM_HIP(hipMalloc((void **)&d_nlf, n_chains * sizeof(unsigned long long)));
        MM_HIP(hipMemset(d_nlf, 0, n_chains * sizeof(unsigned long long)));
        MM_HIP(hipMalloc((void **)&d_hist, (MM_NUTS_JMAX + 1) * sizeof(unsigned int)));
        MM_HIP(hipMemset(d_hist, 0, (MM_NUTS_JMAX + 1) * sizeof(unsigned int)));
        if (k) {
            /* the pending-subtree stack goes to LDS when tile + stack fit comfortably, else to HBM scratch */
            stack_in_lds = (k->tile_bytes_per_wave + k->stack_bytes_per_wave) <= 40 * 1024;
            /* the scratch area also serves the asynchronous-lane kernel when ITS stack (max_depth levels, no tile) is
             * over the LDS limit; max_depth can change after create, so allocate whenever the full stack is */
            if (!stack_in_lds || k->stack_bytes_per_wave + 4096 > 40 * 1024) { /* + the pair kernel's ring of uniforms */
                const size_t waves = (n_chains + 63) / 64;
                MM_HIP(hipMalloc((void **)&d_scratch, waves * k->stack_bytes_per_wave));
            }
        } else if (user) {
            variant = 7;
            const size_t waves0 = (n_chains + 63) / 64;
            MM_HIP(hipMalloc((void **)&d_scratch, waves0 * user_stack_bytes));
            stack_in_lds = false;
        } else {
            variant = 6;
        }
        if (std::is_same<TT, double>::value && std::is_same<ST, double>::value && t->kind == MMCMC_GAUSSIAN_ND) {
            int nl = 0;
            const mm_nuts_lg_entry *lt = mm_nuts_lg_table(&nl);
            for (int i = 0; i < nl; ++i)
                if (lt[i].dim == t->dim)
                    lg = &lt[i];
            if (lg) {
                const size_t waves = (n_chains + 15) / 16;
                c_pad = waves * 16;
                MM_HIP(hipMalloc((void **)&d_lg_scratch, waves * lg->scratch_doubles_per_wave * sizeof(double)));
                MM_HIP(hipMalloc((void **)&d_lg_rec, c_pad * lg->rec_doubles_per_chain * sizeof(double)));
                MM_HIP(hipMemset(d_lg_rec, 0, c_pad * lg->rec_doubles_per_chain * sizeof(double)));
                MM_HIP(hipMalloc((void **)&d_lg_lists, (size_t)MM_LGQ_SHARDS * MM_LGQ_NQ * c_pad * sizeof(unsigned int)));
                MM_HIP(hipMalloc((void **)&d_lg_ctrl, sizeof(mm_lgq_ctrl)));
                hipDeviceProp_t prop;
                MM_HIP(hipGetDeviceProperties(&prop, device));
                n_resident_waves = (unsigned int)prop.multiProcessorCount * 4u; /* one wave per SIMD */
                MM_HIP(hipMalloc((void **)&d_lg_counts, (size_t)kMaxGroups * 2 * (MM_NUTS_JMAX + 1) * sizeof(unsigned int)));
                variant = 3; /* the default where it exists */
            }
        }
        if (!lg && k && k->run_async)
            variant = 4; /* asynchronous lanes */
        if (!lg && k && k->run_pair)
            variant = 5; /* asynchronous lanes, leaves in pairs: the default for the one-chain-per-lane kernels */
        MM_HIP(hipStreamCreateWithFlags(&stream, hipStreamDefault));
        MM_HIP(hipEventCreate(&ev0));
        MM_HIP(hipEventCreate(&ev1));
        return MMCMC_OK;
    }

    int set_variant(int v) override
    {
        if ((v == 0 && k) || (v >= 1 && v <= 3 && lg) || (v == 4 && k && k->run_async) || (v == 5 && k && k->run_pair) ||
            (v == 6 && generic_ok) || (v == 7 && user)) {
            variant = v;
            return MMCMC_OK;
        }
        return (v >= 0 && v <= 7) ? MMCMC_ERR_UNSUPPORTED : MMCMC_ERR_INVALID_ARG;
    }

    /* the lane-group launch exists for <double, double> only; the other instantiations never reach it */
    hipError_t run_lg(const mm_nuts_args<TT, ST> &a, hipStream_t st)
    {
        if constexpr (std::is_same<TT, double>::value && std::is_same<ST, double>::value) {
            mm_nuts_lg_args g;
            g.mat = d_mat;
            g.state = a.state;
            g.adapt = a.adapt;
            g.out = a.out;
            g.n_leapfrog = a.n_leapfrog;
            g.depth_hist = a.depth_hist;
            g.n_chains = a.n_chains;
            g.seed = a.seed;
            g.chain_offset = a.chain_offset;
            g.n_total = a.n_total;
            g.m0 = a.m0;
            g.n_pre = a.n_pre;
            g.n_rec = a.n_rec;
            g.write_initial = a.write_initial;
            g.out_t0 = a.out_t0;
            g.n_discard = a.n_discard;
            g.max_depth = a.max_depth;
            g.target_accept_p = a.target_accept_p;
            g.scratch = d_lg_scratch;
            g.rec = d_lg_rec;
            g.c_pad = c_pad;
            g.lists = d_lg_lists;
            g.counts = d_lg_counts;
            g.j0 = compaction_start < max_depth ? compaction_start : max_depth;
            g.j = 0;
            g.m = 0;
            g.row = 0xffffffffu;
            g.ctrl = nullptr;
            g.slots = nullptr;
            const unsigned int total = a.n_pre + a.n_rec;
            /* few chains cannot fill the queues of a scheduler: one launch, every wave keeps its chains */
            if (variant == 1 || total == 0 || (variant == 3 && n_chains < 2048))
                return lg->run(g, st);
            if (variant == 3 && c_pad < (1ull << MM_LGQ_ID_BITS)) {
                /* persistent scheduler: one resident wave per SIMD (fewer if there are fewer groups of 16 chains) */
                g.ctrl = d_lg_ctrl;
                g.slots = d_lg_lists;
                /* idle polls before a wave settles for less than a full unit.  Round 2 (65 536 chains): 4 -> 516.5 ms, 16 -> 518.5,
                 * 64 -> 520.9, 256 -> 530.3; round 6, with the leaf loop 25 % faster: 0 / 1 / 2 / 4 / 8 -> 347.9 / 348.0 / 348.5 /
                 * 349.3 / 349.7 ms at 15.75-15.77 chains per unit (profiles/r6zh_lgq_patience.log): waiting buys no fuller units */
                g.patience = 1u;
                if (const char *ev = mm_tuning_env("MMCMC_LGQ_PATIENCE"))
                    g.patience = (unsigned int)atoi(ev);
                /* a wave stays with the chains it kept only when the queue of their level makes a FULL unit of them; else they
                 * are queued and the wave takes the deepest full unit there is (config 5 with the edges in registers: 10 ->
                 * 408.6 ms, 12 -> 390.1, 14 -> 381.2, 16 -> 377.5: 15.27 -> 15.78 chains per unit;
                 * profiles/r5zzz_lgq_min_unit.log) */
                g.min_unit = 16u;
                if (const char *ev = mm_tuning_env("MMCMC_LGQ_MIN_UNIT"))
                    g.min_unit = (unsigned int)atoi(ev);
                const unsigned int groups16 = (unsigned int)(c_pad / 16);
                /* waves per SIMD the scheduler is built for (mm_lg_cfg): 2 wherever two per SIMD can be filled */
                int occ = lgq_occ;
                if (const char *ev = mm_tuning_env("MMCMC_LGQ_OCC"))
                    occ = atoi(ev) == 2 ? 2 : 1;
                if (groups16 < 2u * n_resident_waves)
                    occ = 1;
                const unsigned int resident = n_resident_waves * (unsigned int)occ;
                unsigned int nw = groups16 < resident ? groups16 : resident;
                if (const char *ev = mm_tuning_env("MMCMC_LGQ_WAVES")) {
                    const unsigned int w = (unsigned int)atoi(ev);
                    if (w >= 1 && w < nw)
                        nw = w;
                }
                /* one launch of the scheduler; the kernel reports a stuck queue instead of hanging */
                auto launch = [&](const mm_nuts_lg_args &gq) -> hipError_t {
                    ++n_launches_run;
                    hipError_t e = lg->run_queue(gq, nw, occ, st);
                    if (e != hipSuccess)
                        return e;
                    mm_lgq_ctrl hc;
                    if ((e = hipMemcpyAsync(&hc, d_lg_ctrl, sizeof(hc), hipMemcpyDeviceToHost, st)) != hipSuccess)
                        return e;
                    if ((e = hipStreamSynchronize(st)) != hipSuccess)
                        return e;
                    if (mm_tuning_env("MMCMC_LGQ_STATS"))
                        fprintf(stderr, "lgq: first level %d units %llu chains %llu (%.2f per unit) leaf iterations %llu idle polls %llu error %u ticks pick %.3g fetch %.3g work %.3g handover %.3g\n",
                                gq.j0, hc.stat_units, hc.stat_chains,
                                (double)hc.stat_chains / (double)(hc.stat_units ? hc.stat_units : 1), hc.stat_leaf_iters,
                                hc.stat_polls, (unsigned int)hc.error, (double)hc.stat_t[0], (double)hc.stat_t[1],
                                (double)hc.stat_t[2], (double)hc.stat_t[3]);
                    return (hc.error != 0ull || hc.remaining != 0ull) ? hipErrorLaunchFailure : hipSuccess;
                };
                /* The first compaction level pays when it sits just below the depth most trees reach (config 5, trees
                 * 7 - 8 deep: level 5 660 ms, 7 648 ms, 8 782 ms at 65 536 chains; 234 -> 203 ms at 16 384).  Unless the
                 * caller fixed it, it is the mode of the handle's depth histogram minus one; a fresh handle runs its
                 * first 16 transitions as a pilot with the default to have a histogram.  Splitting a run never changes a
                 * result (the stream is keyed by chain and step). */
                auto level_from_history = [&](int *level) -> hipError_t {
                    unsigned int hh[MM_NUTS_JMAX + 1];
                    hipError_t e = hipMemcpyAsync(hh, d_hist, sizeof(hh), hipMemcpyDeviceToHost, st);
                    if (e != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess)
                        return e;
                    unsigned long long sum = 0;
                    int mode = 0;
                    for (int i = 0; i <= MM_NUTS_JMAX; ++i) {
                        sum += hh[i];
                        if (hh[i] > hh[mode])
                            mode = i;
                    }
                    if (sum >= 8ull * n_chains) {
                        int l = mode - 1;
                        l = l < 3 ? 3 : l;
                        *level = l < max_depth ? l : max_depth;
                    }
                    return hipSuccess;
                };
                if (compaction_auto) {
                    /* once the histogram has fixed the level it is kept: no device-to-host copy and no stream
                     * synchronisation per run() after that */
                    int level = auto_level;
                    hipError_t e = level < 0 ? level_from_history(&level) : hipSuccess;
                    if (e != hipSuccess)
                        return e;
                    if (level >= 0)
                        auto_level = level;
                    unsigned int pilot = 16;
                    if (const char *ev = mm_tuning_env("MMCMC_LGQ_PILOT"))
                        pilot = (unsigned int)atoi(ev);
                    if (level < 0 && total >= 4 * pilot) {
                        mm_nuts_lg_args g1 = g, g2 = g;
                        if (const char *ev = mm_tuning_env("MMCMC_LGQ_PILOT_LEVEL"))
                            g1.j0 = atoi(ev);
                        g1.n_pre = a.n_pre < pilot ? a.n_pre : pilot;
                        g1.n_rec = pilot - g1.n_pre;
                        if ((e = launch(g1)) != hipSuccess)
                            return e;
                        if ((e = level_from_history(&level)) != hipSuccess)
                            return e;
                        if (level >= 0)
                            auto_level = level;
                        g2.m0 = a.m0 + pilot;
                        g2.n_pre = a.n_pre - g1.n_pre;
                        g2.n_rec = a.n_rec - g1.n_rec;
                        g2.write_initial = 0;
                        g2.out_t0 = a.out_t0 + (a.write_initial ? 1u : 0u) + g1.n_rec;
                        if (level >= 0)
                            g2.j0 = level < max_depth ? level : max_depth;
                        return launch(g2);
                    }
                    if (level >= 0)
                        g.j0 = level < max_depth ? level : max_depth;
                }
                return launch(g);
            }
            /* Tree-depth compaction: 1 + (max_depth - j0) launches per transition.  The late launches of a
             * transition hold few chains (a few per cent reach the deepest level), so the chains are split into
             * groups, each with its own launch sequence on its own stream: one group's narrow launches run beside
             * another's wide ones.  Groups are contiguous blocks of chains (a multiple of 16 each). */
            int n_groups = compaction_groups > 0 ? compaction_groups : (int)(n_chains / 16384);
            n_groups = n_groups < 1 ? 1 : (n_groups > kMaxGroups ? kMaxGroups : n_groups);
            const size_t waves = c_pad / 16;
            if ((size_t)n_groups > waves)
                n_groups = (int)waves;
            hipError_t e = hipMemsetAsync(d_lg_counts, 0, (size_t)kMaxGroups * 2 * (MM_NUTS_JMAX + 1) * sizeof(unsigned int), st);
            if (e != hipSuccess)
                return e;
            if (!lg_fork && (e = hipEventCreateWithFlags(&lg_fork, hipEventDisableTiming)) != hipSuccess)
                return e;
            if ((e = hipEventRecord(lg_fork, st)) != hipSuccess)
                return e;
            for (int gi = 0; gi < n_groups; ++gi) {
                if (!lg_streams[gi] && (e = hipStreamCreateWithFlags(&lg_streams[gi], hipStreamNonBlocking)) != hipSuccess)
                    return e;
                if (!lg_join[gi] && (e = hipEventCreateWithFlags(&lg_join[gi], hipEventDisableTiming)) != hipSuccess)
                    return e;
                const size_t w0 = waves * gi / n_groups, w1 = waves * (gi + 1) / n_groups;
                const size_t off = w0 * 16;
                mm_nuts_lg_args q = g;
                q.n_chains = (w1 * 16 < a.n_chains ? w1 * 16 : a.n_chains) - off;
                q.c_pad = (w1 - w0) * 16;
                q.chain_offset = a.chain_offset + off;
                q.state = a.state + off * (size_t)dim;
                q.adapt = a.adapt + off;
                q.out = a.out ? a.out + off * a.n_total * (size_t)dim : nullptr;
                q.n_leapfrog = a.n_leapfrog ? a.n_leapfrog + off : nullptr;
                q.scratch = d_lg_scratch + w0 * lg->scratch_doubles_per_wave;
                q.rec = d_lg_rec + off * lg->rec_doubles_per_chain;
                q.lists = d_lg_lists + (size_t)MM_NUTS_JMAX * off;
                q.counts = d_lg_counts + (size_t)gi * 2 * (MM_NUTS_JMAX + 1);
                hipStream_t gs = lg_streams[gi];
                if ((e = hipStreamWaitEvent(gs, lg_fork, 0)) != hipSuccess)
                    return e;
                for (unsigned int t = 0; t < total && e == hipSuccess; ++t) {
                    q.m = a.m0 + t + 1;
                    q.write_initial = (t == 0) ? a.write_initial : 0u;
                    q.row = (t >= a.n_pre) ? (a.out_t0 + (a.write_initial ? 1u : 0u) + (t - a.n_pre)) : 0xffffffffu;
                    e = lg->run_transition(q, gs);
                }
                if (e != hipSuccess)
                    return e;
                if ((e = hipEventRecord(lg_join[gi], gs)) != hipSuccess)
                    return e;
                if ((e = hipStreamWaitEvent(st, lg_join[gi], 0)) != hipSuccess)
                    return e;
            }
            return hipSuccess;
        } else {
            (void)a;
            (void)st;
            return hipErrorInvalidValue;
        }
    }
    hipError_t init_lg(hipStream_t st)
    {
        if constexpr (std::is_same<TT, double>::value && std::is_same<ST, double>::value)
            return lg->init(P, d_state, d_adapt, n_chains, seed, chain_offset, st);
        else
            return hipErrorInvalidValue;
    }

    /* one launch of the pair kernel (compiled instance or run-time compiled unit) */
    hipError_t launch_pair(const mm_nuts_args<TT, ST> &a, bool rtc_target, int type_mode, hipStream_t st)
    {
        if (rtc_target) {
            mm_nuts_args<TT, ST> b = a;
            const unsigned int grid = (unsigned int)((b.n_chains + 63) / 64);
            return mm_rtc_launch_nuts(user, type_mode, 2, &b, sizeof(b), grid, (size_t)MM_NUTS_RING * 64 * sizeof(double), st);
        }
        return k->run_pair(a, st);
    }

    int run(size_t n_collect, size_t n_discard, void *out, int out_is_device, int progress, void *stream_v) override
    {
        if (progress < 0 || progress > 2)
            return MMCMC_ERR_INVALID_ARG;
        /* progress == 2: run_progress' stepping with EVERY state recorded, burn-in included (what the per-chain trackers of
         * nuts.rs:486-506 are fed): out has n_collect + n_discard rows per chain */
        const size_t n_rows = progress == 2 ? n_collect + n_discard : n_collect;
        if ((uint64_t)n_rows * (uint64_t)dim >= (1ull << 30) || (uint64_t)m + n_collect + n_discard >= (1ull << 32))
            return MMCMC_ERR_SHAPE;
        DevGuard g(device);
        hipStream_t st = stream_v ? (hipStream_t)stream_v : stream;
        const size_t out_bytes = n_chains * n_rows * (size_t)dim * sizeof(TT);
        TT *d_out = nullptr;
        /* host output: a device staging buffer, released on every path out of this function */
        struct Staging {
            void *p = nullptr;
            ~Staging()
            {
                if (p)
                    (void)hipFree(p);
            }
        } staging;
        bool staged = false;
        if (out && n_rows > 0) {
            if (out_is_device) {
                d_out = (TT *)out;
            } else {
                MM_HIP(hipMalloc(&staging.p, out_bytes));
                d_out = (TT *)staging.p;
                staged = true;
            }
        }
        n_launches_run = 0;
        /* init_chain (nuts.rs:528-545) on every run() call */
        const bool use_lg = variant >= 1 && variant <= 3 && lg;
        const bool use_generic = variant == 6;
        mm_gen_nuts_args<TT, ST> ga;
        if (use_generic) {
            /* the store: mm_gen_nuts_vectors(max_depth) vectors of dim elements per chain, lane-interleaved */
            if (!d_gstore || gstore_depth < max_depth) {
                (void)hipFree(d_gstore);
                d_gstore = nullptr;
                MM_HIP(hipMalloc((void **)&d_gstore, (size_t)mm_gen_nuts_vectors(max_depth) * (size_t)dim * g_pad * sizeof(TT)));
                gstore_depth = max_depth;
            }
            ga.P = P;
            ga.kind = kind;
            ga.dim = dim;
            ga.max_depth = max_depth;
            ga.state = d_state;
            ga.adapt = d_adapt;
            ga.out = nullptr;
            ga.n_leapfrog = d_nlf;
            ga.depth_hist = d_hist;
            ga.store = d_gstore;
            ga.n_chains = n_chains;
            ga.c_pad = g_pad;
            ga.seed = seed;
            ga.chain_offset = chain_offset;
            ga.n_total = n_rows;
            ga.m0 = m;
            ga.n_pre = ga.n_rec = ga.write_initial = ga.out_t0 = 0;
            ga.n_discard = (unsigned int)n_discard;
            ga.target_accept_p = (ST)target_accept_p;
            ga.eps_tol = sizeof(ST) == 4 ? (ST)1.1920929e-7 : (ST)2.220446049250313e-16; /* T::epsilon() */
        }
        const int type_mode = std::is_same<TT, double>::value ? 2 : (std::is_same<ST, double>::value ? 0 : 1);
        const unsigned int grid64 = (unsigned int)((n_chains + 63) / 64);
        hipError_t e;
        const bool rtc_target = user && variant == 7; /* kernels of a run-time compiled unit */
        if (rtc_target) {
            mm_nuts_init_args<TT, ST> ia;
            ia.P = P;
            ia.state = d_state;
            ia.adapt = d_adapt;
            ia.n_chains = n_chains;
            ia.seed = seed;
            ia.chain_offset = chain_offset;
            ia.eps_tol = sizeof(ST) == 4 ? (ST)1.1920929e-7 : (ST)2.220446049250313e-16;
            e = mm_rtc_launch_nuts(user, type_mode, 1, &ia, sizeof(ia), grid64, 0, st);
        } else {
            e = use_lg ? init_lg(st) : use_generic ? launch_generic(ga, 1, st) : k->init(P, d_state, d_adapt, n_chains, seed, chain_offset, st);
        }
        if (e != hipSuccess)
            return (int)e;
        mm_nuts_args<TT, ST> a;
        a.P = P;
        a.state = d_state;
        a.adapt = d_adapt;
        a.out = d_out;
        a.n_leapfrog = d_nlf;
        a.depth_hist = d_hist;
        a.n_chains = n_chains;
        a.seed = seed;
        a.chain_offset = chain_offset;
        a.n_total = n_rows;
        a.m0 = m;
        a.out_t0 = 0;
        a.n_discard = (unsigned int)n_discard;
        a.max_depth = max_depth;
        a.target_accept_p = (ST)target_accept_p;
        a.stack_in_lds = stack_in_lds ? 1 : 0;
        a.async_batch = 0;
        if (const char *e = mm_tuning_env("MMCMC_NUTS_ASYNC_BATCH")) /* tuning aid; no result depends on it */
            a.async_batch = (unsigned int)atoi(e);
        a.scratch = d_scratch;
        const size_t total = n_collect + n_discard;
        if (progress == 2) {
            a.write_initial = 0;
            a.n_pre = 0;
            a.n_rec = (unsigned int)total;
        } else if (progress) {
            /* run_progress (nuts.rs:491-522): all N transitions, the last n_collect recorded */
            a.write_initial = 0;
            a.n_pre = (unsigned int)n_discard;
            a.n_rec = (unsigned int)n_collect;
        } else if (total == 0) {
            a.write_initial = 0;
            a.n_pre = a.n_rec = 0;
        } else if (n_discard == 0) {
            /* run (nuts.rs:457-471): N - 1 transitions; row 0 is the initial point */
            a.write_initial = n_collect > 0 ? 1 : 0;
            a.n_pre = 0;
            a.n_rec = (unsigned int)(n_collect > 0 ? n_collect - 1 : 0);
        } else {
            a.write_initial = 0;
            a.n_pre = (unsigned int)(n_discard - 1);
            a.n_rec = (unsigned int)n_collect;
        }
        MM_HIP(hipEventRecord(ev0, st));
        const unsigned int a_pre = a.n_pre, a_rec = a.n_rec;
        if (rtc_target) {
            /* a run-time compiled target: asynchronous lanes with the leaves in pairs (mm_nuts_pair_body; dynamic LDS = the
             * ring of uniforms, stack in the scratch area).  The unit's lanes-in-step kernel (mm_nuts_run_body) is NOT
             * launched: it gave wrong, run-to-run different samples at RosenbrockND(19) / (23) in f64 and a memory fault at
             * StandardNormal(25) in f32, while the same template compiled into the library by hipcc is correct at those
             * dimensions (tools/experiments/repro_nuts_dims.py).  Round 6 (mm_rtc.hip, at g_compiler): it was the hipRTC the
             * process had loaded -- PyTorch's bundled one, whose comgr is the 7.0.2 compiler; the system's 7.2 hipRTC emits
             * hipcc's code and is correct, and today's sources compile correctly under both.  The policy stays: the pair
             * kernel passed every comparison under every compiler, and every unit is checked against a second kernel before
             * its first use (rtc_unit_verified) */
            int run_kernel = rtc_run_kernel;
            /* measurement builds: launch the unit's lanes-in-step kernel from an ordinary handle (never inside the unit's
             * check) -- tools/experiments/rtc_lanes_in_step.py asks whether hipRTC still gets that kernel wrong */
            if (const char *rk = mm_tuning_env("MMCMC_RTC_RUN_KERNEL"))
                if (g_rtc_create_mode == 0)
                    run_kernel = atoi(rk) == 0 ? 0 : 2;
            e = run_kernel == 2
                    ? mm_rtc_launch_nuts(user, type_mode, 2, &a, sizeof(a), grid64, (size_t)MM_NUTS_RING * 64 * sizeof(double), st)
                    : mm_rtc_launch_nuts(user, type_mode, 0, &a, sizeof(a), grid64, user_lds, st); /* dynamic LDS = its output tile */
        } else if (use_generic) {
            ga.out = d_out;
            ga.n_pre = a.n_pre;
            ga.n_rec = a.n_rec;
            ga.write_initial = a.write_initial;
            e = launch_generic(ga, 0, st);
        } else {
            e = use_lg ? run_lg(a, st)
                       : (variant == 5 && k->run_pair) ? k->run_pair(a, st)
                       : (variant == 4 && k->run_async) ? k->run_async(a, st) : k->run(a, st);
        }
        if (e != hipSuccess)
            return (int)e;
        MM_HIP(hipEventRecord(ev1, st));
        m += a_pre + a_rec;
        timed = true;
        /* sampling launches of this run (the persistent scheduler of a fresh handle: a 16-transition pilot plus the
         * rest; compaction by launches: one per level and transition); kernel_ms is the time on the stream between the
         * first and the last, host gaps of multi-launch runs included */
        timing.n_launches = n_launches_run ? n_launches_run : 1;
        timing.out_bytes = d_out ? out_bytes : 0;
        timing.state_bytes = 2ull * n_chains * dim * sizeof(TT);
        timing.kernel_ms = -1.f;
        if (staged)
            MM_HIP(mm_copy_to_host(out, d_out, out_bytes, device, st));
        return MMCMC_OK;
    }

    int state(void *out) override
    {
        DevGuard g(device);
        MM_HIP(hipDeviceSynchronize());
        MM_HIP(hipMemcpy(out, d_state, n_chains * (size_t)dim * sizeof(TT), hipMemcpyDeviceToHost));
        return MMCMC_OK;
    }
    int adapt_state(double *out) override
    {
        DevGuard g(device);
        MM_HIP(hipDeviceSynchronize());
        std::vector<mm_nuts_adapt<ST>> ad(n_chains);
        MM_HIP(hipMemcpy(ad.data(), d_adapt, n_chains * sizeof(mm_nuts_adapt<ST>), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n_chains; ++i) {
            out[4 * i + 0] = (double)ad[i].epsilon;
            out[4 * i + 1] = (double)ad[i].epsilon_bar;
            out[4 * i + 2] = (double)ad[i].h_bar;
            out[4 * i + 3] = (double)ad[i].mu;
        }
        return MMCMC_OK;
    }
    int leapfrog_counts(uint64_t *out) override
    {
        DevGuard g(device);
        MM_HIP(hipDeviceSynchronize());
        MM_HIP(hipMemcpy(out, d_nlf, n_chains * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return MMCMC_OK;
    }
    int depth_histogram(uint32_t *out) override
    {
        DevGuard g(device);
        MM_HIP(hipDeviceSynchronize());
        MM_HIP(hipMemcpy(out, d_hist, (MM_NUTS_JMAX + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return MMCMC_OK;
    }
};

} // namespace

struct mmcmc_nuts {
    NutsBase *p;
};

/* A run-time compiled unit is checked once per (unit, type mode) and process before a handle relies on it: 96 chains, 5 + 5
 * transitions from a fixed start, (a) twice -- the two must be the same bits --, (b) against a SECOND, differently built
 * kernel of the same transitions, which must agree bit for bit: for a built-in target the run-time-dimension kernel
 * (variant 6, compiled into the library by hipcc); for a USER target -- whose functor exists only inside its unit -- the
 * unit's own lanes-in-step kernel (mm_nuts_run_body: another control structure around the same functor, the pair the
 * library's compiled instances are tested to agree on, variants 0 and 5) WHEN hipcc built the unit.  Why: tools/experiments/repro_nuts_dims.py found
 * hipRTC-built kernels that do not reproduce themselves or are plainly wrong (the lanes-in-step kernel at RosenbrockND(19)
 * and (23) in f64, StandardNormal(25) in f32) while hipcc builds of the same source are right; units are now built by hipcc
 * where it exists (mm_rtc.hip) and what IS launched is still checked.  A unit that fails leaves a built-in target on the
 * run-time-dimension kernel; a user target is refused.  Only real verdicts are cached: an allocation or launch error
 * inside the check fails this create and is tried again by the next one. */
extern "C" int mmcmc_nuts_create(mmcmc_nuts **out, const mmcmc_target_desc *target, const double *init, size_t n_chains,
                                 double target_accept_p, int mode, int device);
extern "C" int mmcmc_nuts_destroy(mmcmc_nuts *h);
static bool rtc_unit_verified(const void *unit, const mmcmc_target_desc *target, int mode, int device)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, bool> verdict;
    {
        std::lock_guard<std::mutex> l(mu);
        auto it = verdict.find({unit, mode});
        if (it != verdict.end())
            return it->second;
    }
    const size_t n = 96, dim = (size_t)target->dim, nc = 5, nd = 5, esz = mode == 2 ? 8 : 4;
    std::vector<double> x0(n * dim);
    for (size_t c = 0; c < n; ++c)
        for (size_t i = 0; i < dim; ++i)
            x0[c * dim + i] = 0.05 * (double)((int)((c * 7 + i * 13) % 17) - 8);
    bool errored = false;
    /* variant < 0: the handle's default (the unit's pair kernel); rtc_kernel: which run kernel of the unit */
    auto one = [&](int variant, int rtc_kernel, std::vector<unsigned char> &bytes) -> bool {
        mmcmc_nuts *t = nullptr;
        if (mmcmc_nuts_create(&t, target, x0.data(), n, 0.8, mode, device) != MMCMC_OK) {
            errored = true;
            return false;
        }
        bytes.assign(n * nc * dim * esz, 0);
        t->p->seed = 0x5eedull;
        t->p->rtc_run_kernel = rtc_kernel;
        bool ok = (variant < 0 ? t->p->rtc_unit == unit : t->p->set_variant(variant) == MMCMC_OK);
        if (ok && t->p->run(nc, nd, bytes.data(), 0, 1, nullptr) != MMCMC_OK) {
            errored = true;
            ok = false;
        }
        (void)mmcmc_nuts_destroy(t);
        return ok;
    };
    g_rtc_create_mode = 1;
    std::vector<unsigned char> a, a2, g;
    bool ok = one(-1, 2, a) && one(-1, 2, a2) && a == a2;
    /* the second opinion of a USER unit is the unit's lanes-in-step kernel -- the very kernel hipRTC was caught miscompiling
     * (wrong samples at RosenbrockND(19) / (23) f64, a device memory fault at StandardNormal(25) f32): it is launched only
     * from units hipcc built -- and (round 6, when the compiler was identified: the hipRTC PyTorch bundles, ROCm 7.0.2's;
     * the system's 7.2 emits hipcc's code) from units built by the hipRTC of a runtime at least as new as the compiler this
     * library was built with.  A user unit built by an OLDER hipRTC (no hipcc on the machine, torch imported first) keeps
     * the run-twice check alone: refusing a correct pair kernel on the word of a broken referee, or faulting the process
     * inside the check, would be worse than the exposure it closes (advisor r4). */
    int process_hip = 0, built_hip = 0;
    (void)mmcmc_rtc_compiler_info(nullptr, 0, &process_hip, &built_hip);
    const bool referee_trusted = mm_rtc_compiler((const mm_user_target *)unit) == MMCMC_RTC_COMPILER_HIPCC ||
                                 (process_hip > 0 && process_hip / 100000 >= built_hip / 100000);
    if (ok && target->kind < MM_USER_KIND_BASE)
        ok = one(6, 2, g) && a == g;
    else if (ok && referee_trusted)
        ok = one(-1, 0, g) && a == g;
    g_rtc_create_mode = 0;
    if (!errored) {
        std::lock_guard<std::mutex> l(mu);
        verdict[{unit, mode}] = ok;
    }
    return ok;
}

extern "C" {

int mmcmc_nuts_create(mmcmc_nuts **out, const mmcmc_target_desc *target, const double *init, size_t n_chains,
                      double target_accept_p, int mode, int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!target || !init || n_chains == 0 || target->dim <= 0 || mode < 0 || mode > 2 ||
        !(target_accept_p > 0.0 && target_accept_p < 1.0))
        return MMCMC_ERR_INVALID_ARG;
    if (target->kind == MMCMC_GAUSSIAN_ND && !target->matrix)
        return MMCMC_ERR_INVALID_ARG;
    if ((target->kind == MMCMC_GAUSSIAN2D || target->kind == MMCMC_DIFFABLE_GAUSSIAN2D ||
         target->kind == MMCMC_ROSENBROCK2D) &&
        target->dim != 2)
        return MMCMC_ERR_SHAPE;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= nd)
        return MMCMC_ERR_INVALID_ARG;
    NutsBase *p = nullptr;
    if (mode == 0)
        p = new (std::nothrow) Nuts<float, double>();
    else if (mode == 1)
        p = new (std::nothrow) Nuts<float, float>();
    else
        p = new (std::nothrow) Nuts<double, double>();
    if (!p)
        return (int)hipErrorOutOfMemory;
    p->device = device;
    p->mode = mode;
    p->kind = target->kind;
    p->dim = target->dim;
    p->n_chains = n_chains;
    p->target_accept_p = target_accept_p;
    int st;
    if (mode == 0)
        st = static_cast<Nuts<float, double> *>(p)->setup(target, init);
    else if (mode == 1)
        st = static_cast<Nuts<float, float> *>(p)->setup(target, init);
    else
        st = static_cast<Nuts<double, double> *>(p)->setup(target, init);
    if (st != MMCMC_OK) {
        delete p;
        return st;
    }
    mmcmc_nuts *h = new (std::nothrow) mmcmc_nuts{p};
    if (!h) {
        delete p;
        return (int)hipErrorOutOfMemory;
    }
    if (p->rtc_unit && g_rtc_create_mode == 0 && !rtc_unit_verified(p->rtc_unit, target, mode, device)) {
        /* the unit's kernels do not reproduce themselves (or the run-time-dimension kernel): not used */
        (void)mmcmc_nuts_destroy(h);
        if (target->kind >= MM_USER_KIND_BASE)
            return MMCMC_ERR_UNSUPPORTED;
        g_rtc_create_mode = 2;
        const int st2 = mmcmc_nuts_create(out, target, init, n_chains, target_accept_p, mode, device);
        g_rtc_create_mode = 0;
        return st2;
    }
    *out = h;
    return MMCMC_OK;
}

int mmcmc_nuts_seed(mmcmc_nuts *h, uint64_t seed)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->p->seed = seed;
    return MMCMC_OK;
}
int mmcmc_nuts_set_chain_offset(mmcmc_nuts *h, uint64_t off)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->p->chain_offset = off;
    return MMCMC_OK;
}
int mmcmc_nuts_set_max_depth(mmcmc_nuts *h, int max_depth)
{
    if (!h || max_depth < 1 || max_depth > MM_NUTS_JMAX)
        return MMCMC_ERR_INVALID_ARG;
    h->p->max_depth = max_depth;
    return MMCMC_OK;
}
int mmcmc_nuts_set_kernel_variant(mmcmc_nuts *h, int variant)
{
    return h ? h->p->set_variant(variant) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_kernel_variant(mmcmc_nuts *h) { return h ? h->p->variant : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_set_compaction(mmcmc_nuts *h, int first_level, int n_groups)
{
    if (!h || first_level < 0 || first_level > MM_NUTS_JMAX || n_groups < 0 || n_groups > 16)
        return MMCMC_ERR_INVALID_ARG;
    h->p->compaction_start = first_level;
    h->p->compaction_auto = false;
    h->p->compaction_groups = n_groups;
    return MMCMC_OK;
}
int mmcmc_nuts_run(mmcmc_nuts *h, size_t n_collect, size_t n_discard, void *out, int out_is_device, int progress,
                   void *stream)
{
    return h ? h->p->run(n_collect, n_discard, out, out_is_device, progress, stream) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_shape(mmcmc_nuts *h, size_t *n_chains, int *dim, int *mode, int *device)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if (n_chains)
        *n_chains = h->p->n_chains;
    if (dim)
        *dim = h->p->dim;
    if (mode)
        *mode = h->p->mode;
    if (device)
        *device = h->p->device;
    return MMCMC_OK;
}
int mmcmc_nuts_state(mmcmc_nuts *h, void *out) { return (h && out) ? h->p->state(out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_nuts_adapt_state(mmcmc_nuts *h, double *out)
{
    return (h && out) ? h->p->adapt_state(out) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_leapfrog_counts(mmcmc_nuts *h, uint64_t *out)
{
    return (h && out) ? h->p->leapfrog_counts(out) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_depth_histogram(mmcmc_nuts *h, uint32_t *out)
{
    return (h && out) ? h->p->depth_histogram(out) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_nuts_sync(mmcmc_nuts *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->p->device);
    MM_HIP(hipStreamSynchronize(h->p->stream));
    return MMCMC_OK;
}
int mmcmc_nuts_timing(mmcmc_nuts *h, mmcmc_timing *t)
{
    if (!h || !t)
        return MMCMC_ERR_INVALID_ARG;
    if (!h->p->timed)
        return MMCMC_ERR_STATE;
    DevGuard g(h->p->device);
    MM_HIP(hipEventSynchronize(h->p->ev1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, h->p->ev0, h->p->ev1));
    h->p->timing.kernel_ms = ms;
    *t = h->p->timing;
    return MMCMC_OK;
}
int mmcmc_nuts_destroy(mmcmc_nuts *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    delete h->p;
    delete h;
    return MMCMC_OK;
}

} /* extern "C" */
