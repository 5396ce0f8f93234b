/*
 * mm_api.hip -- implementation of the C ABI in include/mmcmc.h (MH, HMC, densities, init).
 *
 * Host logic only: handles, argument checking, kernel lookup, launches, timing.  All arithmetic of the hot path
 * lives in mm_samplers.h / mm_targets.h / mm_rng.h and runs on the device; nothing here computes a transition
 * on the CPU and nothing here touches oracle/.
 */
#include "../../include/mmcmc.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <type_traits>
#include <tuple>
#include <vector>

#include "mm_generic.h"
#include "mm_hmc_lg.h"
#include "mm_hostcopy.h"
#include "mm_kernels.h"
#include "mm_host_rng.h"
#include "mm_params.h"
#include "mm_wide.h"
#include "mm_tuning.h"
#include "mm_rtc.h"

size_t mm_split_lds_bytes_f32(int dim, int mh); /* mm_inst_f32.hip */

#define MM_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return (int)_e;                                                                                       \
    } while (0)

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= n)
        return MMCMC_ERR_INVALID_ARG;
    return MMCMC_OK;
}

template <class T> const mm_kernel_entry<T> *table(int *n);
template <> const mm_kernel_entry<float> *table<float>(int *n) { return mm_kernel_table_f32(n); }
template <> const mm_kernel_entry<double> *table<double>(int *n) { return mm_kernel_table_f64(n); }
template <class T> const mm_noise_entry<T> *noise_table(int *n);
template <> const mm_noise_entry<float> *noise_table<float>(int *n) { return mm_noise_table_f32(n); }
template <> const mm_noise_entry<double> *noise_table<double>(int *n) { return mm_noise_table_f64(n); }

template <class T> const mm_kernel_entry<T> *find_kernel(int kind, int dim)
{
    int n = 0;
    const mm_kernel_entry<T> *t = table<T>(&n);
    for (int i = 0; i < n; ++i)
        if (t[i].kind == kind && t[i].dim == dim)
            return &t[i];
    return nullptr;
}

/* set by mmcmc_mh_create while it creates the sampler of a target + proposal model */
static thread_local bool g_allow_model = false;

int validate_target(const mmcmc_target_desc *t)
{
    if (!t || t->dim <= 0)
        return MMCMC_ERR_INVALID_ARG;
    switch (t->kind) {
    case MMCMC_GAUSSIAN2D:
    case MMCMC_DIFFABLE_GAUSSIAN2D:
    case MMCMC_ROSENBROCK2D:
        if (t->dim != 2)
            return MMCMC_ERR_SHAPE;
        break;
    case MMCMC_ISOTROPIC_GAUSSIAN:
        if (!(t->params[0] > 0.0))
            return MMCMC_ERR_INVALID_ARG;
        break;
    case MMCMC_ROSENBROCK_ND:
    case MMCMC_STANDARD_NORMAL:
        break;
    case MMCMC_GAUSSIAN_ND:
        if (!t->matrix)
            return MMCMC_ERR_INVALID_ARG;
        break;
    default: {
        /* a kind handed out by mmcmc_target_register_source (mm_rtc.hip) */
        const mm_user_target *u = mm_rtc_find(t->kind);
        if (!u)
            return MMCMC_ERR_UNSUPPORTED;
        if (mm_rtc_is_internal(u))
            return MMCMC_ERR_UNSUPPORTED; /* a unit the library built for itself is not a kind a caller may name */
        if (mm_rtc_dim(u) != t->dim)
            return MMCMC_ERR_SHAPE;
        /* a target + proposal model (mmcmc_proposal_register_source) is reached through mmcmc_mh_create's proposal
         * description only, which checks its base target itself */
        if (mm_rtc_is_model(u) && !g_allow_model)
            return MMCMC_ERR_UNSUPPORTED;
        if (mm_rtc_is_discrete(u))
            return MMCMC_ERR_UNSUPPORTED; /* an integer-state model belongs to mmcmc_mh_discrete_create */
        if (mm_rtc_is_model(u) && mm_rtc_base_kind(u) == MMCMC_GAUSSIAN_ND && !t->matrix)
            return MMCMC_ERR_INVALID_ARG;
        break;
    }
    }
    return MMCMC_OK;
}

/* device-side parameter block in element type T (mm_params.h) + the GaussianND matrix uploaded to HBM */
template <class T> int make_params(const mmcmc_target_desc *t, mm_tparams<T> *P, T **d_mat)
{
    *d_mat = nullptr;
    /* a model over a BUILT-IN target derives its parameter block like that target (Sigma^-1, 1 / sigma^2, ...) */
    const int base = t->kind >= MM_USER_KIND_BASE ? mm_rtc_base_kind(mm_rtc_find(t->kind)) : t->kind;
    if (t->kind >= MM_USER_KIND_BASE && !(base >= 0 && base < MM_USER_KIND_BASE)) {
        /* user target: the description's parameters as they are; `matrix`, if given, is dim x dim */
        std::memset(P, 0, sizeof *P);
        for (int i = 0; i < 8; ++i)
            P->p[i] = (T)t->params[i];
    } else if (mm_fill_params<T>(base, t->params, P) != 0)
        return MMCMC_ERR_INVALID_ARG;
    if (t->kind == MMCMC_GAUSSIAN_ND || (t->kind >= MM_USER_KIND_BASE && t->matrix)) {
        size_t n = (size_t)t->dim * t->dim;
        std::vector<T> h(n);
        for (size_t i = 0; i < n; ++i)
            h[i] = (T)t->matrix[i];
        MM_HIP(hipMalloc((void **)d_mat, n * sizeof(T)));
        MM_HIP(hipMemcpy(*d_mat, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        P->mat = *d_mat;
    }
    return MMCMC_OK;
}

/* one sampler handle; MH and HMC differ only in the kernel entry and two scalars */
struct Sampler {
    int sampler = 0; /* MM_SAMPLER_MH / MM_SAMPLER_HMC */
    int dtype = MMCMC_F32;
    int device = 0;
    int kind = 0, dim = 0;
    size_t n_chains = 0;
    double scale = 0; /* proposal std or step size */
    int n_leapfrog = 0;
    uint64_t seed = 0, chain_offset = 0;
    uint64_t iter = 0;
    uint32_t iters_per_launch = 0;
    int variant = 2; /* mm_run_kernel PIPE: 0 plain; 2 (default) noise of two iterations packed + pipelined (PIPE = 1,
                        noise of t+1 pipelined, measured equal to 2 and is no longer instantiated; 1 selects 2);
                        3 = lane-group / MFMA kernel (mm_hmc_lg.h): HMC, f64, GaussianND of dim 16 or 32;
                        5 = noise waves + transition waves (mm_split_kernels.h), the default for f32 up to dim 8 */
    bool lg_ok = false;
    const mm_user_target *user = nullptr; /* run-time compiled target (mm_rtc.hip), variant 7 */
    bool generic = false;      /* no fixed-dimension kernel: the run-time-dimension path (mm_generic.h), variant 6 */
    bool generic_ok = false;   /* the target kind has a run-time-dimension form */
    bool wide_ok = false;      /* HMC with one chain per workgroup (mm_wide.hip): a huge dimension, variant 8 */
    void *d_gscratch = nullptr; /* its HBM store when the chain vectors do not fit LDS */
    void *d_stage = nullptr;    /* device staging of a sample that goes to the host (sampler_run) */
    size_t stage_cap = 0;
    size_t c_pad = 0;
    unsigned int block = 64;
    void *d_state = nullptr;
    void *d_mat = nullptr;
    unsigned long long *d_accept = nullptr;
    unsigned long long *d_accept_total = nullptr;
    mm_tparams<float> Pf;
    mm_tparams<double> Pd;
    const mm_kernel_entry<float> *kf = nullptr;
    const mm_kernel_entry<double> *kd = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool want_accept = true; /* this run(): per-chain accept counts requested */
    bool timing_enabled = true;
    mmcmc_timing timing{};

    size_t esize() const { return dtype == MMCMC_F32 ? 4 : 8; }
};

int sampler_create(Sampler **out, int sampler, const mmcmc_target_desc *target, double scale, int n_leapfrog,
                   const void *init, size_t n_chains, int dtype, int device);
int sampler_destroy(Sampler *s);
int sampler_run(Sampler *s, size_t n_collect, size_t n_discard, void *out, int out_is_device, uint64_t *accept_counts,
                void *stream_v, size_t n_total_rows = 0, size_t row0 = 0);

thread_local bool g_in_unit_check = false;

/* A built-in target's run-time compiled MH / HMC unit (mm_rtc_builtin, variant 7) is checked once per (unit, sampler,
 * dtype) and process before a handle relies on it, like the NUTS units (mm_nuts_api.hip): 96 chains, 5 + 5 transitions from
 * a fixed start, run twice -- the same bits -- and against the run-time-dimension kernel (variant 6), bit for bit.  The
 * compiler that builds these units was caught miscompiling a sibling kernel, and a miscompiled unit would
 * otherwise return wrong samples silently.  Only real verdicts are cached: a failed allocation or launch inside the check
 * leaves this handle on variant 6 and is tried again by the next one. */
bool builtin_unit_verified(const mm_user_target *unit, int sampler, const mmcmc_target_desc *target, int dtype, int device)
{
    static std::mutex mu;
    static std::map<std::tuple<const void *, int, int>, bool> verdict;
    const auto key = std::make_tuple((const void *)unit, sampler, dtype);
    {
        std::lock_guard<std::mutex> l(mu);
        auto it = verdict.find(key);
        if (it != verdict.end())
            return it->second;
    }
    const size_t n = 96, dim = (size_t)target->dim, nc = 5, nd = 5, esz = dtype == MMCMC_F32 ? 4 : 8;
    std::vector<unsigned char> x0(n * dim * esz);
    for (size_t c = 0; c < n; ++c)
        for (size_t i = 0; i < dim; ++i) {
            const double v = 0.05 * (double)((int)((c * 7 + i * 13) % 17) - 8);
            if (dtype == MMCMC_F32)
                ((float *)x0.data())[c * dim + i] = (float)v;
            else
                ((double *)x0.data())[c * dim + i] = v;
        }
    bool errored = false;
    auto one = [&](int variant, std::vector<unsigned char> &bytes) -> bool {
        Sampler *t = nullptr;
        /* modest fixed scales: the check is about the kernels, not about the caller's tuning */
        if (sampler_create(&t, sampler, target, sampler == MM_SAMPLER_HMC ? 0.01 : 0.25, sampler == MM_SAMPLER_HMC ? 3 : 0,
                           x0.data(), n, dtype, device) != MMCMC_OK) {
            errored = true;
            return false;
        }
        bytes.assign(n * nc * dim * esz, 0);
        t->seed = 0x5eedull;
        bool ok = t->user == unit; /* the nested handle must be on this very unit */
        if (ok)
            t->variant = variant;
        if (ok && sampler_run(t, nc, nd, bytes.data(), 0, nullptr, nullptr, 0, 0) != MMCMC_OK) {
            errored = true;
            ok = false;
        }
        (void)sampler_destroy(t);
        return ok;
    };
    g_in_unit_check = true;
    std::vector<unsigned char> a, a2, g;
    const bool ok = one(7, a) && one(7, a2) && one(6, g) && a == a2 && a == g;
    g_in_unit_check = false;
    if (!errored) {
        std::lock_guard<std::mutex> l(mu);
        verdict[key] = ok;
    }
    return ok;
}

int sampler_create(Sampler **out, int sampler, const mmcmc_target_desc *target, double scale, int n_leapfrog,
                   const void *init, size_t n_chains, int dtype, int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!init || n_chains == 0 || (dtype != MMCMC_F32 && dtype != MMCMC_F64) || !(scale > 0.0) ||
        !std::isfinite(scale) || n_leapfrog < 0)
        return MMCMC_ERR_INVALID_ARG;
    int st = validate_target(target);
    if (st != MMCMC_OK)
        return st;
    st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    Sampler *s = new (std::nothrow) Sampler();
    if (!s)
        return (int)hipErrorOutOfMemory;
    s->sampler = sampler;
    s->dtype = dtype;
    s->device = device;
    s->kind = target->kind;
    s->dim = target->dim;
    s->n_chains = n_chains;
    s->scale = scale;
    s->n_leapfrog = n_leapfrog;
    if (dtype == MMCMC_F32)
        s->kf = find_kernel<float>(s->kind, s->dim);
    else
        s->kd = find_kernel<double>(s->kind, s->dim);
    s->generic_ok = mm_generic_kind_ok(s->kind);
    s->user = s->kind >= MM_USER_KIND_BASE ? mm_rtc_find(s->kind) : nullptr;
    if (s->user) {
        s->variant = 7;
    } else if (!s->kf && !s->kd) {
        /* a dimension without a register-resident kernel: the run-time-dimension path */
        if (!s->generic_ok) {
            delete s;
            return MMCMC_ERR_UNSUPPORTED;
        }
        s->generic = true;
        s->variant = 6;
        /* up to dimension 32 the target's own functor is compiled into the register-resident skeletons on first use
         * (hipRTC, mm_rtc_builtin): variant 7 then, with the run-time-D kernel (6) still selectable */
        if (s->dim <= 32) {
            DeviceGuard gb(device);
            s->user = mm_rtc_builtin(s->kind, s->dim);
            if (s->user)
                s->variant = 7;
        }
    }
    /* dense Gaussian at dim 16 / 32 under HMC (f64 and f32): the lane-group / MFMA kernels, and the default there */
    s->lg_ok = sampler == MM_SAMPLER_HMC && s->kind == MMCMC_GAUSSIAN_ND && (s->dim == 16 || s->dim == 32);
    /* at dim 32 the paired / pipelined form holds four noise vectors next to the state and spills: the plain form
     * is twice as fast there (RosenbrockND(32) f32: 1.0 ms vs 2.2 ms for run(100, 20) of 65 536 chains) */
    if (s->dim > 16 && !s->generic && !s->user)
        s->variant = 0;
    /* f32 up to dim 8: noise waves + transition waves, four waves per SIMD (mm_split_kernels.h; config 3: 0.192 ms
     * against 0.268 ms for variant 2) */
    if (dtype == MMCMC_F32 && s->kf && s->kf->run_mh_split)
        s->variant = 5;
    if (s->lg_ok)
        s->variant = 3;
    if (s->user && !s->generic)
        s->variant = 7;
    /* few chains of a huge dimension (hmc.rs:882-916: 6 x 10 000): the coordinates of a chain across a workgroup */
    s->wide_ok = sampler == MM_SAMPLER_HMC && !s->user && mm_wide_kind_ok(s->kind) && s->dim >= 4 && s->dim <= MM_WIDE_MAX_DIM;
    if (s->wide_ok && s->generic && s->dim >= 128 && n_chains < 1024)
        s->variant = 8;
    DeviceGuard g(device);
    auto fail = [&](int code) {
        if (s->d_state)
            (void)hipFree(s->d_state);
        if (s->d_mat)
            (void)hipFree(s->d_mat);
        if (s->d_accept)
            (void)hipFree(s->d_accept);
        if (s->d_accept_total)
            (void)hipFree(s->d_accept_total);
        if (s->d_gscratch)
            (void)hipFree(s->d_gscratch);
        if (s->stream)
            (void)hipStreamDestroy(s->stream);
        if (s->ev0)
            (void)hipEventDestroy(s->ev0);
        if (s->ev1)
            (void)hipEventDestroy(s->ev1);
        delete s;
        return code;
    };
    if (dtype == MMCMC_F32) {
        float *m = nullptr;
        st = make_params<float>(target, &s->Pf, &m);
        s->d_mat = m;
    } else {
        double *m = nullptr;
        st = make_params<double>(target, &s->Pd, &m);
        s->d_mat = m;
    }
    if (st != MMCMC_OK)
        return fail(st);
    size_t bytes = n_chains * (size_t)s->dim * s->esize();
    hipError_t e;
    if ((e = hipMalloc(&s->d_state, bytes)) != hipSuccess)
        return fail((int)e);
    if ((e = hipMemcpy(s->d_state, init, bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return fail((int)e);
    s->c_pad = (n_chains + 63) / 64 * 64;
    if (s->generic_ok) {
        /* the run-time-dimension path (default without a fixed-dimension kernel, selectable as variant 6 otherwise)
         * keeps a chain's vectors in LDS when they fit, else in this lane-interleaved HBM store */
        const size_t lds = dtype == MMCMC_F32 ? mm_generic_store_bytes<float>(sampler, s->dim) : mm_generic_store_bytes<double>(sampler, s->dim);
        if (lds > MM_GENERIC_LDS_MAX && s->generic) {
            const size_t nvec = sampler == MM_SAMPLER_HMC ? MM_GV_HMC : MM_GV_MH;
            if ((e = hipMalloc(&s->d_gscratch, nvec * (size_t)s->dim * s->c_pad * s->esize())) != hipSuccess)
                return fail((int)e);
        }
    }
    if ((e = hipMalloc((void **)&s->d_accept, n_chains * sizeof(unsigned long long))) != hipSuccess)
        return fail((int)e);
    if ((e = hipMalloc((void **)&s->d_accept_total, sizeof(unsigned long long))) != hipSuccess)
        return fail((int)e);
    if ((e = hipMemset(s->d_accept_total, 0, sizeof(unsigned long long))) != hipSuccess)
        return fail((int)e);
    if ((e = hipStreamCreateWithFlags(&s->stream, hipStreamDefault)) != hipSuccess)
        return fail((int)e);
    if ((e = hipEventCreate(&s->ev0)) != hipSuccess)
        return fail((int)e);
    if ((e = hipEventCreate(&s->ev1)) != hipSuccess)
        return fail((int)e);
    if (s->user && s->generic && !g_in_unit_check && !builtin_unit_verified(s->user, sampler, target, dtype, device)) {
        s->user = nullptr; /* the run-time-dimension kernel only (set_kernel_variant(7) is refused then) */
        s->variant = 6;
    }
    *out = s;
    return MMCMC_OK;
}

int sampler_destroy(Sampler *s)
{
    if (!s)
        return MMCMC_ERR_INVALID_ARG;
    DeviceGuard g(s->device);
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(s->d_state);
    if (s->d_mat)
        (void)hipFree(s->d_mat);
    (void)hipFree(s->d_accept);
    (void)hipFree(s->d_accept_total);
    if (s->d_gscratch)
        (void)hipFree(s->d_gscratch);
    if (s->d_stage)
        (void)hipFree(s->d_stage);
    (void)hipEventDestroy(s->ev0);
    (void)hipEventDestroy(s->ev1);
    (void)hipStreamDestroy(s->stream);
    delete s;
    return MMCMC_OK;
}

/* mm_tile<T, D>::lds_bytes_table + lds_bytes_per_wave for a dimension known only at run time (one wave per workgroup):
 * must mirror mm_kernels.h (mm_tile_default / mm_tile_t) */
static size_t mm_tile_lds_bytes_rt(size_t esz, int D)
{
    const int target = esz == 4 ? 96 : 48;
    const int tile_t = (target / D) >= 2 ? ((target / D) & ~1) : 1;
    const int run = tile_t * D, epl = (int)(16 / esz);
    const int stride = (run % epl) == 0 ? run + epl : (run | 1);
    return (esz == 4 ? (size_t)MM_NOISE_TABLE_BYTES : 0) + (size_t)64 * stride * esz;
}

template <class T>
int launch_range(Sampler *s, const mm_kernel_entry<T> *k, const mm_tparams<T> &P, T *d_out, size_t n_total,
                 uint32_t n_discard, uint32_t n_collect, uint32_t out_t0, hipStream_t stream)
{
    mm_run_args<T> a;
    a.P = P;
    a.scale = (T)s->scale;
    a.n_leapfrog = s->n_leapfrog;
    a.state = (T *)s->d_state;
    a.out = d_out;
    a.accept = s->want_accept ? s->d_accept : nullptr;
    a.accept_total = s->d_accept_total;
    a.n_chains = s->n_chains;
    a.seed = s->seed;
    a.chain_offset = s->chain_offset;
    a.iter0 = (unsigned int)s->iter;
    a.n_discard = n_discard;
    a.n_collect = n_collect;
    a.out_t0 = out_t0;
    a.n_total = n_total;
    unsigned int grid = (unsigned int)((s->n_chains + s->block - 1) / s->block);
    hipError_t e;
    const bool mh = s->sampler == MM_SAMPLER_MH, l10 = s->n_leapfrog == 10;
    if (s->variant == 7 && s->user) {
        /* the same skeleton (mm_run_kernel_body, PIPE = 2) around the user's functor, from the run-time compiled module */
        /* f32 up to dim 8: the split-role skeleton (four waves per SIMD) where the module has it; else PIPE = 2, one wave per SIMD */
        e = hipErrorNotFound;
        const char *us = mm_tuning_env("MMCMC_USER_SPLIT"); /* measurement aid: "0" keeps the one-wave-per-SIMD skeleton */
        if (std::is_same<T, float>::value && s->dim <= 8 && !(us && us[0] == '0')) {
            const size_t lds_split = mm_split_lds_bytes_f32(s->dim, mh ? 1 : 0);
            if (lds_split)
                e = mm_rtc_launch_run_split(s->user, mh ? 0 : (l10 ? 2 : 1), &a, sizeof(a), (unsigned int)((s->n_chains + 255) / 256), lds_split, stream);
        }
        if (e != hipSuccess && e != hipErrorNotFound) {
            (void)hipGetLastError(); /* a runtime that refuses the launch (e.g. the dynamic LDS size): the one-wave skeleton */
            e = hipErrorNotFound;
        }
        if (e == hipErrorNotFound) {
            const size_t lds = mm_tile_lds_bytes_rt(sizeof(T), s->dim);
            e = mm_rtc_launch_run(s->user, mh ? 0 : 1, std::is_same<T, float>::value ? 0 : 1, &a, sizeof(a), grid, s->block, lds, stream);
        }
    } else if (s->variant == 8 && s->wide_ok) {
        mm_wide_args<T> q;
        q.P = P;
        q.kind = s->kind;
        q.dim = s->dim;
        q.eps = (T)s->scale;
        q.n_leapfrog = s->n_leapfrog;
        q.state = (T *)s->d_state;
        q.out = d_out;
        q.accept = a.accept;
        q.accept_total = a.accept_total;
        q.n_chains = a.n_chains;
        q.seed = a.seed;
        q.chain_offset = a.chain_offset;
        q.n_total = n_total;
        q.iter0 = a.iter0;
        q.n_discard = n_discard;
        q.n_collect = n_collect;
        q.out_t0 = out_t0;
        if constexpr (std::is_same<T, float>::value)
            e = mm_launch_hmc_wide_f32(q, stream);
        else
            e = mm_launch_hmc_wide_f64(q, stream);
    } else if (s->variant == 6) {
        mm_gen_args<T> q;
        q.P = P;
        q.kind = s->kind;
        q.dim = s->dim;
        q.sampler = mh ? 0 : 1;
        q.scale = (T)s->scale;
        q.n_leapfrog = s->n_leapfrog;
        q.state = (T *)s->d_state;
        q.out = d_out;
        q.accept = a.accept;
        q.accept_total = a.accept_total;
        q.scratch = (T *)s->d_gscratch;
        q.n_chains = a.n_chains;
        q.c_pad = s->c_pad;
        q.seed = a.seed;
        q.chain_offset = a.chain_offset;
        q.n_total = n_total;
        q.iter0 = a.iter0;
        q.n_discard = n_discard;
        q.n_collect = n_collect;
        q.out_t0 = out_t0;
        if constexpr (std::is_same<T, float>::value)
            e = mm_launch_run_generic_f32(q, stream);
        else
            e = mm_launch_run_generic_f64(q, stream);
    } else if (s->variant == 3 && s->lg_ok) {
        if constexpr (std::is_same<T, double>::value) {
            mm_hmc_lg_args q;
            q.mat = (const double *)s->d_mat;
            q.state = (double *)s->d_state;
            q.out = (double *)d_out;
            q.accept = a.accept;
            q.accept_total = a.accept_total;
            q.n_chains = a.n_chains;
            q.seed = a.seed;
            q.chain_offset = a.chain_offset;
            q.n_total = n_total;
            q.iter0 = a.iter0;
            q.n_discard = n_discard;
            q.n_collect = n_collect;
            q.out_t0 = out_t0;
            q.eps = s->scale;
            q.n_leapfrog = s->n_leapfrog;
            e = mm_launch_hmc_lg(s->dim, q, stream);
        } else {
            mm_hmc_lg32_args q;
            q.mat = (const float *)s->d_mat;
            q.state = (float *)s->d_state;
            q.out = (float *)d_out;
            q.accept = a.accept;
            q.accept_total = a.accept_total;
            q.n_chains = a.n_chains;
            q.seed = a.seed;
            q.chain_offset = a.chain_offset;
            q.n_total = n_total;
            q.iter0 = a.iter0;
            q.n_discard = n_discard;
            q.n_collect = n_collect;
            q.out_t0 = out_t0;
            q.eps = s->scale;
            q.n_leapfrog = s->n_leapfrog;
            e = mm_launch_hmc_lg32(s->dim, q, stream);
        }
    } else if (s->variant == 5) {
        /* noise waves + transition waves (mm_split_kernels.h); the setter has checked that the instance exists */
        e = mh ? k->run_mh_split(a, stream) : (l10 ? k->run_hmc_split10(a, stream) : k->run_hmc_split(a, stream));
    } else if (s->variant != 0)
        e = mh ? k->run_mh_pp(a, grid, s->block, stream)
               : (l10 ? k->run_hmc_pp10(a, grid, s->block, stream) : k->run_hmc_pp(a, grid, s->block, stream));
    else
        e = mh ? k->run_mh(a, grid, s->block, stream) : k->run_hmc(a, grid, s->block, stream);
    if (e != hipSuccess)
        return (int)e;
    s->iter += (uint64_t)n_discard + n_collect;
    return MMCMC_OK;
}

/* n_total_rows / row0: the collected rows go to rows [row0, row0 + n_collect) of out [n_chains, n_total_rows, dim] (device
 * memory only when n_total_rows != n_collect): how run_progress fills one sample by several launches (mm_progress.hip) */
int sampler_run(Sampler *s, size_t n_collect, size_t n_discard, void *out, int out_is_device,
                uint64_t *accept_counts, void *stream_v, size_t n_total_rows, size_t row0)
{
    if (!s)
        return MMCMC_ERR_INVALID_ARG;
    if (n_collect + n_discard == 0)
        return MMCMC_OK;
    if (n_total_rows == 0)
        n_total_rows = n_collect;
    if (row0 + n_collect > n_total_rows || ((n_total_rows != n_collect || row0 != 0) && out && !out_is_device))
        return MMCMC_ERR_INVALID_ARG;
    if ((uint64_t)n_total_rows * (uint64_t)s->dim >= (1ull << 30) || s->iter + n_collect + n_discard >= (1ull << 32))
        return MMCMC_ERR_SHAPE;
    DeviceGuard g(s->device);
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : s->stream;
    const size_t out_bytes = s->n_chains * n_collect * (size_t)s->dim * s->esize();
    void *d_out = nullptr;
    /* host output: the kernels write a device staging buffer the handle keeps (grown on demand, freed with the handle: a
     * hipMalloc + hipFree of the sample's size per call cost more than the kernel), mm_copy_to_host brings it over */
    bool staged = false;
    if (out && n_collect > 0) {
        if (out_is_device) {
            d_out = out;
        } else {
            if (out_bytes > s->stage_cap) {
                if (s->d_stage) {
                    MM_HIP(hipStreamSynchronize(stream));
                    (void)hipFree(s->d_stage);
                }
                s->d_stage = nullptr;
                s->stage_cap = 0;
                MM_HIP(hipMalloc(&s->d_stage, out_bytes));
                s->stage_cap = out_bytes;
            }
            d_out = s->d_stage;
            staged = true;
        }
    }
    /* per-chain accept counts only when asked for: otherwise neither the clearing fill nor the kernel's stores */
    s->want_accept = accept_counts != nullptr;
    if (s->want_accept)
        MM_HIP(hipMemsetAsync(s->d_accept, 0, s->n_chains * sizeof(unsigned long long), stream));

    /* split the run into launches of at most iters_per_launch transitions (0 = one launch) */
    uint64_t remaining_discard = n_discard, remaining_collect = n_collect, t0 = row0;
    const uint64_t cap = s->iters_per_launch ? s->iters_per_launch : (n_discard + n_collect);
    uint32_t launches = 0;
    if (s->timing_enabled)
        MM_HIP(hipEventRecord(s->ev0, stream));
    while (remaining_discard + remaining_collect > 0) {
        uint32_t nd = (uint32_t)std::min<uint64_t>(remaining_discard, cap);
        uint32_t nc = (uint32_t)std::min<uint64_t>(remaining_collect, cap - nd);
        int st;
        if (s->dtype == MMCMC_F32)
            st = launch_range<float>(s, s->kf, s->Pf, (float *)d_out, n_total_rows, nd, nc, (uint32_t)t0, stream);
        else
            st = launch_range<double>(s, s->kd, s->Pd, (double *)d_out, n_total_rows, nd, nc, (uint32_t)t0, stream);
        if (st != MMCMC_OK)
            return st;
        remaining_discard -= nd;
        remaining_collect -= nc;
        t0 += nc;
        ++launches;
    }
    if (s->timing_enabled)
        MM_HIP(hipEventRecord(s->ev1, stream));
    s->timed = s->timing_enabled;
    s->timing.n_launches = launches;
    s->timing.out_bytes = d_out ? out_bytes : 0;
    s->timing.state_bytes = (uint64_t)launches * 2ull * s->n_chains * s->dim * s->esize();
    s->timing.kernel_ms = -1.0f;

    if (staged)
        MM_HIP(mm_copy_to_host(out, d_out, out_bytes, s->device, stream));
    if (accept_counts) {
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64");
        MM_HIP(hipMemcpyAsync(accept_counts, s->d_accept, s->n_chains * sizeof(uint64_t), hipMemcpyDeviceToHost,
                              stream));
        MM_HIP(hipStreamSynchronize(stream));
    }
    return MMCMC_OK;
}

int sampler_state(Sampler *s, void *out)
{
    if (!s || !out)
        return MMCMC_ERR_INVALID_ARG;
    DeviceGuard g(s->device);
    MM_HIP(hipStreamSynchronize(s->stream));
    MM_HIP(hipDeviceSynchronize());
    MM_HIP(hipMemcpy(out, s->d_state, s->n_chains * (size_t)s->dim * s->esize(), hipMemcpyDeviceToHost));
    return MMCMC_OK;
}

int sampler_sync(Sampler *s)
{
    if (!s)
        return MMCMC_ERR_INVALID_ARG;
    DeviceGuard g(s->device);
    if (s->timed)
        MM_HIP(hipEventSynchronize(s->ev1));
    MM_HIP(hipStreamSynchronize(s->stream));
    return MMCMC_OK;
}

int sampler_timing(Sampler *s, mmcmc_timing *t)
{
    if (!s || !t)
        return MMCMC_ERR_INVALID_ARG;
    if (!s->timed)
        return MMCMC_ERR_STATE; /* no timed run yet */
    DeviceGuard g(s->device);
    MM_HIP(hipEventSynchronize(s->ev1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->timing.kernel_ms = ms;
    *t = s->timing;
    return MMCMC_OK;
}

template <class T>
int logp_grad_batch_t(const mmcmc_target_desc *target, const void *x, size_t n, void *logp, void *grad)
{
    const mm_user_target *user = target->kind >= MM_USER_KIND_BASE ? mm_rtc_find(target->kind) : nullptr;
    const mm_kernel_entry<T> *k = user ? nullptr : find_kernel<T>(target->kind, target->dim);
    if (!k && !user && !mm_generic_kind_ok(target->kind))
        return MMCMC_ERR_UNSUPPORTED;
    mm_tparams<T> P;
    T *d_mat = nullptr;
    int st = make_params<T>(target, &P, &d_mat);
    if (st != MMCMC_OK)
        return st;
    size_t d = (size_t)target->dim;
    T *dx = nullptr, *dl = nullptr, *dg = nullptr, *dscr = nullptr;
    int rc = MMCMC_OK;
    hipError_t e;
    do {
        if ((e = hipMalloc((void **)&dx, n * d * sizeof(T))) != hipSuccess)
            break;
        if ((e = hipMalloc((void **)&dl, n * sizeof(T))) != hipSuccess)
            break;
        if (grad && (e = hipMalloc((void **)&dg, n * d * sizeof(T))) != hipSuccess)
            break;
        if ((e = hipMemcpy(dx, x, n * d * sizeof(T), hipMemcpyHostToDevice)) != hipSuccess)
            break;
        if (user) {
            struct {
                mm_tparams<T> P;
                const T *x;
                T *logp, *grad;
                unsigned long long n;
            } ua{P, dx, dl, dg, (unsigned long long)n};
            if ((e = mm_rtc_launch_logp_grad(user, std::is_same<T, float>::value ? 0 : 1, &ua, sizeof(ua), n, nullptr)) != hipSuccess)
                break;
        } else if (k) {
            if ((e = k->logp_grad(P, dx, dl, dg, (unsigned long long)n, nullptr)) != hipSuccess)
                break;
        } else {
            /* run-time dimension: position and gradient of row i in a lane-interleaved scratch store */
            const unsigned long long n_pad = (n + 63) / 64 * 64;
            if ((e = hipMalloc((void **)&dscr, 2 * d * n_pad * sizeof(T))) != hipSuccess)
                break;
            if constexpr (std::is_same<T, float>::value)
                e = mm_launch_logp_grad_generic_f32(P, target->kind, target->dim, dx, dl, dg, dscr, n, n_pad, nullptr);
            else
                e = mm_launch_logp_grad_generic_f64(P, target->kind, target->dim, dx, dl, dg, dscr, n, n_pad, nullptr);
            if (e != hipSuccess)
                break;
        }
        if ((e = hipDeviceSynchronize()) != hipSuccess)
            break;
        if ((e = hipMemcpy(logp, dl, n * sizeof(T), hipMemcpyDeviceToHost)) != hipSuccess)
            break;
        if (grad && (e = hipMemcpy(grad, dg, n * d * sizeof(T), hipMemcpyDeviceToHost)) != hipSuccess)
            break;
    } while (0);
    if (e != hipSuccess)
        rc = (int)e;
    (void)hipFree(dx);
    (void)hipFree(dl);
    if (dg)
        (void)hipFree(dg);
    if (dscr)
        (void)hipFree(dscr);
    if (d_mat)
        (void)hipFree(d_mat);
    return rc;
}

template <class T>
int draw_noise_t(uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n, int dim, void *z, void *u, int mh = 0)
{
    int nn = 0;
    const mm_noise_entry<T> *t = noise_table<T>(&nn);
    const mm_noise_entry<T> *k = nullptr;
    /* mh: the MH sampler's paired stream (f32, dim <= 2: mm_rng.h) through the run-time-dimension kernel; the same noise as
     * mmcmc_draw_noise's everywhere else */
    const bool paired = mh && std::is_same<T, float>::value && dim <= 2;
    for (int i = 0; i < nn && !paired; ++i)
        if (t[i].dim == dim)
            k = &t[i];
    if (!k && dim <= 0)
        return MMCMC_ERR_INVALID_ARG;
    T *dz = nullptr, *du = nullptr;
    hipError_t e;
    do {
        if ((e = hipMalloc((void **)&dz, n * (size_t)dim * sizeof(T))) != hipSuccess)
            break;
        if ((e = hipMalloc((void **)&du, n * sizeof(T))) != hipSuccess)
            break;
        if (k)
            e = k->noise(seed, chain_offset, iteration, (unsigned long long)n, dz, du, nullptr);
        else if constexpr (std::is_same<T, float>::value)
            e = mm_launch_noise_generic_f32(seed, chain_offset, iteration, dim, (unsigned long long)n, dz, du, nullptr, paired ? 1 : 0);
        else
            e = mm_launch_noise_generic_f64(seed, chain_offset, iteration, dim, (unsigned long long)n, dz, du, nullptr);
        if (e != hipSuccess)
            break;
        if ((e = hipDeviceSynchronize()) != hipSuccess)
            break;
        if ((e = hipMemcpy(z, dz, n * (size_t)dim * sizeof(T), hipMemcpyDeviceToHost)) != hipSuccess)
            break;
        if ((e = hipMemcpy(u, du, n * sizeof(T), hipMemcpyDeviceToHost)) != hipSuccess)
            break;
    } while (0);
    (void)hipFree(dz);
    (void)hipFree(du);
    return e == hipSuccess ? MMCMC_OK : (int)e;
}

} // namespace

/* ------------------------------------------------------------------ C ABI */

struct mmcmc_mh {
    Sampler *s;
};
struct mmcmc_hmc {
    Sampler *s;
};

extern "C" {

int mmcmc_version(void) { return MMCMC_VERSION; }

const char *mmcmc_status_string(int status)
{
    switch (status) {
    case MMCMC_OK:
        return "ok";
    case MMCMC_ERR_INVALID_ARG:
        return "invalid argument";
    case MMCMC_ERR_UNSUPPORTED:
        return "unsupported target kind / dimension / element type";
    case MMCMC_ERR_SHAPE:
        return "shape error";
    case MMCMC_ERR_NO_DEVICE:
        return "no HIP device (the engine has no CPU fallback)";
    case MMCMC_ERR_STATE:
        return "handle in the wrong state";
    case MMCMC_ERR_GROUP_BROKEN:
        return "device group broken: a run failed after some shards had advanced";
    default:
        return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}

int mmcmc_device_pci_bus_id(int device, char *buf, size_t len)
{
    if (!buf || len < 16)
        return MMCMC_ERR_INVALID_ARG;
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= n)
        return MMCMC_ERR_INVALID_ARG;
    const hipError_t e = hipDeviceGetPCIBusId(buf, (int)len, device);
    return e == hipSuccess ? MMCMC_OK : (int)e;
}

int mmcmc_device_count(int *count)
{
    if (!count)
        return MMCMC_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return MMCMC_ERR_NO_DEVICE;
    }
    *count = n;
    return MMCMC_OK;
}

int mmcmc_init_with_seed(size_t n, size_t d, uint64_t seed, double *out)
{
    if (!out && n * d > 0)
        return MMCMC_ERR_INVALID_ARG;
    mm_host::SmallRng rng(seed);
    for (size_t i = 0; i < n * d; ++i)
        out[i] = rng.standard_normal();
    return MMCMC_OK;
}

/* ---- MH ---- */
int mmcmc_mh_create(mmcmc_mh **out, const mmcmc_target_desc *target, const mmcmc_proposal_desc *proposal,
                    const void *init, size_t n_chains, int dtype, int device)
{
    if (!out || !proposal)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    Sampler *s = nullptr;
    int st;
    if (proposal->kind >= MMCMC_USER_PROPOSAL_BASE) {
        /* a proposal compiled from source (mmcmc_proposal_register_source): the target + proposal model stands in for the
         * target; it was compiled for exactly one target kind and dimension */
        const mm_user_target *model = mm_rtc_find(proposal->kind);
        if (!model || !mm_rtc_is_model(model) || !target)
            return MMCMC_ERR_UNSUPPORTED;
        if (mm_rtc_base_kind(model) != target->kind)
            return MMCMC_ERR_UNSUPPORTED;
        if (mm_rtc_dim(model) != target->dim)
            return MMCMC_ERR_SHAPE;
        if (target->kind < MM_USER_KIND_BASE) { /* the built-in target's own description must be valid */
            st = validate_target(target);
            if (st != MMCMC_OK)
                return st;
        }
        mmcmc_target_desc as_model = *target;
        as_model.kind = proposal->kind;
        g_allow_model = true;
        st = sampler_create(&s, MM_SAMPLER_MH, &as_model, proposal->std, 0, init, n_chains, dtype, device);
        g_allow_model = false;
    } else if (proposal->kind != MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN) {
        return MMCMC_ERR_UNSUPPORTED;
    } else {
        st = sampler_create(&s, MM_SAMPLER_MH, target, proposal->std, 0, init, n_chains, dtype, device);
    }
    if (st != MMCMC_OK)
        return st;
    mmcmc_mh *h = new (std::nothrow) mmcmc_mh{s};
    if (!h) {
        sampler_destroy(s);
        return (int)hipErrorOutOfMemory;
    }
    *out = h;
    return MMCMC_OK;
}
int mmcmc_mh_seed(mmcmc_mh *h, uint64_t seed)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->seed = seed;
    return MMCMC_OK;
}
int mmcmc_mh_set_chain_offset(mmcmc_mh *h, uint64_t off)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->chain_offset = off;
    return MMCMC_OK;
}
/* variant 5 (noise waves + transition waves, two to four waves per SIMD) exists up to dim 8 */
static bool split_ok(const Sampler *s)
{
    return s->dtype == MMCMC_F32 ? (s->kf && s->kf->run_mh_split) : (s->kd && s->kd->run_mh_split);
}
/* variant 6 = the run-time-dimension path: always for dimensions without a fixed kernel (then the only variant); on
 * request wherever the target kind has one and the chain vectors fit LDS (what the bit-identity tests use) */
static int set_variant_common(Sampler *s, int variant)
{
    if (s->user && s->generic) { /* a built-in target on its run-time compiled register kernels (7) or the run-time-D kernel (6) */
        if (variant != 6 && variant != 7)
            return MMCMC_ERR_UNSUPPORTED;
        s->variant = variant;
        return MMCMC_OK;
    }
    if (s->user)
        return variant == 7 ? MMCMC_OK : MMCMC_ERR_UNSUPPORTED;
    if (variant == 8) {
        if (!s->wide_ok)
            return MMCMC_ERR_UNSUPPORTED;
        s->variant = 8;
        return MMCMC_OK;
    }
    if (s->generic) {
        if (variant != 6)
            return MMCMC_ERR_UNSUPPORTED;
        s->variant = 6;
        return MMCMC_OK;
    }
    if (variant == 6) {
        const size_t lds = s->dtype == MMCMC_F32 ? mm_generic_store_bytes<float>(s->sampler, s->dim)
                                                 : mm_generic_store_bytes<double>(s->sampler, s->dim);
        if (!s->generic_ok || lds > MM_GENERIC_LDS_MAX)
            return MMCMC_ERR_UNSUPPORTED;
        s->variant = 6;
        return MMCMC_OK;
    }
    return 1; /* not handled here */
}

int mmcmc_mh_set_kernel_variant(mmcmc_mh *h, int variant)
{
    if (!h || variant < 0 || (variant > 2 && variant != 5 && variant != 6 && variant != 7))
        return MMCMC_ERR_INVALID_ARG;
    if (const int st = set_variant_common(h->s, variant); st <= 0)
        return st;
    if (variant == 5 && !split_ok(h->s))
        return MMCMC_ERR_UNSUPPORTED;
    h->s->variant = variant;
    return MMCMC_OK;
}
int mmcmc_hmc_set_kernel_variant(mmcmc_hmc *h, int variant)
{
    if (!h || variant < 0 || (variant > 3 && variant != 5 && variant != 6 && variant != 7 && variant != 8))
        return MMCMC_ERR_INVALID_ARG;
    if (const int st = set_variant_common(h->s, variant); st <= 0)
        return st;
    if (variant == 3 && !h->s->lg_ok)
        return MMCMC_ERR_UNSUPPORTED;
    if (variant == 5 && !split_ok(h->s))
        return MMCMC_ERR_UNSUPPORTED;
    h->s->variant = variant;
    return MMCMC_OK;
}
int mmcmc_hmc_kernel_variant(mmcmc_hmc *h) { return h ? h->s->variant : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_enable_timing(mmcmc_mh *h, int on)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->timing_enabled = on != 0;
    return MMCMC_OK;
}
int mmcmc_hmc_enable_timing(mmcmc_hmc *h, int on)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->timing_enabled = on != 0;
    return MMCMC_OK;
}
int mmcmc_mh_set_iters_per_launch(mmcmc_mh *h, uint32_t iters)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->iters_per_launch = iters;
    return MMCMC_OK;
}
int mmcmc_mh_run(mmcmc_mh *h, size_t n_collect, size_t n_discard, void *out, int out_is_device,
                 uint64_t *accept_counts, void *stream)
{
    return h ? sampler_run(h->s, n_collect, n_discard, out, out_is_device, accept_counts, stream)
             : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_run_rows(mmcmc_mh *h, size_t n_rows, size_t n_discard, void *out_device, size_t n_total_rows, size_t row0,
                      void *stream)
{
    return h ? sampler_run(h->s, n_rows, n_discard, out_device, 1, nullptr, stream, n_total_rows, row0) : MMCMC_ERR_INVALID_ARG;
}
static int sampler_shape(const Sampler *s, size_t *n_chains, int *dim, int *dtype, int *device)
{
    if (n_chains)
        *n_chains = s->n_chains;
    if (dim)
        *dim = s->dim;
    if (dtype)
        *dtype = s->dtype;
    if (device)
        *device = s->device;
    return MMCMC_OK;
}
int mmcmc_mh_shape(mmcmc_mh *h, size_t *n_chains, int *dim, int *dtype, int *device)
{
    return h ? sampler_shape(h->s, n_chains, dim, dtype, device) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_mh_state(mmcmc_mh *h, void *out) { return h ? sampler_state(h->s, out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_sync(mmcmc_mh *h) { return h ? sampler_sync(h->s) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_timing(mmcmc_mh *h, mmcmc_timing *t) { return h ? sampler_timing(h->s, t) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_mh_destroy(mmcmc_mh *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    int st = sampler_destroy(h->s);
    delete h;
    return st;
}

/* ---- HMC ---- */
int mmcmc_hmc_create(mmcmc_hmc **out, const mmcmc_target_desc *target, const void *init, size_t n_chains,
                     double step_size, int n_leapfrog, int dtype, int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    Sampler *s = nullptr;
    int st = sampler_create(&s, MM_SAMPLER_HMC, target, step_size, n_leapfrog, init, n_chains, dtype, device);
    if (st != MMCMC_OK)
        return st;
    mmcmc_hmc *h = new (std::nothrow) mmcmc_hmc{s};
    if (!h) {
        sampler_destroy(s);
        return (int)hipErrorOutOfMemory;
    }
    *out = h;
    return MMCMC_OK;
}
int mmcmc_hmc_seed(mmcmc_hmc *h, uint64_t seed)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->seed = seed;
    return MMCMC_OK;
}
int mmcmc_hmc_set_chain_offset(mmcmc_hmc *h, uint64_t off)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->chain_offset = off;
    return MMCMC_OK;
}
int mmcmc_hmc_set_iters_per_launch(mmcmc_hmc *h, uint32_t iters)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->s->iters_per_launch = iters;
    return MMCMC_OK;
}
int mmcmc_hmc_run(mmcmc_hmc *h, size_t n_collect, size_t n_discard, void *out, int out_is_device,
                  uint64_t *accept_counts, void *stream)
{
    return h ? sampler_run(h->s, n_collect, n_discard, out, out_is_device, accept_counts, stream)
             : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_run_rows(mmcmc_hmc *h, size_t n_rows, size_t n_discard, void *out_device, size_t n_total_rows, size_t row0,
                       void *stream)
{
    return h ? sampler_run(h->s, n_rows, n_discard, out_device, 1, nullptr, stream, n_total_rows, row0) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_shape(mmcmc_hmc *h, size_t *n_chains, int *dim, int *dtype, int *device)
{
    return h ? sampler_shape(h->s, n_chains, dim, dtype, device) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_step(mmcmc_hmc *h, void *stream)
{
    return h ? sampler_run(h->s, 0, 1, nullptr, 1, nullptr, stream) : MMCMC_ERR_INVALID_ARG;
}
int mmcmc_hmc_state(mmcmc_hmc *h, void *out) { return h ? sampler_state(h->s, out) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_sync(mmcmc_hmc *h) { return h ? sampler_sync(h->s) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_timing(mmcmc_hmc *h, mmcmc_timing *t) { return h ? sampler_timing(h->s, t) : MMCMC_ERR_INVALID_ARG; }
int mmcmc_hmc_destroy(mmcmc_hmc *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    int st = sampler_destroy(h->s);
    delete h;
    return st;
}

/* ---- densities / noise (parity tests) ---- */
int mmcmc_logp_grad_batch(const mmcmc_target_desc *target, int dtype, const void *x, size_t n, void *logp,
                          void *grad, int device)
{
    if (!x || !logp || n == 0 || (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    int st = validate_target(target);
    if (st != MMCMC_OK)
        return st;
    st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DeviceGuard g(device);
    return dtype == MMCMC_F32 ? logp_grad_batch_t<float>(target, x, n, logp, grad)
                              : logp_grad_batch_t<double>(target, x, n, logp, grad);
}

int mmcmc_draw_noise(uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n_chains, int dim, int dtype,
                     void *z, void *u, int device)
{
    if (!z || !u || n_chains == 0 || dim <= 0 || (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    int st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DeviceGuard g(device);
    return dtype == MMCMC_F32 ? draw_noise_t<float>(seed, chain_offset, iteration, n_chains, dim, z, u)
                              : draw_noise_t<double>(seed, chain_offset, iteration, n_chains, dim, z, u);
}

int mmcmc_draw_noise_mh(uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n_chains, int dim, int dtype,
                        void *z, void *u, int device)
{
    if (!z || !u || n_chains == 0 || dim <= 0 || (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    int st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DeviceGuard g(device);
    return dtype == MMCMC_F32 ? draw_noise_t<float>(seed, chain_offset, iteration, n_chains, dim, z, u, 1)
                              : draw_noise_t<double>(seed, chain_offset, iteration, n_chains, dim, z, u, 1);
}

} /* extern "C" */
