/*
 * mm_discrete.hip -- integer-state Metropolis-Hastings on the GPU and its C ABI (include/mmcmc.h: mmcmc_mh_discrete_*).
 * One chain per lane, the whole run in one launch (ChainRunner::run over MHMarkovChain<i32 / usize>,
 * core.rs:55-73, 176-186); the transition is mm_discrete_step (mm_discrete.h).  Samples [C, n_collect] i32 are staged
 * per wave in LDS (64 chains x 64 iterations) and written as 256-byte rows.
 */
#include "../../include/mmcmc.h"
#include "mm_hostcopy.h"

#include <hip/hip_runtime.h>

#include <atomic>

#include <new>
#include <vector>

#include "mm_discrete.h"
#include "mm_discrete_kernels.h"
#include "mm_kernels.h"
#include "mm_rtc.h"

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

struct run_args {
    mm_discrete_params P;
    int32_t *state;                 /* [C] */
    int32_t *out;                   /* [C, n_collect] or NULL */
    unsigned long long *accept;     /* [C] running totals */
    unsigned long long n_chains, seed, chain_offset;
    unsigned int iter0, n_discard, n_collect;
};


__global__ __launch_bounds__(64) void mm_discrete_kernel(const run_args a)
{
    /* samples are staged and flushed like the continuous samplers' (mm_kernels.h): 96 transitions per tile, 16-byte stores */
    using Tile = mm_tile<int32_t, 1>;
    constexpr unsigned int TILE = Tile::tile_t;
    constexpr int STRIDE = Tile::stride;
    __shared__ __attribute__((aligned(16))) int32_t tile[64 * STRIDE];
    /* the log-density table goes to LDS: two dependent per-lane look-ups per transition at HBM / L2 latency, with one
     * wave per SIMD to hide them, were most of this kernel's time */
    __shared__ double tab_lds[MM_DISCRETE_POISSON_TABLE];
    const int lane = threadIdx.x;
    mm_discrete_params P = a.P;
    /* table_len <= MM_DISCRETE_POISSON_TABLE by construction (mm_discrete_fill_table's cap); unconditional, so that the
     * pointer is an LDS pointer to the compiler and the look-ups ds_read, not flat_load */
    for (int i = lane; i < P.table_len && i < MM_DISCRETE_POISSON_TABLE; i += 64)
        tab_lds[i] = a.P.logp[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    P.logp = tab_lds;
    P.table_len = P.table_len < MM_DISCRETE_POISSON_TABLE ? P.table_len : MM_DISCRETE_POISSON_TABLE;
    const unsigned long long c0 = (unsigned long long)blockIdx.x * 64, c = c0 + lane;
    const bool active = c < a.n_chains;
    int32_t x = active ? a.state[c] : 0;
    unsigned long long n_acc = 0;
    const unsigned int total = a.n_discard + a.n_collect;
    unsigned int col = 0, t0 = 0;
    for (unsigned int t = 0; t < total; ++t) {
        n_acc += (unsigned long long)mm_discrete_step(P, &x, a.seed, a.chain_offset + c, a.iter0 + t);
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

/* The same run with the work of a chain spread over four waves of one SIMD, like the continuous samplers'
 * (mm_split_kernels.h): workgroup = 1024 threads = 4 transition waves (64 chains each) + 3 noise waves per transition
 * wave.  The noise block is a pure function of (seed, chain, iteration): noise wave r draws the blocks of iterations
 * r, r + 3, ... of a batch of RB into an LDS ring, one barrier per batch hands a filled half over.  Bit-identical to
 * mm_discrete_kernel (same blocks, same accept decisions).
 *
 * Round 3 (tools/experiments/ds_probe.py, cycle stamps in the transition wave): the transition wave never waited at the
 * barrier -- it spent 627 cycles per transition on ~30 instructions: an LDS round trip for the draw, two dependent
 * look-ups of the log-density with their bounds checks as exec-mask branches, six dependent f64 operations; and the
 * bare skeleton (no Philox, no transition, no output) still took 240 cycles per transition, the LDS pipe of the CU
 * moving 16 bytes per chain and transition through the ring twice.  Now:
 *   - the ring carries 4 bytes per draw: the f32 table logarithm of the accept uniform with the direction in its sign
 *     bit (ln u <= 0).  The uniform itself is needed only inside the filter's band (one transition in ~10^5) and is
 *     drawn again there;
 *   - a transition reads no log-density: the workgroup tabulates, per state, the log accept ratio of the move down
 *     and of the move up (the expression of mm_discrete_step_draws, evaluated once), the chain carries the pair of its
 *     state, and the pair of the PROPOSED state is fetched while the accept test runs (its address depends on the state
 *     and the direction, not on the decision);
 *   - a batch's draws are read from the ring in one go and its transitions are unrolled.
 * A wave with a chain outside the table (only possible with such an initial state) keeps the plain transition. */
#ifndef MM_DS_NN /* measurement knobs (tools/experiments/ds_knobs.sh) */
#define MM_DS_NN 3
#endif
#ifndef MM_DS_RB
#define MM_DS_RB 12
#endif
constexpr int DS_NN = MM_DS_NN, DS_RB = MM_DS_RB, DS_TILE_T = 48;
using DsTile = mm_tile_t<int32_t, 1, DS_TILE_T>;
constexpr size_t DS_TILE_BYTES = (DsTile::lds_bytes_per_wave + 15) / 16 * 16;
constexpr size_t DS_RING_BYTES = (size_t)2 * DS_RB * 64 * 4;
constexpr size_t DS_TABLE_BYTES = (size_t)MM_DISCRETE_POISSON_TABLE * sizeof(double);
constexpr size_t DS_RATIO_BYTES = (size_t)(MM_DISCRETE_POISSON_TABLE + 1) * 16; /* {down, up} per state + one row past the end */
constexpr size_t DS_LNM_BYTES = (size_t)MM_LNM_ROWS * 16; /* the f32 logarithm's table (mm_lnu_f32), read by the noise waves */
constexpr size_t DS_LDS_BYTES = DS_RATIO_BYTES + DS_TABLE_BYTES + DS_LNM_BYTES + 4 * DS_TILE_BYTES + 4 * DS_RING_BYTES;
static_assert(DS_LDS_BYTES <= 160 * 1024 && DS_RB % DS_NN == 0 && DS_TILE_T % DS_RB == 0, "LDS plan of the split discrete kernel");

struct ds_lnm_lds { /* Tab of mm_lnu_f32 with the logarithm's rows in LDS */
    mm_lds_f4_ptr tab;
    __device__ __forceinline__ void row_ln(uint32_t r, float *c) const
    {
        const mm_v4f v = tab[r];
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
};
typedef double ds_ratio_pair __attribute__((ext_vector_type(2))); /* log accept ratio of the move down, of the move up */

/* what a noise wave hands over per transition: -|table logarithm of (float)u| with the sign bit replaced by the
 * direction (set = up).  Any value within the filter's band of ln u yields the exact decision, ln u <= 0, so dropping
 * a positive sign the table's rounding may produce next to u = 1 moves the value towards ln u. */
__device__ __forceinline__ uint32_t ds_draw_pack(const mm_u32x4 &b, const ds_lnm_lds &lt)
{
    const double u = mm_u53(b.w[2], b.w[3]);
    return (mm_f2u(mm_lnu_f32((float)u, lt)) & 0x7fffffffu) | (b.w[0] & 0x80000000u);
}

/* the batches of a run: DS_RB transitions, cut at n_discard -- so that the collected samples of a batch fill whole
 * columns k * DS_RB .. of the tile and the tile is full exactly at the end of a batch (DS_TILE_T % DS_RB == 0) -- and at
 * the end of the run.  Both roles walk the same sequence. */
__device__ __forceinline__ unsigned int ds_batch(unsigned int done, unsigned int n_discard, unsigned int total)
{
    const unsigned int left = (done < n_discard ? n_discard : total) - done;
    return left < (unsigned int)DS_RB ? left : (unsigned int)DS_RB;
}

__device__ __forceinline__ void ds_setprio(int p)
{
    switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

/* KIND: the model (compile time: the proposal rule is a switch inside the transition) */
template <int KIND>
__global__ __launch_bounds__(256 * (1 + DS_NN)) void mm_discrete_split_kernel(const run_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ds_lds[];
    ds_ratio_pair *ratio_lds = reinterpret_cast<ds_ratio_pair *>(ds_lds);
    double *tab_lds = reinterpret_cast<double *>(ds_lds + DS_RATIO_BYTES);
    float *lnm_lds = reinterpret_cast<float *>(ds_lds + DS_RATIO_BYTES + DS_TABLE_BYTES);
    unsigned char *const per_pair = ds_lds + DS_RATIO_BYTES + DS_TABLE_BYTES + DS_LNM_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave & 3, noise_rank = (wave >> 2) - 1;
    int32_t *tile = reinterpret_cast<int32_t *>(per_pair + (size_t)pair * DS_TILE_BYTES);
    uint32_t *ring = reinterpret_cast<uint32_t *>(per_pair + 4 * DS_TILE_BYTES + (size_t)pair * DS_RING_BYTES);
    mm_discrete_params P = a.P;
    P.kind = KIND;
    constexpr int NT = 256 * (1 + DS_NN);
    for (int i = threadIdx.x; i < P.table_len && i < MM_DISCRETE_POISSON_TABLE; i += NT)
        tab_lds[i] = a.P.logp[i];
    for (int i = threadIdx.x; i < MM_LNM_ROWS; i += NT)
        reinterpret_cast<mm_v4f *>(lnm_lds)[i] = *reinterpret_cast<const mm_v4f *>(mm_lnm_tab_d + 4 * i);
    P.logp = tab_lds;
    P.table_len = P.table_len < MM_DISCRETE_POISSON_TABLE ? P.table_len : MM_DISCRETE_POISSON_TABLE;
    __syncthreads(); /* the log-density table is complete */
    for (int i = threadIdx.x; i <= P.table_len; i += NT) {
        int prop;
        ds_ratio_pair r;
        r.x = mm_discrete_log_ratio(P, i, -1, &prop);
        r.y = mm_discrete_log_ratio(P, i, 1, &prop);
        ratio_lds[i] = r;
    }
    const unsigned long long c0 = (unsigned long long)blockIdx.x * 256ull + (unsigned long long)pair * 64ull, c = c0 + lane;
    const unsigned long long chain = a.chain_offset + c;
    const unsigned int total = a.n_discard + a.n_collect;
    __syncthreads(); /* ... and the table of the ratios */
    if (noise_rank >= 0) {
        ds_lnm_lds lt;
        lt.tab = (mm_lds_f4_ptr)reinterpret_cast<const mm_v4f *>(lnm_lds);
        int half = 0;
        for (unsigned int done = 0, nb; done < total; done += nb) {
            nb = ds_batch(done, a.n_discard, total);
            uint32_t *const batch = ring + (size_t)half * DS_RB * 64 + lane;
            if (nb == (unsigned int)DS_RB) {
                MM_UNROLL
                for (int k = 0; k < DS_RB / DS_NN; ++k) { /* independent blocks: their Philox rounds interleave */
                    /* the arbiter serves the highest priority first, then the oldest wave: a noise wave lowers its priority
                     * with every draw of the batch it has made, so that the three end their shares together instead of one
                     * after the other (the last one alone on the SIMD at a lone wave's issue rate): -3 % */
                    ds_setprio(3 - k > 0 ? 3 - k : 0);
                    const unsigned int j = (unsigned int)(noise_rank + k * DS_NN);
                    batch[(size_t)j * 64] = ds_draw_pack(mm_block(a.seed, chain, a.iter0 + done + j, 0u), lt);
                }
            } else {
                for (unsigned int j = (unsigned int)noise_rank; j < nb; j += DS_NN)
                    batch[(size_t)j * 64] = ds_draw_pack(mm_block(a.seed, chain, a.iter0 + done + j, 0u), lt);
            }
            half ^= 1;
            __syncthreads();
        }
        __syncthreads();
        return;
    }
    __builtin_amdgcn_s_setprio(3); /* the transition wave's dependent chain is the longer leg of a batch */
    const bool active = c < a.n_chains;
    int32_t x = active ? a.state[c] : 0;
    /* wave-uniform: every chain inside the table (and then for good: a move out of it has ratio -inf) */
    const bool tabulated = __all(x >= 0 && x < P.table_len) != 0;
    ds_ratio_pair rx = tabulated ? ratio_lds[x] : ds_ratio_pair{0.0, 0.0};
    unsigned int n_acc = 0, col = 0, t0 = 0;
    int half = 0;
    for (unsigned int done = 0, nb; done < total; done += nb) {
        nb = ds_batch(done, a.n_discard, total);
        __syncthreads(); /* ring half `half` holds the draws of this batch */
        const uint32_t *const batch = ring + (size_t)half * DS_RB * 64 + lane;
        const bool collecting = done >= a.n_discard && a.out; /* a batch lies on one side of n_discard (ds_batch) */
        int32_t *const row = tile + lane * DsTile::stride + col;
        unsigned int j = 0;
        if (tabulated && nb == (unsigned int)DS_RB) {
            /* the common case, unrolled and free of branches: the batch's draws come out of the ring in one go and the
             * filter's bounds are formed ahead of the chain of transitions.  A transition that finds its ratio inside
             * the band (about 4e-6 of them) is decided provisionally; the batch is then run again, from the state it
             * started in, by the loop below. */
            double hi[DS_RB], lo[DS_RB];
            bool up[DS_RB];
            MM_UNROLL
            for (int k = 0; k < DS_RB; ++k) {
                const uint32_t d = batch[(size_t)k * 64];
                const double mid = (double)mm_u2f(d | 0x80000000u), band = 1e-6 + 1e-6 * -mid; /* mm_ratio_exceeds_ln_u_given */
                up[k] = (d >> 31) != 0u;
                hi[k] = mid + band;
                lo[k] = mid - band;
            }
            const int32_t x0 = x;
            const ds_ratio_pair rx0 = rx;
            const unsigned int n_acc0 = n_acc;
            bool in_band = false, accept = false;
            ds_ratio_pair rp = rx;
            MM_UNROLL
            for (int k = 0; k < DS_RB; ++k) {
                /* the pair of the proposed state is asked for first and used by the NEXT transition: a look-up has a
                 * whole transition to come back (two transitions share one LDS latency) */
                const int prop = mm_discrete_propose(P, x, up[k] ? 1 : -1);
                const ds_ratio_pair rp_next = ratio_lds[prop];
                __builtin_amdgcn_sched_barrier(0);
                rx = accept ? rp : rx; /* the previous transition's move, now that its look-up is back */
                rp = rp_next;
                const double ratio = up[k] ? rx.y : rx.x;
                accept = ratio > hi[k];
                in_band |= !accept && !(ratio <= lo[k]);
                x = accept ? prop : x;
                n_acc += accept ? 1u : 0u;
                if (collecting)
                    row[k] = x;
            }
            rx = accept ? rp : rx;
            if (__any(in_band)) {
                x = x0;
                rx = rx0;
                n_acc = n_acc0;
            } else {
                j = nb;
            }
        }
        for (; j < nb; ++j) {
            const unsigned int t = done + j;
            if (tabulated) {
                const uint32_t draw = batch[(size_t)j * 64];
                const bool up = (draw >> 31) != 0u;
                const int prop = mm_discrete_propose(P, x, up ? 1 : -1);
                const double ratio = up ? rx.y : rx.x;
                const double mid = (double)mm_u2f(draw | 0x80000000u), band = 1e-6 + 1e-6 * -mid; /* mm_ratio_exceeds_ln_u_given */
                bool accept = ratio > mid + band;
                if (!accept && !(ratio <= mid - band)) { /* inside the band: the uniform again, and its f64 logarithm */
                    const mm_u32x4 b = mm_block(a.seed, chain, a.iter0 + t, 0u);
                    accept = ratio > mm_log(mm_u53(b.w[2], b.w[3]));
                }
                if (accept) {
                    x = prop;
                    rx = ratio_lds[prop];
                    ++n_acc;
                }
            } else {
                n_acc += (unsigned int)mm_discrete_step(P, &x, a.seed, chain, a.iter0 + t);
            }
            if (collecting)
                row[j] = x;
        }
        if (collecting) {
            col += nb;
            if (col == (unsigned int)DS_TILE_T || done + nb == total) {
                mm_flush_tile_raw<int32_t, 1, DsTile>(a.out, a.n_collect, a.n_chains, tile, lane, c0, t0, col);
                t0 += col;
                col = 0;
            }
        }
        half ^= 1;
    }
    __syncthreads();
    if (active) {
        a.state[c] = x;
        a.accept[c] += n_acc;
    }
}

} // namespace

struct mmcmc_mh_discrete {
    int device = 0, kind = 0;
    size_t n_chains = 0;
    uint64_t seed = 0, chain_offset = 0;
    uint32_t iter = 0;
    int variant = 1; /* 1: noise waves + transition waves (from 4096 chains on); 0: one wave per SIMD */
    mm_discrete_params P{};
    const mm_user_target *user = nullptr; /* a model compiled from source (mmcmc_discrete_register_source) */
    double user_params[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double *d_table = nullptr;
    int32_t *d_state = nullptr;
    unsigned long long *d_accept = nullptr;
    hipStream_t stream = nullptr;
};

extern "C" {

int mmcmc_mh_discrete_create(mmcmc_mh_discrete **out, int kind, const double *params, const int32_t *init,
                             size_t n_chains, int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!params || !init || n_chains == 0)
        return MMCMC_ERR_INVALID_ARG;
    const mm_user_target *user = kind >= MM_USER_KIND_BASE ? mm_rtc_find(kind) : nullptr;
    if (user && !mm_rtc_is_discrete(user))
        return MMCMC_ERR_UNSUPPORTED;
    if (!user && kind != MMCMC_POISSON_REFLECT && kind != MMCMC_BINOMIAL_CLAMP && kind != MMCMC_POISSON_NONNEG)
        return MMCMC_ERR_UNSUPPORTED;
    if (user) {
        /* params: eight doubles, handed to the functor as they are */
    } else if (kind == MMCMC_BINOMIAL_CLAMP) {
        if (!(params[0] >= 1.0 && params[0] < 1e6) || !(params[1] > 0.0 && params[1] < 1.0))
            return MMCMC_ERR_INVALID_ARG;
    } else if (!(params[0] > 0.0 && params[0] <= 256.0)) {
        return MMCMC_ERR_INVALID_ARG; /* the Poisson table ends at 1023 */
    }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= nd)
        return MMCMC_ERR_INVALID_ARG;
    mmcmc_mh_discrete *h = new (std::nothrow) mmcmc_mh_discrete();
    if (!h)
        return (int)hipErrorOutOfMemory;
    h->device = device;
    h->kind = kind;
    h->n_chains = n_chains;
    DevGuard g(device);
    h->user = user;
    if (user)
        for (int i = 0; i < 8; ++i)
            h->user_params[i] = params[i];
    /* the kernel keeps the log-density table in LDS: at most MM_DISCRETE_POISSON_TABLE states */
    if (kind == MMCMC_BINOMIAL_CLAMP && !(params[0] >= 0 && params[0] < (double)MM_DISCRETE_POISSON_TABLE)) {
        delete h;
        return MMCMC_ERR_UNSUPPORTED;
    }
    const int cap = user ? 1 : kind == MMCMC_BINOMIAL_CLAMP ? (int)params[0] + 1 : MM_DISCRETE_POISSON_TABLE;
    std::vector<double> tab((size_t)cap, 0.0);
    const int len = user ? 1 : mm_discrete_fill_table(kind, params, tab.data(), cap);
    h->P.kind = kind;
    h->P.n = kind == MMCMC_BINOMIAL_CLAMP ? (int)params[0] : 0;
    h->P.table_len = len;
    h->P.ln_half = mm_log(0.5);
    auto fail = [&](int st) {
        mmcmc_mh_discrete_destroy(h);
        return st;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&h->d_table, (size_t)len * sizeof(double))) != hipSuccess)
        return fail((int)e);
    if ((e = hipMemcpy(h->d_table, tab.data(), (size_t)len * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
        return fail((int)e);
    h->P.logp = h->d_table;
    if ((e = hipMalloc((void **)&h->d_state, n_chains * sizeof(int32_t))) != hipSuccess)
        return fail((int)e);
    if ((e = hipMemcpy(h->d_state, init, n_chains * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return fail((int)e);
    if ((e = hipMalloc((void **)&h->d_accept, n_chains * sizeof(unsigned long long))) != hipSuccess)
        return fail((int)e);
    if ((e = hipMemset(h->d_accept, 0, n_chains * sizeof(unsigned long long))) != hipSuccess)
        return fail((int)e);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamDefault)) != hipSuccess)
        return fail((int)e);
    *out = h;
    return MMCMC_OK;
}

int mmcmc_mh_discrete_set_kernel_variant(mmcmc_mh_discrete *h, int variant)
{
    if (!h || (variant != 0 && variant != 1))
        return MMCMC_ERR_INVALID_ARG;
    h->variant = variant;
    return MMCMC_OK;
}

int mmcmc_mh_discrete_seed(mmcmc_mh_discrete *h, uint64_t seed)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->seed = seed;
    return MMCMC_OK;
}

int mmcmc_mh_discrete_set_chain_offset(mmcmc_mh_discrete *h, uint64_t off)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->chain_offset = off;
    return MMCMC_OK;
}

int mmcmc_mh_discrete_run(mmcmc_mh_discrete *h, size_t n_collect, size_t n_discard, int32_t *out, int out_is_device,
                          void *stream)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if ((uint64_t)h->iter + n_collect + n_discard >= (1ull << 32) || n_collect >= (1ull << 31))
        return MMCMC_ERR_SHAPE;
    DevGuard g(h->device);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t out_bytes = h->n_chains * n_collect * sizeof(int32_t);
    int32_t *d_out = nullptr;
    bool staged = false;
    if (out && n_collect > 0) {
        if (out_is_device) {
            d_out = out;
        } else {
            MM_HIP(hipMalloc((void **)&d_out, out_bytes));
            staged = true;
        }
    }
    run_args a;
    a.P = h->P;
    a.state = h->d_state;
    a.out = d_out;
    a.accept = h->d_accept;
    a.n_chains = h->n_chains;
    a.seed = h->seed;
    a.chain_offset = h->chain_offset;
    a.iter0 = h->iter;
    a.n_discard = (unsigned int)n_discard;
    a.n_collect = (unsigned int)n_collect;
    /* four waves per SIMD from 4096 chains on (below that the plain kernel's waves are spread thinner than one per SIMD
     * anyway); variant 0 keeps the one-wave kernel (A/B and tests: the two are bit-identical) */
    hipError_t e = hipSuccess;
    if (h->user) {
        mm_discrete_user_args ua;
        for (int i = 0; i < 8; ++i)
            ua.params[i] = h->user_params[i];
        ua.state = a.state;
        ua.out = a.out;
        ua.accept = a.accept;
        ua.n_chains = a.n_chains;
        ua.seed = a.seed;
        ua.chain_offset = a.chain_offset;
        ua.iter0 = a.iter0;
        ua.n_discard = a.n_discard;
        ua.n_collect = a.n_collect;
        e = mm_rtc_launch_discrete(h->user, &ua, sizeof(ua), (unsigned int)((h->n_chains + 63) / 64), st);
    } else if (h->n_chains >= 4096 && h->variant == 1) {
        const void *fn = h->kind == MMCMC_POISSON_REFLECT  ? reinterpret_cast<const void *>(mm_discrete_split_kernel<MM_POISSON_REFLECT>)
                         : h->kind == MMCMC_BINOMIAL_CLAMP ? reinterpret_cast<const void *>(mm_discrete_split_kernel<MM_BINOMIAL_CLAMP>)
                                                           : reinterpret_cast<const void *>(mm_discrete_split_kernel<MM_POISSON_NONNEG>);
        const int slot = h->kind == MMCMC_POISSON_REFLECT ? 0 : h->kind == MMCMC_BINOMIAL_CLAMP ? 1 : 2;
        static std::atomic<unsigned long long> attr_set[3];
        if (h->device >= 64 || !((attr_set[slot].load(std::memory_order_relaxed) >> h->device) & 1ull)) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DS_LDS_BYTES);
            if (e == hipSuccess && h->device < 64)
                attr_set[slot].fetch_or(1ull << h->device, std::memory_order_relaxed);
        }
        if (e == hipSuccess) {
            const dim3 grid((unsigned int)((h->n_chains + 255) / 256)), block(256 * (1 + DS_NN));
            if (slot == 0)
                hipLaunchKernelGGL(mm_discrete_split_kernel<MM_POISSON_REFLECT>, grid, block, DS_LDS_BYTES, st, a);
            else if (slot == 1)
                hipLaunchKernelGGL(mm_discrete_split_kernel<MM_BINOMIAL_CLAMP>, grid, block, DS_LDS_BYTES, st, a);
            else
                hipLaunchKernelGGL(mm_discrete_split_kernel<MM_POISSON_NONNEG>, grid, block, DS_LDS_BYTES, st, a);
            e = hipGetLastError();
        }
    } else {
        hipLaunchKernelGGL(mm_discrete_kernel, dim3((unsigned int)((h->n_chains + 63) / 64)), dim3(64), 0, st, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        if (staged)
            (void)hipFree(d_out);
        return (int)e;
    }
    h->iter += (uint32_t)(n_collect + n_discard);
    if (staged) {
        MM_HIP(mm_copy_to_host(out, d_out, out_bytes, h->device, st));
        MM_HIP(hipFree(d_out));
    }
    return MMCMC_OK;
}

int mmcmc_mh_discrete_state(mmcmc_mh_discrete *h, int32_t *out)
{
    if (!h || !out)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    MM_HIP(hipDeviceSynchronize());
    MM_HIP(hipMemcpy(out, h->d_state, h->n_chains * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MMCMC_OK;
}

int mmcmc_mh_discrete_accept_counts(mmcmc_mh_discrete *h, uint64_t *out)
{
    if (!h || !out)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    MM_HIP(hipDeviceSynchronize());
    MM_HIP(hipMemcpy(out, h->d_accept, h->n_chains * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MMCMC_OK;
}

int mmcmc_mh_discrete_sync(mmcmc_mh_discrete *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    MM_HIP(hipStreamSynchronize(h->stream));
    return MMCMC_OK;
}

int mmcmc_mh_discrete_destroy(mmcmc_mh_discrete *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_table);
    (void)hipFree(h->d_state);
    (void)hipFree(h->d_accept);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
    return MMCMC_OK;
}

} /* extern "C" */
