/*
 * mm_gibbs.hip -- the Gibbs sampler with the built-in mixture conditional on the GPU and its C ABI
 * (include/mmcmc.h: mmcmc_gibbs_mixture_*).  One chain per lane, the whole run in one launch (GibbsSampler through
 * ChainRunner::run, gibbs.rs:139-203, core.rs:176-186); the sweep is mm_gibbs_mixture_step (mm_gibbs.h).  Samples
 * [C, n_collect, 2] f64 are staged per wave in LDS (64 chains x 16 sweeps) and written as 256-byte rows.
 */
#include "../../include/mmcmc.h"
#include "mm_hostcopy.h"

#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdlib>
#include <cstring>
#include <new>

#include "mm_gibbs.h"
#include "mm_kernels.h"

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
    mm_mixture_params P;
    double *state;  /* [C, 2] */
    double *out;    /* [C, n_collect, 2] or NULL */
    unsigned long long n_chains, seed, chain_offset;
    unsigned int iter0, n_discard, n_collect;
};


__global__ __launch_bounds__(64) void mm_gibbs_mixture_kernel(const run_args a)
{
    /* samples are staged and flushed like the continuous samplers' (mm_kernels.h): 24 sweeps per tile, 16-byte stores */
    using Tile = mm_tile<double, 2>;
    constexpr unsigned int TILE = Tile::tile_t;
    constexpr int STRIDE = Tile::stride;
    __shared__ __attribute__((aligned(16))) double tile[64 * STRIDE];
    const int lane = threadIdx.x;
    const unsigned long long c0 = (unsigned long long)blockIdx.x * 64, c = c0 + lane;
    const bool active = c < a.n_chains;
    double s[2] = {active ? a.state[2 * c] : 0.0, active ? a.state[2 * c + 1] : 0.0};
    const unsigned int total = a.n_discard + a.n_collect;
    unsigned int col = 0, t0 = 0;
    for (unsigned int t = 0; t < total; ++t) {
        mm_gibbs_mixture_step(a.P, s, a.seed, a.chain_offset + c, a.iter0 + t);
        if (t >= a.n_discard && a.out) {
            tile[lane * STRIDE + 2 * col] = s[0];
            tile[lane * STRIDE + 2 * col + 1] = s[1];
            ++col;
            if (col == TILE || t + 1 == total) {
                mm_flush_tile_raw<double, 2>(a.out, a.n_collect, a.n_chains, tile, lane, c0, t0, col);
                t0 += col;
                col = 0;
            }
        }
    }
    if (active) {
        a.state[2 * c] = s[0];
        a.state[2 * c + 1] = s[1];
    }
}

/* The same run with a chain's work spread over four waves of one SIMD, like the continuous samplers' (mm_split_kernels.h)
 * and the integer-state MH (mm_discrete.hip): workgroup = 1024 threads = 4 transition waves (64 chains each) + 3 noise
 * waves per transition wave.  A sweep is two Philox blocks + Box-Muller (log, sqrt, cos) on the noise side and two normal
 * pdfs (exp, divisions) + the conditional draws on the transition side, about as many instructions each; one wave per
 * SIMD (all that 65 536 chains give mm_gibbs_mixture_kernel) issues one every 4.4+ cycles.  (z0, u) of a sweep is a pure
 * function of (seed, chain, iteration): noise wave r fills the sweeps r, r + 3, ... of a batch of GS_RB into an LDS ring
 * (16 bytes per chain and sweep), one barrier per batch hands a filled half over.  Bit-identical to the plain kernel
 * (same mm_gibbs_mixture_noise, same mm_gibbs_mixture_step_noise). */
constexpr int GS_NN = 3, GS_RB = 6, GS_TILE_T = 24;
using GsTile = mm_tile_t<double, 2, GS_TILE_T>;
constexpr size_t GS_TILE_BYTES = (GsTile::lds_bytes_per_wave + 15) / 16 * 16;
constexpr size_t GS_RING_BYTES = (size_t)2 * GS_RB * 64 * 16;
constexpr size_t GS_LDS_BYTES = 4 * GS_TILE_BYTES + 4 * GS_RING_BYTES;
static_assert(GS_LDS_BYTES <= 160 * 1024 && GS_RB % GS_NN == 0 && GS_TILE_T % GS_RB == 0, "LDS plan of the split Gibbs kernel");

struct __attribute__((aligned(16))) gs_draw {
    double z0, u;
};

/* the batches of a run: GS_RB sweeps, cut at n_discard (the collected sweeps of a batch fill whole columns of the tile and
 * the tile is full exactly at the end of a batch: GS_TILE_T % GS_RB == 0) and at the end of the run; both roles walk the
 * same sequence (as in mm_discrete.hip) */
__device__ __forceinline__ unsigned int gs_batch(unsigned int done, unsigned int n_discard, unsigned int total)
{
    const unsigned int left = (done < n_discard ? n_discard : total) - done;
    return left < (unsigned int)GS_RB ? left : (unsigned int)GS_RB;
}

__global__ __launch_bounds__(256 * (1 + GS_NN)) void mm_gibbs_mixture_split_kernel(const run_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave & 3, noise_rank = (wave >> 2) - 1;
    double *tile = reinterpret_cast<double *>(gs_lds + (size_t)pair * GS_TILE_BYTES);
    gs_draw *ring = reinterpret_cast<gs_draw *>(gs_lds + 4 * GS_TILE_BYTES + (size_t)pair * GS_RING_BYTES);
    const unsigned long long c0 = (unsigned long long)blockIdx.x * 256ull + (unsigned long long)pair * 64ull, c = c0 + lane;
    const unsigned long long chain = a.chain_offset + c;
    const unsigned int total = a.n_discard + a.n_collect;
    if (noise_rank >= 0) {
        int half = 0;
        for (unsigned int done = 0, nb; done < total; done += nb) {
            nb = gs_batch(done, a.n_discard, total);
            gs_draw *const batch = ring + (size_t)half * GS_RB * 64 + lane;
            for (unsigned int j = (unsigned int)noise_rank; j < nb; j += GS_NN) {
                gs_draw d;
                mm_gibbs_mixture_noise(a.seed, chain, a.iter0 + done + j, &d.z0, &d.u);
                batch[(size_t)j * 64] = d;
            }
            half ^= 1;
            __syncthreads();
        }
        __syncthreads();
        return;
    }
    const bool active = c < a.n_chains;
    double s[2] = {active ? a.state[2 * c] : 0.0, active ? a.state[2 * c + 1] : 0.0};
    unsigned int col = 0, t0 = 0;
    int half = 0;
    for (unsigned int done = 0, nb; done < total; done += nb) {
        nb = gs_batch(done, a.n_discard, total);
        __syncthreads(); /* ring half `half` holds the draws of this batch */
        const gs_draw *const batch = ring + (size_t)half * GS_RB * 64 + lane;
        const bool collecting = done >= a.n_discard && a.out; /* a batch lies on one side of n_discard */
        double *const row = tile + lane * GsTile::stride + 2 * col;
        if (nb == (unsigned int)GS_RB) {
            /* a full batch: its draws come out of the ring in one go (one LDS latency per batch, not one per sweep, in
             * the kernel's one dependent chain), the sweeps are unrolled */
            gs_draw d[GS_RB];
            MM_UNROLL
            for (int k = 0; k < GS_RB; ++k)
                d[k] = batch[(size_t)k * 64];
            MM_UNROLL
            for (int k = 0; k < GS_RB; ++k) {
                mm_gibbs_mixture_step_noise(a.P, s, d[k].z0, d[k].u);
                if (collecting) {
                    row[2 * k] = s[0];
                    row[2 * k + 1] = s[1];
                }
            }
        } else {
            for (unsigned int k = 0; k < nb; ++k) {
                const gs_draw d = batch[(size_t)k * 64];
                mm_gibbs_mixture_step_noise(a.P, s, d.z0, d.u);
                if (collecting) {
                    row[2 * k] = s[0];
                    row[2 * k + 1] = s[1];
                }
            }
        }
        if (collecting) {
            col += nb;
            if (col == (unsigned int)GS_TILE_T || done + nb == total) {
                mm_flush_tile_raw<double, 2, GsTile>(a.out, a.n_collect, a.n_chains, tile, lane, c0, t0, col);
                t0 += col;
                col = 0;
            }
        }
        half ^= 1;
    }
    __syncthreads();
    if (active) {
        a.state[2 * c] = s[0];
        a.state[2 * c + 1] = s[1];
    }
}

} // namespace

struct mmcmc_gibbs_mixture {
    int device = 0;
    size_t n_chains = 0;
    uint64_t seed = 0, chain_offset = 0;
    uint32_t iter = 0;
    int variant = 1; /* 1: noise waves + transition waves (from 4096 chains on); 0: one wave per SIMD (mmcmc_gibbs_mixture_set_kernel_variant) */
    mm_mixture_params P{};
    double *d_state = nullptr;
    hipStream_t stream = nullptr;
};

extern "C" {

int mmcmc_gibbs_mixture_create(mmcmc_gibbs_mixture **out, const double *params, const double *init, size_t n_chains,
                               int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!params || !init || n_chains == 0 || !(params[1] > 0.0) || !(params[3] > 0.0) ||
        !(params[4] >= 0.0 && params[4] <= 1.0))
        return MMCMC_ERR_INVALID_ARG;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= nd)
        return MMCMC_ERR_INVALID_ARG;
    mmcmc_gibbs_mixture *h = new (std::nothrow) mmcmc_gibbs_mixture();
    if (!h)
        return (int)hipErrorOutOfMemory;
    h->device = device;
    h->n_chains = n_chains;
    h->P = mm_mixture_params{params[0], params[1], params[2], params[3], params[4]};
    DevGuard g(device);
    hipError_t e;
    if ((e = hipMalloc((void **)&h->d_state, n_chains * 2 * sizeof(double))) != hipSuccess ||
        (e = hipMemcpy(h->d_state, init, n_chains * 2 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->stream, hipStreamDefault)) != hipSuccess) {
        mmcmc_gibbs_mixture_destroy(h);
        return (int)e;
    }
    *out = h;
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_set_kernel_variant(mmcmc_gibbs_mixture *h, int variant)
{
    if (!h || (variant != 0 && variant != 1))
        return MMCMC_ERR_INVALID_ARG;
    h->variant = variant;
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_seed(mmcmc_gibbs_mixture *h, uint64_t seed)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->seed = seed;
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_set_chain_offset(mmcmc_gibbs_mixture *h, uint64_t off)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    h->chain_offset = off;
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_run(mmcmc_gibbs_mixture *h, size_t n_collect, size_t n_discard, double *out, int out_is_device,
                            void *stream)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if ((uint64_t)h->iter + n_collect + n_discard >= (1ull << 32) || n_collect >= (1ull << 30))
        return MMCMC_ERR_SHAPE;
    DevGuard g(h->device);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t out_bytes = h->n_chains * n_collect * 2 * sizeof(double);
    double *d_out = nullptr;
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
    a.n_chains = h->n_chains;
    a.seed = h->seed;
    a.chain_offset = h->chain_offset;
    a.iter0 = h->iter;
    a.n_discard = (unsigned int)n_discard;
    a.n_collect = (unsigned int)n_collect;
    /* four waves per SIMD from 4096 chains on; variant 0 keeps the one-wave kernel (A/B and tests: the two are bit-identical) */
    hipError_t e = hipSuccess;
    if (h->n_chains >= 4096 && h->variant == 1) {
        static std::atomic<unsigned long long> attr_set{0};
        if (h->device >= 64 || !((attr_set.load(std::memory_order_relaxed) >> h->device) & 1ull)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(mm_gibbs_mixture_split_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS_LDS_BYTES);
            if (e == hipSuccess && h->device < 64)
                attr_set.fetch_or(1ull << h->device, std::memory_order_relaxed);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mm_gibbs_mixture_split_kernel, dim3((unsigned int)((h->n_chains + 255) / 256)),
                               dim3(256 * (1 + GS_NN)), GS_LDS_BYTES, st, a);
            e = hipGetLastError();
        }
    } else {
        hipLaunchKernelGGL(mm_gibbs_mixture_kernel, dim3((unsigned int)((h->n_chains + 63) / 64)), dim3(64), 0, st, a);
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

int mmcmc_gibbs_mixture_state(mmcmc_gibbs_mixture *h, double *out)
{
    if (!h || !out)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    MM_HIP(hipDeviceSynchronize());
    MM_HIP(hipMemcpy(out, h->d_state, h->n_chains * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_sync(mmcmc_gibbs_mixture *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    MM_HIP(hipStreamSynchronize(h->stream));
    return MMCMC_OK;
}

int mmcmc_gibbs_mixture_destroy(mmcmc_gibbs_mixture *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_state);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
    return MMCMC_OK;
}

} /* extern "C" */
