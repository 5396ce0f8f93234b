/*
 * mm_tracker.hip -- running multi-chain diagnostics on the GPU (include/mmcmc.h: mmcmc_tracker_*).
 *
 * Replaces MultiChainTracker (stats.rs:189-306), which run_progress feeds one [n_chains, n_params] state per step
 * (hmc.rs:242-281, core.rs:326-344 through ChainTracker) to show `p(accept)` and `max(rhat)` while sampling.  Here
 * the tracker lives next to the sample in HBM and is fed BLOCKS of consecutive states [n_chains, k, n_params] (a slice
 * of the sampler's output); the host polls it between launches.
 *   step kernel   one lane per chain, the reference's f32 recurrences in the reference's order (stats.rs:244-249):
 *                     mean    = (mean * (n-1) + x) / n
 *                     mean_sq = n == 1 ? x^2 : (mean_sq * (n-1) + x^2) / n
 *                 -- per chain these are exactly the reference's values -- and a flag "this state differs from the
 *                 chain's previous one" per (step, chain).
 *   p_accept      the reference folds p <- (1-a) p + a * flag over the chains of a step IN ORDER, step after step
 *                 (a = 0.01, stats.rs:252-258): one long recurrence.  Its value forgets its start at the rate
 *                 0.99^m, so one lane replays the last 16 384 flags sequentially (0.99^16384 ~ 1e-72): the same f32
 *                 operations in the same order as the reference over the part of the sequence that still matters.
 *   per chain     ChainTracker (stats.rs:26-141, what the generic run_progress feeds, core.rs:90-140): the same means
 *                 plus a per-chain acceptance EMA started from the first comparison of coordinate 0 (stats.rs:109-123,
 *                 quirk Q12); mmcmc_tracker_chain_stats = collect_rhat (stats.rs:150-178, between / (C P - 1): quirk
 *                 Q9) and the average of the per-chain EMAs (core.rs:268-281).
 *   rhat          within_and_var (stats.rs:288-306) from the per-chain means: block reduction with f64 accumulators
 *                 (the reference sums f32 in ndarray's order; parity to ~1e-6 relative, tests/test_tracker.py).
 */
#include "../../include/mmcmc.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <vector>
#include "mm_tuning.h"

#define MM_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return (int)_e;                                                                                       \
    } while (0)

namespace {

constexpr float kAlpha = 0.01f;      /* stats.rs:13 */
constexpr size_t kTail = 16384;      /* flags replayed sequentially for p_accept */
constexpr int kMaxDim = 64;

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

/* states [C, n_rows, D] of T; this call consumes rows t0 .. t0+k-1 of every chain; n_before = steps taken so far */
template <class T>
__global__ void tracker_step_kernel(const T *__restrict__ states, unsigned long long C, unsigned long long n_rows,
                                    unsigned long long t0, unsigned int k, unsigned int D, unsigned long long n_before,
                                    float *__restrict__ mean, float *__restrict__ mean_sq, float *__restrict__ last,
                                    float *__restrict__ p_chain, unsigned char *__restrict__ flags /* [k, C] */)
{
    const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C)
        return;
    float m[kMaxDim], q[kMaxDim], l[kMaxDim];
    float pc = p_chain[c]; /* ChainTracker::p_accept, -1 before the first step */
    for (unsigned int d = 0; d < D; ++d) {
        m[d] = mean[c * D + d];
        q[d] = mean_sq[c * D + d];
        l[d] = last[c * D + d];
    }
    for (unsigned int t = 0; t < k; ++t) {
        const unsigned long long n_i = n_before + t + 1;
        const float n = (float)n_i;
        const T *row = states + (c * n_rows + t0 + t) * D;
        int ne = 0, ne0 = 0;
        for (unsigned int d = 0; d < D; ++d) {
            const float x = (float)row[d];
            m[d] = (m[d] * (n - 1.0f) + x) / n;
            q[d] = (n_i == 1) ? x * x : (q[d] * (n - 1.0f) + x * x) / n;
            const int dif = (x != l[d]) ? 1 : 0;
            ne |= dif;
            if (d == 0)
                ne0 = dif;
            l[d] = x;
        }
        flags[(size_t)t * C + c] = (unsigned char)ne;
        /* stats.rs:109-123: the first step starts the EMA from the comparison of coordinate 0 alone (Q12) */
        const float p_start = (pc >= 0.0f) ? pc : (float)ne0;
        pc = (1.0f - kAlpha) * p_start + kAlpha * (float)ne;
    }
    p_chain[c] = pc;
    for (unsigned int d = 0; d < D; ++d) {
        mean[c * D + d] = m[d];
        mean_sq[c * D + d] = q[d];
        last[c * D + d] = l[d];
    }
}

/* The same step for D <= 8 (f32) / 4 (f64) with coalesced reads.  In the kernel above a wave's load touches 64 chains
 * 4800 bytes apart and uses 12 bytes of every 64-byte sector (measured: 360 GB/s of useful reads at [65536, 400, 3]).
 * Here a wave owns 64 chains and walks the rows in tiles of 16: the tile of a chain is one contiguous run of 16 D
 * elements, the wave fetches the 64 runs as 16-byte pieces (piece p = 64 kk + lane belongs to chain p / V), parks them
 * in registers while the previous tile is consumed, drops them into LDS (row per chain) and each lane then reads its
 * own chain's 16 rows.  Same recurrences in the same order: bit-identical results.  k must be a multiple of 16 (the
 * host sends the remainder through the kernel above). */
template <class T, int D>
__global__ __launch_bounds__(64) void tracker_step_tiled_kernel(const T *__restrict__ states, unsigned long long C,
                                                                unsigned long long n_rows, unsigned long long t0,
                                                                unsigned int k, unsigned long long n_before,
                                                                float *__restrict__ mean, float *__restrict__ mean_sq,
                                                                float *__restrict__ last, float *__restrict__ p_chain,
                                                                unsigned char *__restrict__ flags /* [k, C] */)
{
    constexpr int TT = 16, EPL = 16 / (int)sizeof(T), RUN = TT * D, V = RUN / EPL, PITCH = RUN + EPL;
    typedef T vec16 __attribute__((ext_vector_type(EPL), aligned(sizeof(T))));
    typedef T vec16a __attribute__((ext_vector_type(EPL)));
    __shared__ __attribute__((aligned(16))) T tile[64 * PITCH];
    const int lane = threadIdx.x;
    const unsigned long long wave_c0 = (unsigned long long)blockIdx.x * 64, c = wave_c0 + lane;
    const bool active = c < C;
    const unsigned int n_valid = (unsigned int)(C - wave_c0 < 64 ? C - wave_c0 : 64);
    float m[D], q[D], l[D];
    float pc = active ? p_chain[c] : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        m[d] = active ? mean[c * D + d] : 0.f;
        q[d] = active ? mean_sq[c * D + d] : 0.f;
        l[d] = active ? last[c * D + d] : 0.f;
    }
    vec16 pre[V];
    auto request = [&](unsigned int trow) {
#pragma unroll
        for (int kk = 0; kk < V; ++kk) {
            const int p = kk * 64 + lane, j = p / V, e = (p - j * V) * EPL;
            vec16 v = {};
            if ((unsigned int)j < n_valid)
                v = *reinterpret_cast<const vec16 *>(states + ((wave_c0 + j) * n_rows + t0 + trow) * D + e);
            pre[kk] = v;
        }
    };
    request(0);
    for (unsigned int tr = 0; tr < k; tr += TT) {
#pragma unroll
        for (int kk = 0; kk < V; ++kk) {
            const int p = kk * 64 + lane, j = p / V, e = (p - j * V) * EPL;
            *reinterpret_cast<vec16a *>(tile + j * PITCH + e) = pre[kk];
        }
        if (tr + TT < k)
            request(tr + TT);
        __builtin_amdgcn_wave_barrier(); /* LDS operations of one wave execute in order */
        if (active) {
            const T *rows = tile + lane * PITCH;
#pragma unroll 4
            for (int t = 0; t < TT; ++t) {
                const unsigned long long n_i = n_before + tr + t + 1;
                const float n = (float)n_i;
                int ne = 0, ne0 = 0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float x = (float)rows[t * D + d];
                    m[d] = (m[d] * (n - 1.0f) + x) / n;
                    q[d] = (n_i == 1) ? x * x : (q[d] * (n - 1.0f) + x * x) / n;
                    const int dif = (x != l[d]) ? 1 : 0;
                    ne |= dif;
                    if (d == 0)
                        ne0 = dif;
                    l[d] = x;
                }
                flags[(size_t)(tr + t) * C + c] = (unsigned char)ne;
                const float p_start = (pc >= 0.0f) ? pc : (float)ne0; /* stats.rs:109-123 (Q12) */
                pc = (1.0f - kAlpha) * p_start + kAlpha * (float)ne;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (active) {
        p_chain[c] = pc;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            mean[c * D + d] = m[d];
            mean_sq[c * D + d] = q[d];
            last[c * D + d] = l[d];
        }
    }
}

/* The same step with the parameters of a chain spread over the waves of a workgroup: 64 chains x D waves, wave d keeps
 * (mean, mean of squares, last value) of parameter d of its lane's chain.  The kernel above is bound by instruction issue,
 * not by memory -- 107 vector instructions per row (two IEEE divisions per parameter) at the 5 cycles a lone wave per SIMD
 * gets: 180 us for 315 MB -- and 65 536 chains are one wave per SIMD; here they are D, issuing side by side (2.2 cycles
 * per instruction and SIMD).  The tile is fetched by all 64 D threads (V / D 16-byte pieces each), a wave
 * reads column d of its lane's 16 rows, the "this row differs from the last" bits of the D parameters meet in LDS and
 * wave 0 folds them into the chain's acceptance average and the flags.  Same recurrences in the same order:
 * bit-identical to tracker_step_kernel. */
template <class T, int D>
__global__ __launch_bounds__(64 * (D + 1)) void tracker_step_dims_kernel(const T *__restrict__ states, unsigned long long C,
                                                                         unsigned long long n_rows, unsigned long long t0,
                                                                         unsigned int k, unsigned long long n_before,
                                                                         float *__restrict__ mean, float *__restrict__ mean_sq,
                                                                         float *__restrict__ last, float *__restrict__ p_chain,
                                                                         unsigned char *__restrict__ flags /* [k, C] */)
{
    constexpr int TT = 16, EPL = 16 / (int)sizeof(T), RUN = TT * D, V = RUN / EPL, PITCH = RUN + EPL;
    constexpr int PER = V / D; /* 16-byte pieces of a tile per thread of the D parameter waves: 4 (f32), 8 (f64) */
    static_assert(V % D == 0, "pieces per thread");
    typedef T vec16 __attribute__((ext_vector_type(EPL), aligned(sizeof(T))));
    typedef T vec16a __attribute__((ext_vector_type(EPL)));
    __shared__ __attribute__((aligned(16))) T tile[64 * PITCH];
    /* per tile and chain: bit t of difs[b][d][lane] = "parameter d of row t differs from the row before"; two buffers: the
     * bookkeeping wave folds tile i while the parameter waves fill tile i + 1 */
    __shared__ __attribute__((aligned(16))) unsigned int difs[2][D + 1][64];
    const int lane = threadIdx.x & 63, dim = threadIdx.x >> 6;
    const unsigned long long wave_c0 = (unsigned long long)blockIdx.x * 64, c = wave_c0 + lane;
    const bool active = c < C;
    const unsigned int n_valid = (unsigned int)(C - wave_c0 < 64 ? C - wave_c0 : 64);
    const unsigned int n_tiles = k / TT;
    if (dim == D) {
        /* ---- the bookkeeping wave: acceptance average of the chain (a sequential recurrence over the rows) and the flags;
         * one tile behind the parameter waves, beside their next tile's rows (it used to be wave 0's tail: every other wave
         * of the workgroup waited for it, tools/experiments/tracker_probe.hip) ---- */
        float pc = active ? p_chain[c] : 0.f;
        for (unsigned int i = 0; i <= n_tiles; ++i) {
            if (i < n_tiles)
                __syncthreads(); /* A_i: tile i staged (nothing to do with it here) */
            if (i > 0) {
                const unsigned int b = (i - 1) & 1u, tr = (i - 1) * TT;
                const unsigned int dif0 = difs[b][0][lane];
                unsigned int any = dif0;
#pragma unroll
                for (int d = 1; d < D; ++d)
                    any |= difs[b][d][lane];
                if (n_valid == 64u && (C & 15ull) == 0ull) {
                    /* the tile's flags, 16 rows x 64 chains, leave as ONE 16-byte store per lane instead of 16 one-byte
                     * stores: lane 4 r + g writes the flags of row r, chains 16 g .. 16 g + 15, assembled from those
                     * chains' 16-bit masks */
                    difs[b][D][lane] = any;
                    __builtin_amdgcn_wave_barrier(); /* LDS operations of one wave execute in order */
                    const int r = lane >> 2, g = lane & 3;
                    unsigned int w[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const uint4 mk = *reinterpret_cast<const uint4 *>(&difs[b][D][16 * g + 4 * q4]);
                        w[q4] = ((mk.x >> r) & 1u) | (((mk.y >> r) & 1u) << 8) | (((mk.z >> r) & 1u) << 16) | (((mk.w >> r) & 1u) << 24);
                    }
                    *reinterpret_cast<uint4 *>(flags + (size_t)(tr + r) * C + wave_c0 + 16 * g) = make_uint4(w[0], w[1], w[2], w[3]);
                } else if (active) {
#pragma unroll 4
                    for (int t = 0; t < TT; ++t)
                        flags[(size_t)(tr + t) * C + c] = (unsigned char)((any >> t) & 1u);
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        const int ne = (int)((any >> t) & 1u), ne0 = (int)((dif0 >> t) & 1u);
                        const float p_start = (pc >= 0.0f) ? pc : (float)ne0; /* stats.rs:109-123 (Q12) */
                        pc = (1.0f - kAlpha) * p_start + kAlpha * (float)ne;
                    }
                }
            }
            if (i < n_tiles)
                __syncthreads(); /* B_i: the bits of tile i are in */
        }
        if (active)
            p_chain[c] = pc;
        return;
    }
    /* ---- parameter wave `dim` ---- */
    float m = active ? mean[c * D + dim] : 0.f, q = active ? mean_sq[c * D + dim] : 0.f, l = active ? last[c * D + dim] : 0.f;
    vec16 pre[PER];
    auto request = [&](unsigned int trow) {
#pragma unroll
        for (int kk = 0; kk < PER; ++kk) {
            const int p = kk * 64 * D + (int)threadIdx.x, j = p / V, e = (p - j * V) * EPL;
            vec16 v = {};
            if ((unsigned int)j < n_valid)
                v = *reinterpret_cast<const vec16 *>(states + ((wave_c0 + j) * n_rows + t0 + trow) * D + e);
            pre[kk] = v;
        }
    };
    request(0);
    for (unsigned int i = 0; i < n_tiles; ++i) {
        const unsigned int tr = i * TT;
#pragma unroll
        for (int kk = 0; kk < PER; ++kk) {
            const int p = kk * 64 * D + (int)threadIdx.x, j = p / V, e = (p - j * V) * EPL;
            *reinterpret_cast<vec16a *>(tile + j * PITCH + e) = pre[kk];
        }
        if (i + 1 < n_tiles)
            request(tr + TT);
        __syncthreads(); /* A_i: the tile is complete */
        unsigned int dif = 0;
        {
            const T *rows = tile + lane * PITCH + dim;
#pragma unroll 4
            for (int t = 0; t < TT; ++t) {
                const unsigned long long n_i = n_before + tr + t + 1;
                const float n = (float)n_i;
                const float x = (float)rows[t * D];
                m = (m * (n - 1.0f) + x) / n;
                q = (n_i == 1) ? x * x : (q * (n - 1.0f) + x * x) / n;
                dif |= (x != l) ? (1u << t) : 0u;
                l = x;
            }
        }
        difs[i & 1u][dim][lane] = dif;
        __syncthreads(); /* B_i: the bits of every parameter are in; nobody reads the tile any more */
    }
    if (active) {
        mean[c * D + dim] = m;
        mean_sq[c * D + dim] = q;
        last[c * D + dim] = l;
    }
}

template <class T>
static bool launch_step_tiled(int D, unsigned int grid, hipStream_t st, const T *states, unsigned long long C,
                              unsigned long long n_rows, unsigned long long t0, unsigned int k,
                              unsigned long long n_before, float *mean, float *mean_sq, float *last, float *p_chain,
                              unsigned char *flags)
{
    /* one wave per 64 chains (all parameters in one lane) only as the measurement's reference: MMCMC_TRACKER_ONE_WAVE */
    const bool one_wave = mm_tuning_env("MMCMC_TRACKER_ONE_WAVE") != nullptr;
#define MM_TILED(DD)                                                                                              \
    case DD:                                                                                                      \
        if (DD * sizeof(T) <= 32) {                                                                               \
            if (DD > 1 && !one_wave)                                                                              \
                hipLaunchKernelGGL((tracker_step_dims_kernel<T, (DD * sizeof(T) <= 32 ? DD : 1)>), dim3(grid),   \
                                   dim3(64 * (DD + 1)), 0, st, states, C, n_rows, t0, k, n_before, mean, mean_sq, last, \
                                   p_chain, flags);                                                               \
            else                                                                                                  \
                hipLaunchKernelGGL((tracker_step_tiled_kernel<T, (DD * sizeof(T) <= 32 ? DD : 1)>), dim3(grid),  \
                                   dim3(64), 0, st, states, C, n_rows, t0, k, n_before, mean, mean_sq, last,      \
                                   p_chain, flags);                                                               \
            return true;                                                                                          \
        }                                                                                                         \
        return false;
    switch (D) {
        MM_TILED(1)
        MM_TILED(2)
        MM_TILED(3)
        MM_TILED(4)
        MM_TILED(5)
        MM_TILED(6)
        MM_TILED(7)
        MM_TILED(8)
    default:
        return false;
    }
#undef MM_TILED
}

/* sequential replay of the last `len` <= kTail flags (time-major, chain-minor order).  One wave.  Lane l first loads
 * its contiguous share of 256 flags into eight words of bits (16-byte loads when the share is aligned).  The recurrence
 * itself runs in the LANES: every lane carries one value of p through the same flags, which are wave-uniform (read from
 * their owner with v_readlane, the bit tests and the addend alpha / 0 on the scalar unit) -- lane 0 starts from p = 0,
 * lane 1 from p = 1: the two replays of the certificate cost one multiply and one add per flag together, where they were
 * two of each plus the bit's extraction and conversion in the one lane that owned the flags (0.097 -> 0.04 ms for the
 * 4096 flags of the certificate, round 3). */
template <bool FULL>
__device__ __forceinline__ float tracker_fold_block(float p, const unsigned int (&w)[kTail / 64 / 32], unsigned int n)
{
#pragma unroll
    for (unsigned int i = 0; i < kTail / 64 / 32; ++i) {
#pragma unroll
        for (unsigned int b = 0; b < 32; ++b)
            if (FULL || i * 32 + b < n) {
                const float a = ((w[i] >> b) & 1u) ? kAlpha : 0.0f; /* = kAlpha * accepted, exactly */
                p = (1.0f - kAlpha) * p + a;                         /* fl(fl((1 - a) p) + fl(a f)): stats.rs:113-123 */
            }
    }
    return p;
}

__global__ __launch_bounds__(64) void tracker_paccept_kernel(const unsigned char *__restrict__ flags, size_t first,
                                                             size_t len, int restart, float *p_accept)
{
    constexpr unsigned int per = (unsigned int)(kTail / 64), NW = per / 32; /* flags, words of bits per lane */
    const unsigned int lane = threadIdx.x;
    unsigned int bits[NW];
    const size_t lo = (size_t)lane * per;
    const unsigned int mine = lo < len ? (unsigned int)((len - lo) < per ? (len - lo) : per) : 0u;
    if (mine == per && ((reinterpret_cast<size_t>(flags) + first) & 15u) == 0u) {
        /* 16 flags per load; four 0 / 1 bytes of a word become four bits by one multiply (the partial products land on
         * distinct bits, the wanted ones on 28..31) */
        const uint4 *src = reinterpret_cast<const uint4 *>(flags + first + lo);
#pragma unroll
        for (unsigned int w = 0; w < NW; ++w) {
            const uint4 q0 = src[2 * w], q1 = src[2 * w + 1];
            const unsigned int d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            unsigned int v = 0;
#pragma unroll
            for (unsigned int k = 0; k < 8; ++k)
                v |= (((d[k] & 0x01010101u) * 0x10204080u) >> 28) << (4 * k);
            bits[w] = v;
        }
    } else {
#pragma unroll
        for (unsigned int w = 0; w < NW; ++w) {
            unsigned int v = 0;
            for (unsigned int b = 0; b < 32; ++b) {
                const size_t i = lo + w * 32 + b;
                if (i < len && flags[first + i])
                    v |= 1u << b;
            }
            bits[w] = v;
        }
    }
    /* the flags of lane l's share, wave-uniform */
    auto fold_share = [&](float p, unsigned int l) -> float {
        unsigned int w[NW];
#pragma unroll
        for (unsigned int i = 0; i < NW; ++i)
            w[i] = (unsigned int)__builtin_amdgcn_readlane((int)bits[i], (int)l);
        const unsigned int n = (unsigned int)__builtin_amdgcn_readlane((int)mine, (int)l);
        return n == per ? tracker_fold_block<true>(p, w, n) : tracker_fold_block<false>(p, w, n);
    };
    /* A certificate first: one step, fl(fl((1 - a) p) + fl(a f)), is a monotone map of p, so is any run of steps, and the
     * true value lies in [0, 1].  Replay only the last kCert flags from p = 0 and from p = 1: if the two arrive at the same
     * number, every start in between does -- that number IS the sequential result over the whole history, bit for bit,
     * whatever came before.  The map contracts by 0.99 per flag ((0.99)^4096 = 1e-18), so this succeeds in practice always
     * (not on a run of equal flags: several f32 fixed points); if it does not, the full replay below decides. */
    const unsigned int last_lane = len ? (unsigned int)((len - 1) / per) : 0u;
    {
        constexpr unsigned int kCertLanes = 16; /* 16 x 256 = 4096 flags */
        const unsigned int l0 = last_lane + 1u > kCertLanes ? last_lane + 1u - kCertLanes : 0u;
        float p = lane == 1u ? 1.0f : 0.0f;
        for (unsigned int l = l0; l <= last_lane; ++l)
            p = fold_share(p, l);
        const float plo = __shfl(p, 0, 64), phi = __shfl(p, 1, 64);
        if (len && plo == phi) { /* wave-uniform */
            if (lane == 0)
                *p_accept = plo;
            return;
        }
    }
    float p = restart ? 0.5f : *p_accept; /* restart: the sequence before `first` is forgotten anyway */
    for (unsigned int l = 0; l <= last_lane && len; ++l)
        p = fold_share(p, l);
    if (lane == 0)
        *p_accept = p;
}

/* within_and_var: out[d] = rhat; one block per parameter */
__global__ __launch_bounds__(1024) void tracker_rhat_kernel(const float *__restrict__ mean,
                                                            const float *__restrict__ mean_sq, unsigned long long C,
                                                            unsigned int D, unsigned long long n_i, float *out)
{
    __shared__ double red[1024];
    const unsigned int d = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](double v) -> double {
        red[tid] = v;
        __syncthreads();
        for (unsigned int s = 512; s > 0; s >>= 1) {
            if (tid < s)
                red[tid] += red[tid + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    const float n = (float)n_i, nch = (float)C;
    double s = 0.0;
    for (unsigned long long c = tid; c < C; c += 1024)
        s += (double)mean[c * D + d];
    const float mean_chain = (float)(block_sum(s) / (double)nch);
    double b = 0.0, w = 0.0;
    for (unsigned long long c = tid; c < C; c += 1024) {
        const float mc = mean[c * D + d];
        const float df = mc - mean_chain;
        b += (double)(df * df);
        const float sm2 = (mean_sq[c * D + d] - mc * mc) * n / (n - 1.0f);
        w += (double)sm2;
    }
    const float between = (float)block_sum(b) * (n / (nch - 1.0f));
    const float within = (float)(block_sum(w) / (double)nch);
    if (tid == 0) {
        const float var = within * ((n - 1.0f) / n) + between * (1.0f / n);
        out[d] = sqrtf(var / within);
    }
}

/* collect_rhat (stats.rs:150-178) over the ChainStats of every chain, and the average per-chain acceptance EMA
 * (core.rs:268-281); out[d] = rhat, out[D] = mean p_accept, out[D + 1 + d] = within, out[2 D + 1 + d] = var.  One block per parameter (+ one for p_accept). */
__global__ __launch_bounds__(1024) void tracker_chain_stats_kernel(const float *__restrict__ mean,
                                                                   const float *__restrict__ mean_sq,
                                                                   const float *__restrict__ p_chain,
                                                                   unsigned long long C, unsigned int D,
                                                                   unsigned long long n_i, float *out)
{
    __shared__ double red[1024];
    const unsigned int d = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](double v) -> double {
        red[tid] = v;
        __syncthreads();
        for (unsigned int s = 512; s > 0; s >>= 1) {
            if (tid < s)
                red[tid] += red[tid + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    if (d == D) {
        double p = 0.0;
        for (unsigned long long c = tid; c < C; c += 1024)
            p += (double)p_chain[c];
        p = block_sum(p);
        if (tid == 0)
            out[D] = (float)(p / (double)C);
        return;
    }
    const float n = (float)n_i, nch = (float)C;
    double sm = 0.0, sw = 0.0;
    for (unsigned long long c = tid; c < C; c += 1024) {
        const float mc = mean[c * D + d];
        sm += (double)mc;
        sw += (double)((mean_sq[c * D + d] - mc * mc) * n / (n - 1.0f)); /* ChainTracker::stats sm2 (stats.rs:135-138) */
    }
    const float global_mean = (float)(block_sum(sm) / (double)nch);
    const float within = (float)(block_sum(sw) / (double)nch);
    double ss = 0.0;
    for (unsigned long long c = tid; c < C; c += 1024) {
        const float df = mean[c * D + d] - global_mean;
        ss += (double)(df * df);
    }
    const float between = (float)block_sum(ss) / (float)(C * D - 1); /* diffs.len() - 1: quirk Q9 */
    if (tid == 0) {
        const float var = between + within * ((n - 1.0f) / n);
        out[d] = sqrtf(var / within);
        out[D + 1 + d] = within; /* withinvar_from_cs's pair (stats.rs:155-178), for ess_from_chainstats */
        out[2 * D + 1 + d] = var;
    }
}

} // namespace

struct mmcmc_tracker {
    int device = 0;
    size_t n_chains = 0, dim = 0;
    unsigned long long n = 0; /* steps taken */
    float *d_mean = nullptr, *d_mean_sq = nullptr, *d_last = nullptr, *d_p = nullptr, *d_rhat = nullptr;
    float *d_p_chain = nullptr; /* [C] ChainTracker::p_accept */
    unsigned char *d_flags = nullptr;
    size_t flags_cap = 0;
};

extern "C" {

int mmcmc_tracker_create(mmcmc_tracker **out, size_t n_chains, size_t dim, int device)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_chains < 2 || dim == 0 || dim > (size_t)kMaxDim)
        return MMCMC_ERR_INVALID_ARG;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= nd)
        return MMCMC_ERR_INVALID_ARG;
    mmcmc_tracker *h = new (std::nothrow) mmcmc_tracker();
    if (!h)
        return (int)hipErrorOutOfMemory;
    h->device = device;
    h->n_chains = n_chains;
    h->dim = dim;
    DevGuard g(device);
    const size_t cd = n_chains * dim * sizeof(float);
    hipError_t e = hipSuccess;
    auto alloc0 = [&](float **p, size_t bytes) {
        if (e == hipSuccess)
            e = hipMalloc((void **)p, bytes);
        if (e == hipSuccess)
            e = hipMemset(*p, 0, bytes);
    };
    /* MultiChainTracker::new (stats.rs:216-228): n = 0, p_accept = 0, everything else zero */
    alloc0(&h->d_mean, cd);
    alloc0(&h->d_mean_sq, cd);
    alloc0(&h->d_last, cd);
    alloc0(&h->d_p, sizeof(float));
    alloc0(&h->d_rhat, (3 * dim + 1) * sizeof(float));
    alloc0(&h->d_p_chain, n_chains * sizeof(float));
    if (e == hipSuccess) {
        std::vector<float> neg(n_chains, -1.0f); /* ChainTracker::new: p_accept = -1 (stats.rs:76) */
        e = hipMemcpy(h->d_p_chain, neg.data(), n_chains * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        mmcmc_tracker_destroy(h);
        return (int)e;
    }
    *out = h;
    return MMCMC_OK;
}

int mmcmc_tracker_steps(mmcmc_tracker *h, const void *states, int states_is_device, int dtype, size_t n_rows, size_t t0,
                        size_t k, void *stream)
{
    if (!h || !states || (dtype != MMCMC_F32 && dtype != MMCMC_F64) || k == 0 || t0 + k > n_rows)
        return MMCMC_ERR_INVALID_ARG;
    if (k >= (1ull << 31))
        return MMCMC_ERR_SHAPE;
    DevGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = dtype == MMCMC_F32 ? 4 : 8;
    const void *d_states = states;
    void *tmp = nullptr;
    if (!states_is_device) {
        const size_t bytes = h->n_chains * n_rows * h->dim * esz;
        MM_HIP(hipMalloc(&tmp, bytes));
        MM_HIP(hipMemcpyAsync(tmp, states, bytes, hipMemcpyHostToDevice, st));
        d_states = tmp;
    }
    const size_t need = k * h->n_chains;
    if (need > h->flags_cap) {
        MM_HIP(hipStreamSynchronize(st));
        (void)hipFree(h->d_flags);
        h->d_flags = nullptr;
        h->flags_cap = 0;
        MM_HIP(hipMalloc((void **)&h->d_flags, need));
        h->flags_cap = need;
    }
    /* rows in tiles of 16 through the coalescing kernel where it exists, the remainder lane by lane */
    unsigned int k_tiled = (unsigned int)(k - k % 16);
    if (k_tiled) {
        const unsigned int grid64 = (unsigned int)((h->n_chains + 63) / 64);
        const bool ok = dtype == MMCMC_F32
                            ? launch_step_tiled<float>((int)h->dim, grid64, st, (const float *)d_states, h->n_chains, n_rows, t0,
                                                       k_tiled, h->n, h->d_mean, h->d_mean_sq, h->d_last, h->d_p_chain, h->d_flags)
                            : launch_step_tiled<double>((int)h->dim, grid64, st, (const double *)d_states, h->n_chains, n_rows,
                                                        t0, k_tiled, h->n, h->d_mean, h->d_mean_sq, h->d_last, h->d_p_chain,
                                                        h->d_flags);
        if (!ok)
            k_tiled = 0;
    }
    const unsigned int k_rest = (unsigned int)k - k_tiled;
    if (k_rest) {
        const unsigned int grid = (unsigned int)((h->n_chains + 255) / 256);
        unsigned char *fl = h->d_flags + (size_t)k_tiled * h->n_chains;
        if (dtype == MMCMC_F32)
            hipLaunchKernelGGL(tracker_step_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)d_states,
                               (unsigned long long)h->n_chains, (unsigned long long)n_rows,
                               (unsigned long long)(t0 + k_tiled), k_rest, (unsigned int)h->dim, h->n + k_tiled, h->d_mean,
                               h->d_mean_sq, h->d_last, h->d_p_chain, fl);
        else
            hipLaunchKernelGGL(tracker_step_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)d_states,
                               (unsigned long long)h->n_chains, (unsigned long long)n_rows,
                               (unsigned long long)(t0 + k_tiled), k_rest, (unsigned int)h->dim, h->n + k_tiled, h->d_mean,
                               h->d_mean_sq, h->d_last, h->d_p_chain, fl);
    }
    MM_HIP(hipGetLastError());
    const size_t len = need < kTail ? need : kTail;
    hipLaunchKernelGGL(tracker_paccept_kernel, dim3(1), dim3(64), 0, st, h->d_flags, need - len, len,
                       need > kTail ? 1 : 0, h->d_p);
    MM_HIP(hipGetLastError());
    h->n += k;
    if (tmp) {
        MM_HIP(hipStreamSynchronize(st));
        MM_HIP(hipFree(tmp));
    }
    return MMCMC_OK;
}

int mmcmc_tracker_stats(mmcmc_tracker *h, float *rhat, float *max_rhat, float *p_accept, void *stream)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if (h->n < 2)
        return MMCMC_ERR_STATE; /* n / (n - 1) */
    DevGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tracker_rhat_kernel, dim3((unsigned int)h->dim), dim3(1024), 0, st, h->d_mean, h->d_mean_sq,
                       (unsigned long long)h->n_chains, (unsigned int)h->dim, h->n, h->d_rhat);
    MM_HIP(hipGetLastError());
    std::vector<float> r(h->dim);
    float p = 0.f;
    MM_HIP(hipMemcpyAsync(r.data(), h->d_rhat, h->dim * sizeof(float), hipMemcpyDeviceToHost, st));
    MM_HIP(hipMemcpyAsync(&p, h->d_p, sizeof(float), hipMemcpyDeviceToHost, st));
    MM_HIP(hipStreamSynchronize(st));
    float mx = r[0];
    for (size_t d = 0; d < h->dim; ++d) {
        if (rhat)
            rhat[d] = r[d];
        mx = r[d] > mx ? r[d] : mx; /* max_rhat (stats.rs:270-274) */
    }
    if (max_rhat)
        *max_rhat = mx;
    if (p_accept)
        *p_accept = p;
    return MMCMC_OK;
}

int mmcmc_tracker_init_last(mmcmc_tracker *h, const void *states, int states_is_device, int dtype, void *stream)
{
    if (!h || !states || (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    if (h->n != 0)
        return MMCMC_ERR_STATE;
    DevGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t cd = h->n_chains * h->dim;
    std::vector<float> f(cd);
    if (dtype == MMCMC_F32) {
        MM_HIP(hipMemcpyAsync(f.data(), states, cd * sizeof(float), states_is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, st));
        MM_HIP(hipStreamSynchronize(st));
    } else {
        std::vector<double> d(cd);
        MM_HIP(hipMemcpyAsync(d.data(), states, cd * sizeof(double), states_is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, st));
        MM_HIP(hipStreamSynchronize(st));
        for (size_t i = 0; i < cd; ++i)
            f[i] = (float)d[i]; /* to_f32 (stats.rs:66-71) */
    }
    MM_HIP(hipMemcpyAsync(h->d_last, f.data(), cd * sizeof(float), hipMemcpyHostToDevice, st));
    MM_HIP(hipStreamSynchronize(st));
    return MMCMC_OK;
}

int mmcmc_tracker_chain_stats(mmcmc_tracker *h, float *rhat, float *max_rhat, float *avg_p_accept, void *stream)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if (h->n < 2)
        return MMCMC_ERR_STATE;
    DevGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tracker_chain_stats_kernel, dim3((unsigned int)h->dim + 1), dim3(1024), 0, st, h->d_mean,
                       h->d_mean_sq, h->d_p_chain, (unsigned long long)h->n_chains, (unsigned int)h->dim, h->n, h->d_rhat);
    MM_HIP(hipGetLastError());
    std::vector<float> r(h->dim + 1);
    MM_HIP(hipMemcpyAsync(r.data(), h->d_rhat, (h->dim + 1) * sizeof(float), hipMemcpyDeviceToHost, st));
    MM_HIP(hipStreamSynchronize(st));
    float mx = r[0];
    for (size_t d = 0; d < h->dim; ++d) {
        if (rhat)
            rhat[d] = r[d];
        if (!(r[d] != r[d]) && (mx != mx || r[d] > mx))
            mx = r[d]; /* max_skipnan (core.rs:292) */
    }
    if (max_rhat)
        *max_rhat = mx;
    if (avg_p_accept)
        *avg_p_accept = r[h->dim];
    return MMCMC_OK;
}

/* withinvar_from_cs (stats.rs:155-178) over the per-chain ChainStats: what collect_rhat and ess_from_chainstats share */
int mmcmc_tracker_within_var(mmcmc_tracker *h, float *within, float *var, void *stream)
{
    if (!h || !within || !var)
        return MMCMC_ERR_INVALID_ARG;
    if (h->n < 2)
        return MMCMC_ERR_STATE;
    DevGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tracker_chain_stats_kernel, dim3((unsigned int)h->dim + 1), dim3(1024), 0, st, h->d_mean,
                       h->d_mean_sq, h->d_p_chain, (unsigned long long)h->n_chains, (unsigned int)h->dim, h->n, h->d_rhat);
    MM_HIP(hipGetLastError());
    std::vector<float> r(2 * h->dim);
    MM_HIP(hipMemcpyAsync(r.data(), h->d_rhat + h->dim + 1, 2 * h->dim * sizeof(float), hipMemcpyDeviceToHost, st));
    MM_HIP(hipStreamSynchronize(st));
    for (size_t d = 0; d < h->dim; ++d) {
        within[d] = r[d];
        var[d] = r[h->dim + d];
    }
    return MMCMC_OK;
}

int mmcmc_tracker_shape(mmcmc_tracker *h, size_t *n_chains, size_t *dim, int *device)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    if (n_chains)
        *n_chains = h->n_chains;
    if (dim)
        *dim = h->dim;
    if (device)
        *device = h->device;
    return MMCMC_OK;
}

int mmcmc_tracker_n(mmcmc_tracker *h, uint64_t *n)
{
    if (!h || !n)
        return MMCMC_ERR_INVALID_ARG;
    *n = h->n;
    return MMCMC_OK;
}

int mmcmc_tracker_destroy(mmcmc_tracker *h)
{
    if (!h)
        return MMCMC_ERR_INVALID_ARG;
    DevGuard g(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->d_mean);
    (void)hipFree(h->d_mean_sq);
    (void)hipFree(h->d_last);
    (void)hipFree(h->d_p);
    (void)hipFree(h->d_rhat);
    (void)hipFree(h->d_flags);
    (void)hipFree(h->d_p_chain);
    delete h;
    return MMCMC_OK;
}

} /* extern "C" */
