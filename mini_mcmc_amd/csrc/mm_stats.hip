/*
 * mm_stats.hip -- split-R-hat / ESS reduction on the GPU (C ABI: mmcmc_stats_*, mmcmc_split_rhat_mean_ess, ...).
 *
 * Replaces stats.rs:416-546 (`split_rhat_mean_ess`: splitcat :396-402, withinvar :429-477, rhat :425-427,
 * ess :496-546, autocov_bf :632-654 / autocov_fft :576-620) and the summaries basic_stats :310-336 /
 * RunStats :339-371.  In the reference the per-chain autocovariances are computed one chain after another on the
 * host (the chain loop :499-504 is serial) and averaged; here
 *
 *   kernel 1 (mm_half_chain_mfma_kernel, or mm_half_chain_kernel when a half-chain does not fit the MFMA kernel's
 *      LDS layout): one wave per half-chain.  The [m, D] block of a half-chain (m = n/2 rows) is
 *      read ONCE from HBM with coalesced loads, transposed into LDS ([D][m], zero padded), reduced to the
 *      half-chain's mean and centred sum of squares per parameter, centred in place, and its biased
 *      autocovariance sum_t y[t] y[t+lag] is accumulated for every lag into a per-wave LDS slab that is carried
 *      across all the half-chains the wave processes; the slab is written out once per wave.
 *   kernel 2 (mm_stats_tail_kernel): sums the per-wave slabs in a fixed order (bitwise reproducible, no atomics) and,
 *     in other blocks of the same launch, the cross-chain sums of withinvar.
 *   host finish (mm_stats_finish): W, B, var+, the reference's "R-hat" sqrt(W/var+) (quirk Q7), rho_t, Geyer's
 *      initial monotone sequence and ESS exactly as stats.rs:449-465, :509-545 order them, in f32.
 *
 * The autocovariance is evaluated directly (the reference does the same for m <= 100 and switches to an FFT above;
 * both compute the same biased, mean-removed estimator).  Only sufficient statistics leave the GPU: 2*C*D means
 * and sums of squares and m*D lag sums -- which is also all a multi-GPU run has to exchange (all-gather of the
 * former, all-reduce of the latter; mini_mcmc_amd/stats.py).
 */
#include "../../include/mmcmc.h"
#include "mm_stats_fft.h"
#include "mm_tuning.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <vector>

#define MM_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return (int)_e;                                                                                       \
    } while (0)

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

/* the same total with data-parallel-primitive adds instead of six LDS permutes: pairs, quads, half rows, rows of 16 (every
 * lane of a row then holds the row's sum), then the four rows */
__device__ __forceinline__ float wave_sum_dpp(float v)
{
#define MM_DPP_ADD(x, ctrl) ((x) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), 0xf, 0xf, false)))
    v = MM_DPP_ADD(v, 0xB1);  /* quad_perm [1, 0, 3, 2] */
    v = MM_DPP_ADD(v, 0x4E);  /* quad_perm [2, 3, 0, 1] */
    v = MM_DPP_ADD(v, 0x141); /* row_half_mirror */
    v = MM_DPP_ADD(v, 0x140); /* row_mirror */
#undef MM_DPP_ADD
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

/* the same total, bit for bit ((r0 + r1) + (r2 + r3) with the sums commuted), with the rows combined by two row-broadcast
 * adds and one readlane instead of four readlanes, two moves and three adds: lane 15 of rows 0 / 2 into rows 1 / 3, then
 * lane 31 into rows 2 and 3; lane 63 holds (r3 + r2) + (r1 + r0) */
__device__ __forceinline__ float wave_sum_dpp_bcast(float v)
{
#define MM_DPP_ADD(x, ctrl, rows) ((x) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), (rows), 0xf, false)))
    v = MM_DPP_ADD(v, 0xB1, 0xf);  /* quad_perm [1, 0, 3, 2] */
    v = MM_DPP_ADD(v, 0x4E, 0xf);  /* quad_perm [2, 3, 0, 1] */
    v = MM_DPP_ADD(v, 0x141, 0xf); /* row_half_mirror */
    v = MM_DPP_ADD(v, 0x140, 0xf); /* row_mirror */
    v = MM_DPP_ADD(v, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */
    v = MM_DPP_ADD(v, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3 */
#undef MM_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

/* sample [C, n, D] of T (f32 or f64; converted to f32 on load like RunStats::from, stats.rs:365).
 * half-chain index hc in [0, 2C): hc < C -> rows [0, m) of chain hc ; else rows [n-m, n) of chain hc-C.
 * means / ssq: [2C, D];  slabs: [n_waves, D, m] un-normalised lag sums over the half-chains of each wave. */
template <class T>
__global__ __launch_bounds__(64) void mm_half_chain_kernel(const T *__restrict__ sample, unsigned long long C,
                                                           unsigned int n, unsigned int D, unsigned int m,
                                                           unsigned int m_pad, float *__restrict__ means,
                                                           float *__restrict__ ssq, float *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *y = lds;                        /* [D][m_pad], rows zero padded past m */
    float *acc = lds + (size_t)D * m_pad;  /* [D][m] lag sums carried over this wave's half-chains */
    const unsigned int lane = threadIdx.x;
    const unsigned long long n_half = 2ull * C;

    for (unsigned int i = lane; i < D * m; i += 64)
        acc[i] = 0.f;
    for (unsigned int i = lane; i < D * m_pad; i += 64)
        y[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (unsigned long long hc = blockIdx.x; hc < n_half; hc += gridDim.x) {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        const T *src = sample + (chain * n + row0) * D;
        /* coalesced read of the contiguous [m, D] block, transposed into LDS */
        for (unsigned int e = lane; e < m * D; e += 64) {
            const unsigned int t = e / D, d = e - t * D;
            y[d * m_pad + t] = (float)src[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (unsigned int d = 0; d < D; ++d) {
            float *yd = y + d * m_pad;
            float s = 0.f;
            for (unsigned int t = lane; t < m; t += 64)
                s += yd[t];
            const float mean = wave_sum(s) / (float)m;
            float q = 0.f;
            for (unsigned int t = lane; t < m; t += 64) {
                const float v = yd[t] - mean;
                yd[t] = v;
                q += v * v;
            }
            q = wave_sum(q);
            if (lane == 0) {
                means[hc * D + d] = mean;
                ssq[hc * D + d] = q;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            /* lag sums: lane owns lags lane, lane+64, ... ; y[t] is a broadcast read, y[t+lag] a unit-stride
             * read, rows are zero padded so no bounds test is needed inside the loop */
            float *accd = acc + d * m;
            for (unsigned int l0 = 0; l0 < m; l0 += 64) {
                const unsigned int lag = l0 + lane;
                float a0 = 0.f, a1 = 0.f;
                const unsigned int tmax = m - l0; /* wave-uniform */
                unsigned int t = 0;
                for (; t + 1 < tmax; t += 2) {
                    a0 = fmaf(yd[t], yd[t + lag], a0);
                    a1 = fmaf(yd[t + 1], yd[t + 1 + lag], a1);
                }
                if (t < tmax)
                    a0 = fmaf(yd[t], yd[t + lag], a0);
                if (lag < m)
                    accd[lag] += a0 + a1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float *slab = slabs + (size_t)blockIdx.x * D * m;
    for (unsigned int i = lane; i < D * m; i += 64)
        slab[i] = acc[i];
}

/* Half-chains of ANY length (round 4; the last resort of stats_partials_impl: D m beyond what the kernels above can stage in
 * LDS and m beyond the power-spectrum kernels' 16 384 -- a few very long chains, e.g. one chain of 10^5 draws).  Nothing of a
 * half-chain is resident: the moments kernel reads it twice from global memory (one wave per half-chain and parameter), the
 * lag kernel gives a workgroup 256 consecutive lags of one parameter and walks its share of the half-chains in windows of
 * 1024 draws staged in LDS (A = y[t], B = y[t + lag0 + ...], zero past the end, so the inner loop has no bounds test);
 * window sums in f32, their total in f64.  O(C D m^2) like stats.rs:632-654's brute-force branch, but on 256 CUs. */
template <class T>
__global__ __launch_bounds__(64) void mm_half_chain_moments_any_kernel(const T *__restrict__ sample, unsigned long long C,
                                                                       unsigned int n, unsigned int D, unsigned int m,
                                                                       float *__restrict__ means, float *__restrict__ ssq)
{
    const unsigned int lane = threadIdx.x;
    const unsigned long long hc = blockIdx.x / D;
    const unsigned int d = blockIdx.x % D;
    const unsigned long long chain = hc < C ? hc : hc - C;
    const unsigned int row0 = hc < C ? 0u : n - m;
    const T *src = sample + (chain * n + row0) * D + d;
    float s = 0.f;
    for (unsigned int t = lane; t < m; t += 64)
        s += (float)src[(size_t)t * D];
    const float mean = wave_sum(s) / (float)m;
    float q = 0.f;
    for (unsigned int t = lane; t < m; t += 64) {
        const float v = (float)src[(size_t)t * D] - mean;
        q = fmaf(v, v, q);
    }
    q = wave_sum(q);
    if (lane == 0) {
        means[hc * D + d] = mean;
        ssq[hc * D + d] = q;
    }
}

template <class T>
__global__ __launch_bounds__(256) void mm_lag_sums_any_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n,
                                                              unsigned int D, unsigned int m, const float *__restrict__ means,
                                                              unsigned int n_slabs, float *__restrict__ slabs)
{
    constexpr unsigned int W = 1024, L = 256;
    __shared__ float A[W], B[W + L];
    const unsigned int tid = threadIdx.x, lag0 = blockIdx.x * L, d = blockIdx.y, slab = blockIdx.z;
    const unsigned long long n_half = 2ull * C;
    double total = 0.0;
    for (unsigned long long hc = slab; hc < n_half; hc += n_slabs) {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        const T *src = sample + (chain * n + row0) * D + d;
        const float mean = means[hc * D + d];
        for (unsigned int t0 = 0; t0 + lag0 < m; t0 += W) { /* windows whose products can be non-zero for this tile */
            __syncthreads();
            for (unsigned int i = tid; i < W; i += L) {
                const unsigned int t = t0 + i;
                A[i] = t < m ? (float)src[(size_t)t * D] - mean : 0.f;
            }
            for (unsigned int j = tid; j < W + L; j += L) {
                const unsigned int t = t0 + lag0 + j;
                B[j] = t < m ? (float)src[(size_t)t * D] - mean : 0.f;
            }
            __syncthreads();
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
            for (unsigned int i = 0; i < W; i += 4) {
                a0 = fmaf(A[i], B[i + tid], a0);
                a1 = fmaf(A[i + 1], B[i + 1 + tid], a1);
                a2 = fmaf(A[i + 2], B[i + 2 + tid], a2);
                a3 = fmaf(A[i + 3], B[i + 3 + tid], a3);
            }
            total += (double)((a0 + a1) + (a2 + a3));
        }
    }
    const unsigned int lag = lag0 + tid;
    if (lag < m)
        slabs[((size_t)slab * D + d) * m + lag] = (float)total;
}

/* The same half-chain statistics with the lag sums on the matrix cores.  For one parameter of one half-chain the
 * biased autocovariance sums c_k = sum_t y[t] y[t+k] are a product of two Hankel/Toeplitz operands:
 *     A[i][t] = y[t + i]  (16 x K),   B[t][j] = y[t - 16 j - 256 tau]  (K x 16)   =>   (A B)[i][j] = c_(i + 16 j + 256 tau)
 * (y = 0 outside [0, m)), so one 16 x 16 accumulator tile of v_mfma_f32_16x16x4_f32 holds 256 consecutive lags and
 * K = m suffices: m / 4 MFMAs per tile instead of ~2 m LDS reads and m FMAs per lane -- the direct kernel above is
 * LDS-issue bound (two reads per FMA).  y sits in LDS with 240 + 256 (tiles - 1) zeros in front and 36 behind, and
 * with one pad word after every 16 (address p + p / 16): the B operand's lanes read addresses 16 apart, which would
 * otherwise land on two banks.  Means, centred sums of squares and the carried per-wave slab are as above. */
typedef float mm_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float *mm_lds_cf;

template <class T>
__global__ __launch_bounds__(64) void mm_half_chain_mfma_kernel(const T *__restrict__ sample, unsigned long long C,
                                                                unsigned int n, unsigned int D, unsigned int m,
                                                                unsigned int n_tiles, float *__restrict__ means,
                                                                float *__restrict__ ssq, float *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned int padl = 240u + 256u * (n_tiles - 1u);
    const unsigned int row_len = padl + m + 36u;            /* logical elements per parameter (36 zeros behind) */
    const unsigned int row_pitch = row_len + row_len / 16u + 1u; /* with the bank-skew words */
    float *y = lds;                                          /* [D][row_pitch] */
    float *acc = lds + (size_t)D * row_pitch;                /* [D][m] lag sums carried over this wave's half-chains */
    const unsigned int lane = threadIdx.x;
    const unsigned int li = lane & 15u, kq = lane >> 4;
    const unsigned long long n_half = 2ull * C;
    auto at = [](unsigned int p) -> unsigned int { return p + (p >> 4); };

    for (unsigned int i = lane; i < D * m; i += 64)
        acc[i] = 0.f;
    for (unsigned int i = lane; i < D * row_pitch; i += 64)
        y[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const unsigned int n_groups = (m + 15u) / 16u; /* groups of four k-steps (16 elements of y) */
    const unsigned int w0 = at(padl + lane);
    /* operand addresses of k-steps 0..3 in a row, tile 0 (tile tau: B moves back by 256 elements = 272 words) */
    unsigned int offa[4], offb[4];
#pragma unroll
    for (unsigned int u = 0; u < 4; ++u) {
        offa[u] = at(padl + kq + li + 4u * u);
        offb[u] = at(padl + kq - 16u * li + 4u * u);
    }
    /* The [m, D] block of a half-chain is m D / 64 elements per lane.  When that is at most NPRE they are requested for
     * the NEXT half-chain before the current one is reduced, and parked in registers: a wave that asked for its block only
     * when it needed it (ten dependent global loads) spent most of its time waiting, four waves per SIMD or not.
     * lds_off[i]: where element lane + 64 i goes (word offset into y), the same for every half-chain. */
    constexpr unsigned int NPRE = 16;
    const unsigned int mD = m * D;
    const bool prefetch = mD <= NPRE * 64u;
    unsigned int lds_off[NPRE];
    float pre[NPRE];
    {
        /* (t, d) of element e = lane + 64 i without a division per element: advance by 64 = q64 * D + r64 */
        const unsigned int q64 = 64u / D, r64 = 64u - q64 * D;
        unsigned int t = lane / D, d = lane - t * D;
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i) {
            lds_off[i] = d * row_pitch + at(padl + t);
            t += q64;
            d += r64;
            if (d >= D) {
                d -= D;
                t += 1;
            }
        }
    }
    auto block_of = [&](unsigned long long hc) -> const T * {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        return sample + (chain * n + row0) * D;
    };
    auto request = [&](unsigned long long hc) {
        const T *src = block_of(hc);
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i) {
            const unsigned int e = lane + 64u * i;
            pre[i] = e < mD ? (float)src[e] : 0.f;
        }
    };
    if (prefetch && blockIdx.x < n_half)
        request(blockIdx.x);
    for (unsigned long long hc = blockIdx.x; hc < n_half; hc += gridDim.x) {
        if (prefetch) {
#pragma unroll
            for (unsigned int i = 0; i < NPRE; ++i)
                if (lane + 64u * i < mD)
                    y[lds_off[i]] = pre[i];
            if (hc + gridDim.x < n_half)
                request(hc + gridDim.x);
        } else {
            const T *src = block_of(hc);
            const unsigned int q64 = 64u / D, r64 = 64u - q64 * D;
            unsigned int t = lane / D, d = lane - t * D;
            for (unsigned int e = lane; e < mD; e += 64) {
                y[d * row_pitch + at(padl + t)] = (float)src[e];
                t += q64;
                d += r64;
                if (d >= D) {
                    d -= D;
                    t += 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (unsigned int d = 0; d < D; ++d) {
            float *yd = y + d * row_pitch;
            /* element t = lane + 64 i sits at word at(padl + lane) + 68 i (64 elements = 68 skewed words) */
            float s = 0.f;
            for (unsigned int t = lane, w = w0; t < m; t += 64, w += 68)
                s += yd[w];
            const float mean = wave_sum_dpp(s) / (float)m;
            float q = 0.f;
            for (unsigned int t = lane, w = w0; t < m; t += 64, w += 68) {
                const float v = yd[w] - mean;
                yd[w] = v;
                q += v * v;
            }
            q = wave_sum_dpp(q);
            if (lane == 0) {
                means[hc * D + d] = mean;
                ssq[hc * D + d] = q;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float *accd = acc + d * m;
            for (unsigned int tau = 0; tau < n_tiles; ++tau) {
                mm_f4 c0 = {0.f, 0.f, 0.f, 0.f};
                /* lane (li, kq): A[li][4 s + kq] = y[4 s + kq + li];  B[4 s + kq][li] = y[4 s + kq - 16 li - 256 tau].
                 * Four k-steps advance the logical index by 16, i.e. the skewed address by 17: four running addresses per
                 * operand, no index arithmetic in the loop, one accumulator chain (dependent MFMAs issue back to back) */
                /* Four running LDS pointers per operand (k-steps 4 g + u, u = 0..3); a group of four k-steps advances the
                 * logical index by 16 = 17 words, so inside a block of four groups every read is pointer + constant
                 * (ds_read_b32 with an immediate offset) and the pointers move once per 16 MFMAs.  The row's tail of
                 * zeros lets the last group run past m. */
                mm_lds_cf pa[4], pb[4];
#pragma unroll
                for (unsigned int u = 0; u < 4; ++u) {
                    pa[u] = (mm_lds_cf)(yd + offa[u]);
                    pb[u] = (mm_lds_cf)(yd + offb[u] - 272u * tau);
                }
                unsigned int g = 0;
                for (; g + 4 <= n_groups; g += 4) {
                    float av[16], bv[16];
#pragma unroll
                    for (unsigned int q = 0; q < 4; ++q)
#pragma unroll
                        for (unsigned int u = 0; u < 4; ++u) {
                            av[4 * q + u] = pa[u][17 * q];
                            bv[4 * q + u] = pb[u][17 * q];
                        }
#pragma unroll
                    for (unsigned int u = 0; u < 4; ++u) {
                        pa[u] += 68;
                        pb[u] += 68;
                    }
#pragma unroll
                    for (unsigned int i = 0; i < 16; ++i)
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], c0, 0, 0, 0);
                }
                for (; g < n_groups; ++g) {
                    float av[4], bv[4];
#pragma unroll
                    for (unsigned int u = 0; u < 4; ++u) {
                        av[u] = pa[u][0];
                        bv[u] = pb[u][0];
                        pa[u] += 17;
                        pb[u] += 17;
                    }
#pragma unroll
                    for (unsigned int u = 0; u < 4; ++u)
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], c0, 0, 0, 0);
                }
                /* accumulator element r of lane (li, kq) = tile row 4 kq + r, column li = lag 4 kq + r + 16 li + 256 tau */
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned int lag = 4u * kq + (unsigned int)r + 16u * li + 256u * tau;
                    if (lag < m)
                        accd[lag] += c0[r];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float *slab = slabs + (size_t)blockIdx.x * D * m;
    for (unsigned int i = lane; i < D * m; i += 64)
        slab[i] = acc[i];
}

/* The same half-chain statistics with the lag sums as REGISTER TILES on the vector ALU -- the default where it fits.
 *
 * Why not the matrix cores: v_mfma_f32_16x16x4_f32 has exactly the f32 vector rate (64 FLOP / clock / SIMD), and the
 * Hankel x Toeplitz product above spends 256 x m multiply-adds on the m (m + 1) / 2 products a half-chain's lag sums
 * need (2.5x at m = 200); its floor for [65 536, 400, 3] is 0.26 ms, the kernel took 0.45 ms.  Only the SUM over the
 * wave's half-chains of every lag sum is wanted, so nothing has to be reduced across lanes per half-chain: the
 * (t, lag) triangle of a parameter is cut into tiles of 8 time steps x 16 lags, every lane owns TPL of the
 * D x tiles(m) tiles for the whole launch and keeps its tiles' 16 lag sums in registers across all the half-chains of
 * the wave.  Per tile and half-chain: y[T0 .. T0 + 8) and y[T0 + K0 .. T0 + K0 + 24) come from LDS as eight
 * 16-byte reads, then 128 FMAs run from registers -- 16 multiply-adds per LDS instruction where the direct kernel did
 * one.  (y is zero past m, so tiles on the diagonal need no masks.)  At the end the lanes' tiles are folded into the
 * per-wave slab in a fixed order (bitwise reproducible, no atomics).  Lanes used: D tiles(m) / (64 TPL) = 99 % at
 * [., 400, 3]; executed multiply-adds 1.09 x the useful ones. */
template <class T, int TPL, bool VEC>
__global__ __launch_bounds__(64) void mm_half_chain_tile_kernel(const T *__restrict__ sample, unsigned long long C,
                                                                unsigned int n, unsigned int D, unsigned int m,
                                                                float *__restrict__ means, float *__restrict__ ssq,
                                                                float *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr unsigned int RT = 8, SK = 16; /* tile: RT time steps x SK lags */
    const unsigned int nbk = (m + SK - 1) / SK;
    /* a row of y: blocks of 8 elements, 12 words apart (element t at word 12 (t / 8) + t % 8), zeros behind (reads reach
     * element T0 + K0 + RT + SK).  The lanes of a 16-byte read are in consecutive time blocks: 48 bytes apart they cover
     * the 64 banks once per 16 lanes, 32 bytes apart (no pad) every read was a 2-way bank conflict */
    auto at = [](unsigned int t) -> unsigned int { return 12u * (t >> 3) + (t & 7u); };
    const unsigned int pitch = 12u * ((m + RT + SK + 8 + 7) / 8) + 4u;
    float *y = lds;                              /* [D][pitch] */
    float *slab = lds + (size_t)D * pitch;       /* [D][m] */
    float *scratch = slab + (size_t)D * m;       /* [64][SK] one tile slot of every lane (final fold) */
    const unsigned int lane = threadIdx.x;
    const unsigned long long n_half = 2ull * C;
    /* tiles of one parameter, lag block kb = 0 .. nbk-1: time blocks tb with RT tb + SK kb < m, i.e. cnt(kb) of them;
     * enumerated kb-major; tile e of the launch = d * n_tiles + first(kb) + tb */
    auto cnt = [&](unsigned int kb) -> unsigned int { return (m - SK * kb + RT - 1) / RT; };
    unsigned int n_tiles = 0;
    for (unsigned int kb = 0; kb < nbk; ++kb)
        n_tiles += cnt(kb);
    /* this lane's tiles: LDS word offsets of y[T0] and y[T0 + K0] in the row of the tile's parameter */
    unsigned int off_t[TPL], off_w[TPL];
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl) {
        const unsigned int e = lane + 64u * (unsigned int)sl;
        /* no tile in this slot: both operands from the zero tail of row 0 (element m + 8 onwards) */
        off_t[sl] = 12u * ((m + 15) / 8);
        off_w[sl] = 12u * ((m + 15) / 8);
        if (e < D * n_tiles) {
            const unsigned int d = e / n_tiles;
            unsigned int r = e - d * n_tiles, kb = 0;
            while (r >= cnt(kb)) {
                r -= cnt(kb);
                ++kb;
            }
            off_t[sl] = d * pitch + 12u * r;
            off_w[sl] = d * pitch + 12u * (r + 2u * kb);
        }
    }
    float acc[TPL][SK];
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl)
#pragma unroll
        for (unsigned int j = 0; j < SK; ++j)
            acc[sl][j] = 0.f;
    for (unsigned int i = lane; i < D * pitch; i += 64)
        y[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* the [m, D] block of the NEXT half-chain is requested before the current one is reduced (as in the MFMA kernel).
     * VEC (f32, every block 16-byte aligned: the host checks n D, (n - m) D and m D for multiples of 4): 16-byte loads,
     * three per lane at [., 400, 3] where 4-byte loads take ten -- the block arrives in a quarter of the instructions */
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr unsigned int NPRE = VEC ? 4 : 16;
    const unsigned int mD = m * D;
    const bool prefetch = mD <= NPRE * 64u * (VEC ? 4u : 1u);
    unsigned int lds_off[VEC ? 4 * NPRE : NPRE];
    float pre[VEC ? 1 : NPRE];
    f4 pre4[VEC ? NPRE : 1];
    if (VEC) {
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i)
#pragma unroll
            for (unsigned int c = 0; c < 4; ++c) {
                const unsigned int e = 4u * (lane + 64u * i) + c, t = e / D, d = e - t * D;
                lds_off[4 * i + c] = e < mD ? d * pitch + at(t) : 0u;
            }
    } else {
        const unsigned int q64 = 64u / D, r64 = 64u - q64 * D;
        unsigned int t = lane / D, d = lane - t * D;
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i) {
            lds_off[i] = d * pitch + at(t);
            t += q64;
            d += r64;
            if (d >= D) {
                d -= D;
                t += 1;
            }
        }
    }
    auto block_of = [&](unsigned long long hc) -> const T * {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        return sample + (chain * n + row0) * D;
    };
    auto request = [&](unsigned long long hc) {
        const T *src = block_of(hc);
        if constexpr (VEC) {
            const f4 *src4 = reinterpret_cast<const f4 *>(src);
#pragma unroll
            for (unsigned int i = 0; i < NPRE; ++i) {
                const unsigned int v = lane + 64u * i;
                pre4[i] = 4u * v < mD ? src4[v] : f4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (unsigned int i = 0; i < NPRE; ++i) {
                const unsigned int e = lane + 64u * i;
                pre[i] = e < mD ? (float)src[e] : 0.f;
            }
        }
    };
#ifdef MM_STATS_DEPHASE /* experiment hook of tools/stats_probe.hip: which waves start late, and by how many 64-cycle units */
    {
        const unsigned int sel = MM_STATS_DEPHASE == 1 ? (blockIdx.x & 1u)
                               : MM_STATS_DEPHASE == 2 ? (blockIdx.x >= gridDim.x / 2 ? 1u : 0u)
                               : MM_STATS_DEPHASE == 3 ? ((blockIdx.x >> 8) & 1u)
                               : ((blockIdx.x >> 2) & 1u);
        if (sel)
            __builtin_amdgcn_s_sleep(MM_STATS_DEPHASE_UNITS);
    }
#endif
    if (prefetch && blockIdx.x < n_half)
        request(blockIdx.x);
    for (unsigned long long hc = blockIdx.x; hc < n_half; hc += gridDim.x) {
        if (prefetch) {
            if constexpr (VEC) {
#pragma unroll
                for (unsigned int i = 0; i < NPRE; ++i)
                    if (4u * (lane + 64u * i) < mD) { /* m D is a multiple of 4: whole vectors */
#pragma unroll
                        for (unsigned int c = 0; c < 4; ++c)
                            y[lds_off[4 * i + c]] = pre4[i][c];
                    }
            } else {
#pragma unroll
                for (unsigned int i = 0; i < NPRE; ++i)
                    if (lane + 64u * i < mD)
                        y[lds_off[i]] = pre[i];
            }
            if (hc + gridDim.x < n_half)
                request(hc + gridDim.x);
        } else {
            const T *src = block_of(hc);
            const unsigned int q64 = 64u / D, r64 = 64u - q64 * D;
            unsigned int t = lane / D, d = lane - t * D;
            for (unsigned int e = lane; e < mD; e += 64) {
                y[d * pitch + at(t)] = (float)src[e];
                t += q64;
                d += r64;
                if (d >= D) {
                    d -= D;
                    t += 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* mean, centred sum of squares, centring in place, up to four parameters at a time in one basic block: their LDS
         * reads and wave reductions are independent chains the scheduler interleaves (one parameter after the other, two
         * waves per SIMD, every DPP / readlane / LDS latency was exposed: 0.10 of the kernel's 0.39 ms) */
#ifndef MM_STATS_PROBE_SKIP_CENTER
        auto centre = [&](unsigned int d0, auto nu_tag) __attribute__((always_inline)) {
            constexpr unsigned int NU = decltype(nu_tag)::value;
            float s0[NU], mean[NU], q[NU];
#pragma unroll
            for (unsigned int u = 0; u < NU; ++u)
                s0[u] = 0.f;
            for (unsigned int t = lane; t < m; t += 64) {
                const unsigned int w = at(t);
#pragma unroll
                for (unsigned int u = 0; u < NU; ++u)
                    s0[u] += y[(d0 + u) * pitch + w];
            }
#pragma unroll
            for (unsigned int u = 0; u < NU; ++u) {
                mean[u] = wave_sum_dpp(s0[u]) / (float)m;
                q[u] = 0.f;
            }
            for (unsigned int t = lane; t < m; t += 64) {
                const unsigned int w = at(t);
#pragma unroll
                for (unsigned int u = 0; u < NU; ++u) {
                    const float v = y[(d0 + u) * pitch + w] - mean[u];
                    y[(d0 + u) * pitch + w] = v;
                    q[u] += v * v;
                }
            }
#pragma unroll
            for (unsigned int u = 0; u < NU; ++u)
                q[u] = wave_sum_dpp(q[u]);
            if (lane == 0) {
#pragma unroll
                for (unsigned int u = 0; u < NU; ++u) {
                    means[hc * D + d0 + u] = mean[u];
                    ssq[hc * D + d0 + u] = q[u];
                }
            }
        };
        for (unsigned int d0 = 0; d0 < D; d0 += 4) {
            const unsigned int left = D - d0;
            if (left >= 4)
                centre(d0, std::integral_constant<unsigned int, 4>());
            else if (left == 3)
                centre(d0, std::integral_constant<unsigned int, 3>());
            else if (left == 2)
                centre(d0, std::integral_constant<unsigned int, 2>());
            else
                centre(d0, std::integral_constant<unsigned int, 1>());
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* the tiles, software-pipelined: the operands of tile sl + 1 are requested before the 128 multiply-adds of tile
         * sl (no branches in here: a lane without a tile in some slot reads zeros) */
        float yt[2][RT], yw[2][RT + SK];
        auto fetch = [&](int sl, int b) __attribute__((always_inline)) {
            const f4 *pt = reinterpret_cast<const f4 *>(y + off_t[sl]);
            const f4 *pw = reinterpret_cast<const f4 *>(y + off_w[sl]);
#pragma unroll
            for (unsigned int i = 0; i < RT / 4; ++i) {
                const f4 v = pt[3 * (i >> 1) + (i & 1u)];
                yt[b][4 * i] = v[0];
                yt[b][4 * i + 1] = v[1];
                yt[b][4 * i + 2] = v[2];
                yt[b][4 * i + 3] = v[3];
            }
#pragma unroll
            for (unsigned int i = 0; i < (RT + SK) / 4; ++i) {
                const f4 v = pw[3 * (i >> 1) + (i & 1u)]; /* block i / 2 at word 12 (i / 2), half i % 2 */
                yw[b][4 * i] = v[0];
                yw[b][4 * i + 1] = v[1];
                yw[b][4 * i + 2] = v[2];
                yw[b][4 * i + 3] = v[3];
            }
        };
#ifndef MM_STATS_PROBE_SKIP_TILES /* measurement aid of tools/stats_probe.hip; never defined in the library build */
        fetch(0, 0);
#pragma unroll
        for (int sl = 0; sl < TPL; ++sl) {
            const int b = sl & 1;
            if (sl + 1 < TPL)
                fetch(sl + 1, b ^ 1);
#pragma unroll
            for (unsigned int r = 0; r < RT; ++r)
#pragma unroll
                for (unsigned int j = 0; j < SK; ++j)
                    acc[sl][j] = fmaf(yt[b][r], yw[b][r + j], acc[sl][j]);
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    /* fold the lanes' tiles into the slab, slot by slot, in a fixed order: output (d, lag) takes, from the lanes whose
     * tile of this slot covers it (same d, kb = lag / SK: a run of consecutive tiles), their sums in ascending tb */
    for (unsigned int i = lane; i < D * m; i += 64)
        slab[i] = 0.f;
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (unsigned int j = 0; j < SK; ++j)
            scratch[lane * SK + j] = acc[sl][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned int e_lo = 64u * (unsigned int)sl, e_hi = e_lo + 64u;
        for (unsigned int o = lane; o < D * m; o += 64) {
            const unsigned int d = o / m, lag = o - d * m, kb = lag / SK, j = lag - kb * SK;
            unsigned int first = d * n_tiles;
            for (unsigned int q = 0; q < kb; ++q)
                first += cnt(q);
            const unsigned int last = first + cnt(kb); /* tiles [first, last) cover (d, kb) */
            const unsigned int lo = first > e_lo ? first : e_lo, hi = last < e_hi ? last : e_hi;
            float sum = slab[o];
            for (unsigned int e = lo; e < hi; ++e)
                sum += scratch[(e - e_lo) * SK + j];
            slab[o] = sum;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float *out = slabs + (size_t)blockIdx.x * D * m;
    for (unsigned int i = lane; i < D * m; i += 64)
        out[i] = slab[i];
}

/* The register-tile kernel with ONE PARAMETER PER WAVE: wave (s, d) = block s D + d reduces parameter d of the half-chains
 * s, s + S, s + 2 S, ... (S = gridDim.x / D) into row d of slab s.
 *
 * Why: the kernel above keeps all D x tiles(m) tiles of a half-chain in one wave -- 128 lag sums per lane at [., 400, 3],
 * 250 registers, TWO waves per SIMD -- and two waves do not fill a SIMD: a lone wave issues one vector instruction per
 * 4.4-5.5 cycles, so its 1024 FMAs alone take 4500+ cycles per half-chain and the SIMD's vector pipe sits at
 * 57 % (SQ counters; de-phasing the two waves changed nothing).  One parameter's tiles(m) tiles are a third of that: TPL
 * = 3 slots of 16 lag sums, the operands fetched per tile (no double buffer: other waves cover the LDS latency), ~110
 * registers = FOUR waves per SIMD, and the half-chain's [m, D] block is read by D waves (the lines stay in L2; HBM traffic
 * is unchanged).  Each (slab, parameter) accumulates the same half-chains in the same order as above; within a lag the
 * time blocks' partial sums are folded in ascending order as above. */
template <class T, int TPL, int NPRE_>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void mm_half_chain_tile1_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n, unsigned int D, unsigned int m,
                                unsigned int P, float *__restrict__ means, float *__restrict__ ssq, float *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr unsigned int RT = 8, SK = 16;
    const unsigned int nbk = (m + SK - 1) / SK;
    auto at = [](unsigned int t) -> unsigned int { return 12u * (t >> 3) + (t & 7u); };
    const unsigned int pitch = 12u * ((m + RT + SK + 8 + 7) / 8) + 4u;
    float *y = lds;              /* [pitch] */
    float *slab = lds + pitch;   /* [m] */
    float *scratch = slab + m;   /* [64][SK] */
    const unsigned int lane = threadIdx.x;
    /* block = (slab, part, parameter): a parameter's tiles are dealt to P waves, 64 TPL consecutive tiles each (P = 1 up to
     * 192 tiles, i.e. m <= 200; [., 1000, 3]: 1008 tiles, six parts); every part loads and centres the column itself */
    const unsigned int dsel = blockIdx.x % D, part = (blockIdx.x / D) % P, bid = blockIdx.x / (D * P), nblk = gridDim.x / (D * P);
    const unsigned int e0 = part * 64u * (unsigned int)TPL;
    const unsigned long long n_half = 2ull * C;
    auto cnt = [&](unsigned int kb) -> unsigned int { return (m - SK * kb + RT - 1) / RT; };
    unsigned int n_tiles = 0;
    for (unsigned int kb = 0; kb < nbk; ++kb)
        n_tiles += cnt(kb);
    unsigned int off_t[TPL], off_w[TPL];
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl) {
        const unsigned int e = e0 + lane + 64u * (unsigned int)sl;
        off_t[sl] = 12u * ((m + 15) / 8); /* no tile in this slot: both operands from the zero tail of the row */
        off_w[sl] = 12u * ((m + 15) / 8);
        if (e < n_tiles) {
            unsigned int r = e, kb = 0;
            while (r >= cnt(kb)) {
                r -= cnt(kb);
                ++kb;
            }
            off_t[sl] = 12u * r;
            off_w[sl] = 12u * (r + 2u * kb);
        }
    }
    float acc[TPL][SK];
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl)
#pragma unroll
        for (unsigned int j = 0; j < SK; ++j)
            acc[sl][j] = 0.f;
    for (unsigned int i = lane; i < pitch; i += 64)
        y[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* the column of the NEXT half-chain is requested before the current one is reduced: element t of the column is
     * src[t D] (a wave's 64 loads span 64 D elements: the D waves of a slab touch the same lines) */
    constexpr unsigned int NPRE = NPRE_; /* m <= 64 NPRE (the host checks) */
    /* No branches in the loop: a lane's element t = lane + 64 i past the end of the column is loaded from the column's last
     * element and masked to zero, and its centred value goes to a spare word behind the scratch area (the row's zero
     * tail must stay zero).  src_idx / lds_off / live are loop invariants. */
    unsigned int lds_off[NPRE], src_idx[NPRE];
    bool live[NPRE];
    float pre[NPRE];
    const float inv_m = 1.0f / (float)m;
#pragma unroll
    for (unsigned int i = 0; i < NPRE; ++i) {
        const unsigned int t = lane + 64u * i;
        live[i] = t < m;
        lds_off[i] = live[i] ? at(t) : pitch + m + 64u * SK + lane; /* the spare words: [64] behind scratch */
        src_idx[i] = (live[i] ? t : m - 1u) * D;
    }
    auto column_of = [&](unsigned long long hc) -> const T * {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        return sample + (chain * n + row0) * D + dsel;
    };
    auto request = [&](unsigned long long hc) {
        const T *src = column_of(hc);
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i)
            pre[i] = (float)src[src_idx[i]];
    };
    if (bid < n_half)
        request(bid);
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (unsigned long long hc = bid; hc < n_half; hc += nblk) {
        /* mean, centred sum of squares, centring: on the registers that hold the column (the row in LDS is written once,
         * centred); the mean is sum x (1 / m) with the reciprocal formed once per wave */
        float s0 = 0.f;
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i) {
            pre[i] = live[i] ? pre[i] : 0.f;
            s0 += pre[i];
        }
        const float mean = wave_sum_dpp_bcast(s0) * inv_m;
        float q = 0.f;
#pragma unroll
        for (unsigned int i = 0; i < NPRE; ++i) {
            const float v = live[i] ? pre[i] - mean : 0.f;
            y[lds_off[i]] = v;
            q = fmaf(v, v, q);
        }
        q = wave_sum_dpp_bcast(q);
        if (lane == 0 && part == 0) {
            means[hc * D + dsel] = mean;
            ssq[hc * D + dsel] = q;
        }
        if (hc + nblk < n_half)
            request(hc + nblk);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sl = 0; sl < TPL; ++sl) {
            float yt[RT], yw[RT + SK];
            const f4 *pt = reinterpret_cast<const f4 *>(y + off_t[sl]);
            const f4 *pw = reinterpret_cast<const f4 *>(y + off_w[sl]);
#pragma unroll
            for (unsigned int i = 0; i < RT / 4; ++i) {
                const f4 v = pt[3 * (i >> 1) + (i & 1u)];
                yt[4 * i] = v[0];
                yt[4 * i + 1] = v[1];
                yt[4 * i + 2] = v[2];
                yt[4 * i + 3] = v[3];
            }
#pragma unroll
            for (unsigned int i = 0; i < (RT + SK) / 4; ++i) {
                const f4 v = pw[3 * (i >> 1) + (i & 1u)];
                yw[4 * i] = v[0];
                yw[4 * i + 1] = v[1];
                yw[4 * i + 2] = v[2];
                yw[4 * i + 3] = v[3];
            }
#pragma unroll
            for (unsigned int r = 0; r < RT; ++r)
#pragma unroll
                for (unsigned int j = 0; j < SK; ++j)
                    acc[sl][j] = fmaf(yt[r], yw[r + j], acc[sl][j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (unsigned int i = lane; i < m; i += 64)
        slab[i] = 0.f;
#pragma unroll
    for (int sl = 0; sl < TPL; ++sl) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (unsigned int j = 0; j < SK; ++j)
            scratch[lane * SK + j] = acc[sl][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned int e_lo = e0 + 64u * (unsigned int)sl, e_hi = e_lo + 64u;
        for (unsigned int lag = lane; lag < m; lag += 64) {
            const unsigned int kb = lag / SK, j = lag - kb * SK;
            unsigned int first = 0;
            for (unsigned int qq = 0; qq < kb; ++qq)
                first += cnt(qq);
            const unsigned int last = first + cnt(kb);
            const unsigned int lo = first > e_lo ? first : e_lo, hi = last < e_hi ? last : e_hi;
            float sum = slab[lag];
            for (unsigned int e = lo; e < hi; ++e)
                sum += scratch[(e - e_lo) * SK + j];
            slab[lag] = sum;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float *out = slabs + (((size_t)bid * P + part) * D + dsel) * m;
    for (unsigned int i = lane; i < m; i += 64)
        out[i] = slab[i];
}

/* ---- lag sums through the power spectrum (mm_stats_fft.h; stats.rs:576-620) ---------------------------------------------
 * mm_chain_fft_kernel<T, R1, DT>: a workgroup of four independent waves; wave w of workgroup g takes the chains 4 g + w,
 * 4 g + w + W, ... (W waves per parameter tile) and DT consecutive parameters d0 .. d0 + DT - 1 of each (the last tile of
 * a dimension that is not a multiple of DT starts at D - DT: the overlap is computed twice, bit-identically).
 *   load      point t = 64 a + lane of BOTH half-chains, all DT parameters in one load per (half, a) (12 bytes per lane at
 *             D = DT = 3: a wave's 64 loads are 768 contiguous bytes); the NEXT chain's points are requested before
 *             the current one is reduced;
 *   centre    mean and centred sum of squares per (half-chain, parameter) by DPP wave sums (the statistics withinvar
 *             needs: means / ssq [2 C, D] in splitcat order, as every other kernel of this file writes them);
 *   FFT       z = a + i b, three passes (mm_stats_fft.h), S[dt] += |Z|^2 on the lane's R1 bins.
 * At the end the four waves fold their S into natural bin order in LDS one after the other (fixed order: bitwise
 * reproducible) and the workgroup writes one slab [DT][N]; slabs: [workgroups][D][N]. */
template <class T, int DT>
__device__ __forceinline__ void mm_load_dt(const T *p, float (&out)[DT])
{
    if constexpr (std::is_same<T, float>::value && DT > 1) {
        typedef float vt __attribute__((ext_vector_type(DT), aligned(4)));
        const vt v = *reinterpret_cast<const vt *>(p);
#pragma unroll
        for (int i = 0; i < DT; ++i)
            out[i] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < DT; ++i)
            out[i] = (float)p[i];
    }
}

#define MM_WAVE_LDS_SYNC()                                                                                          \
    do {                                                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                      \
        __builtin_amdgcn_wave_barrier();                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                      \
    } while (0)

template <class T, int R1, int DT, int WPE, bool TWL, bool TOP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void mm_chain_fft_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n, unsigned int D, unsigned int m,
                         unsigned int n_pt, const mm_cx *__restrict__ tw, float *__restrict__ means,
                         float *__restrict__ ssq, float *__restrict__ slabs)
{
    using pl = mm_fft_plan<R1>;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    mm_cx *lds = reinterpret_cast<mm_cx *>(lds_raw) + wave * pl::LDS_CX;
    const unsigned int pt = blockIdx.x % n_pt, wg = blockIdx.x / n_pt, n_wg = gridDim.x / n_pt;
    const unsigned int d0 = pt * DT + DT <= D ? pt * DT : D - DT;
    const unsigned long long Wt = 4ull * n_wg;

    /* twiddles: in registers (2 (R1 + 6) of them), or -- TWL -- one copy per workgroup in LDS behind the exchange blocks,
     * read where they are used (R1 + 6 more ds_read_b64 per transform, conflict-free: lane-contiguous) */
    mm_cx tw1[TWL ? 1 : R1], tw2[TWL ? 1 : 8];
    const mm_cx *twl = reinterpret_cast<const mm_cx *>(lds_raw) + 4 * pl::LDS_CX;
    if constexpr (TWL) {
        mm_cx *twl_w = reinterpret_cast<mm_cx *>(lds_raw) + 4 * pl::LDS_CX;
        for (unsigned int i = threadIdx.x; i < R1 * 64u + 64u; i += 256u)
            twl_w[i] = tw[i];
        __syncthreads();
    } else {
#pragma unroll
        for (int b = 1; b < R1; ++b)
            tw1[b] = tw[b * 64 + lane];
        tw1[0] = mm_cx{1.f, 0.f};
#pragma unroll
        for (int g = 1; g < 8; ++g)
            tw2[g] = tw[R1 * 64 + g * 8 + (lane & 7u)];
        tw2[0] = mm_cx{1.f, 0.f};
    }
    auto tw1_of = [&](int b) -> mm_cx {
        if constexpr (TWL)
            return twl[b * 64 + lane];
        else
            return tw1[b];
    };
    auto tw2_of = [&](int g) -> mm_cx {
        if constexpr (TWL)
            return twl[R1 * 64 + g * 8 + (lane & 7u)];
        else
            return tw2[g];
    };

    /* TOP: the host found 64 (H - 1) < m, so only the last of a lane's H points can lie in the padding -- the others need
     * no select (a select on a lane mask costs two issue slots: tools/xlane_rate.hip) */
    unsigned int voff[pl::H];
    bool valid[pl::H];
#pragma unroll
    for (int a = 0; a < pl::H; ++a) {
        const unsigned int t = 64u * a + lane;
        valid[a] = (TOP && a < pl::H - 1) ? true : t < m;
        voff[a] = (valid[a] ? t : m - 1u) * D;
    }
    const float inv_m = 1.0f / (float)m;
    const size_t second = (size_t)(n - m) * D; /* the second half-chain: rows [n - m, n) */
    float S[DT][pl::J][8];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < pl::J; ++j)
#pragma unroll
            for (int h = 0; h < 8; ++h)
                S[dt][j][h] = 0.f;

    float pre[2][pl::H][DT];
    auto request = [&](unsigned long long c) {
        const T *base = sample + (size_t)c * n * D + d0;
#pragma unroll
        for (int a = 0; a < pl::H; ++a) {
            mm_load_dt<T, DT>(base + voff[a], pre[0][a]);
            mm_load_dt<T, DT>(base + second + voff[a], pre[1][a]);
        }
    };
    unsigned long long c = 4ull * wg + wave;
    if (c < C)
        request(c);
    for (; c < C; c += Wt) {
        float y[2][pl::H][DT];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int a = 0; a < pl::H; ++a)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    y[hf][a][dt] = valid[a] ? pre[hf][a][dt] : 0.f;
        /* the next chain of this wave (the last trip asks for its own chain again: no branch) */
        request(c + Wt < C ? c + Wt : c);
        float mu[2][DT], qq[2][DT];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float s0 = y[hf][0][dt];
#pragma unroll
                for (int a = 1; a < pl::H; ++a)
                    s0 += y[hf][a][dt];
                mu[hf][dt] = wave_sum_dpp_bcast(s0) * inv_m;
            }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float q = 0.f;
#pragma unroll
                for (int a = 0; a < pl::H; ++a) {
                    const float v = valid[a] ? y[hf][a][dt] - mu[hf][dt] : 0.f;
                    y[hf][a][dt] = v;
                    q = fmaf(v, v, q);
                }
                qq[hf][dt] = wave_sum_dpp_bcast(q);
            }
        if (lane == 0) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const size_t o = ((size_t)c + (hf ? (size_t)C : 0)) * D + d0 + dt;
                    means[o] = mu[hf][dt];
                    ssq[o] = qq[hf][dt];
                }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            mm_cx z[pl::H];
#pragma unroll
            for (int a = 0; a < pl::H; ++a)
                z[a] = mm_cx{y[0][a][dt], y[1][a][dt]};
            mm_fft_pass1<R1>(z, tw1_of, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            mm_cx v[pl::J][8];
            mm_fft_pass2_load<R1>(v, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            mm_fft_pass2_store<R1>(v, tw2_of, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            mm_fft_pass3<R1>(lds, (int)lane, S[dt]);
            MM_WAVE_LDS_SYNC();
        }
    }
    /* fold the four waves' spectra into natural bin order, one wave after the other */
    float *acc = lds_raw;
    for (unsigned int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int j = 0; j < pl::J; ++j)
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        const unsigned int i = dt * pl::N + (unsigned int)mm_fft_bin<R1>((int)lane, j, h);
                        acc[i] = w == 0 ? S[dt][j][h] : acc[i] + S[dt][j][h];
                    }
        }
    }
    __syncthreads();
    float *out = slabs + ((size_t)wg * D + d0) * pl::N;
    for (unsigned int i = threadIdx.x; i < DT * pl::N; i += 256)
        out[i] = acc[i];
}

/* ---- half-chains longer than 1024 draws (n_collect > 2048): the same power spectrum, the transform cut in two --------------
 * N = N1 x 2048 >= 2 m.  With t = n2 + 2048 n1 and f = k1 + N1 k2 (Cooley-Tukey),
 *     Z[k1 + N1 k2] = sum_{n2 < 2048} w_2048^(n2 k2) . [ w_N^(n2 k1) . sum_{n1 < N1 / 2} z[n2 + 2048 n1] w_N1^(n1 k1) ]
 * (the upper half of n1 is the zero padding): for every residue k1 a short sum over n1 per point, a twiddle, and the
 * wave-level 2048-point transform of mm_stats_fft.h (all of whose inputs are data now: mm_fft_pass1_full).  One wave per
 * workgroup takes the chains wg, wg + n_wg, ... of ONE parameter; its power spectrum S[N] lives in LDS (bins in natural
 * order) and leaves as one slab [N] at the end: the tail kernel and the inverse are shared with the short-chain path.
 * The chain is read N1 + 1 times (once for the means), from L2 after the first: this path is for samples whose direct
 * lag sums would cost m^2 (a [65536, 4000, 3] sample: 406 ms through the direct kernel, and no kernel at all beyond
 * m = 6800 at D = 3, where its LDS layout ends).  N <= 32768: m <= 16384. */
#ifndef MM_STATS_LONG_PREFETCH
#define MM_STATS_LONG_PREFETCH 1
#endif
template <class T, int N1, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mm_chain_fft_long_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n,
                                                               unsigned int D, unsigned int m, unsigned int n_wg,
                                                               const mm_cx *__restrict__ tw, const mm_cx *__restrict__ wN,
                                                               float *__restrict__ means, float *__restrict__ ssq,
                                                               float *__restrict__ slabs)
{
    constexpr int R1 = 32, HALF = N1 / 2;
    constexpr unsigned int N = 2048u * N1;
    /* loads are issued in batches of AC x HALF x 2 before any is used: one at a time (a dependent load per point) the kernel
     * waited out a memory latency per point -- 11.9 ms for [65536, 4000, 3] */
    constexpr int AC = HALF >= 8 ? 4 : 8;
    using pl = mm_fft_plan<R1>;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    /* WAVES > 1 (round 5, N1 = 4 and 8): the residues k1 of a chain are dealt to the waves of the workgroup (k1 = wave, wave +
     * WAVES, ...), which share ONE spectrum S[N] -- the bins k1 + N1 k2 of different residues are different words, so no two waves
     * ever meet and S needs no synchronisation between the first and the last barrier of the launch -- and each have their own
     * exchange block.  With one wave per workgroup the 64 KB spectrum of N1 = 8 left one wave per CU (S is what fills the LDS);
     * four waves beside one S are four per CU, and each reads the chain N1 / WAVES + 1 / WAVES times instead of N1 + 1. */
    static_assert(WAVES == 1 || N1 % WAVES == 0, "residues are dealt evenly");
    static_assert(N1 != 4 || WAVES > 1, "the staged column becomes visible at the barrier of the means");
    __shared__ float msum[2][WAVES][2]; /* the waves' shares of the two half-chains' sums, by chain parity */
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *const S = lds_raw;                                        /* [N] */
    mm_cx *const lds = reinterpret_cast<mm_cx *>(lds_raw + N) + (size_t)wave * pl::LDS_CX; /* the wave's exchange block */
    /* Workgroups go to the eight XCDs in turn (blockIdx.x mod 8) and every XCD has its own L2: with d = blockIdx.x mod D the D
     * workgroups that read the SAME chains (one parameter each, interleaved in the same cache lines) sat on D different XCDs and
     * every line came from HBM D times.  Here the D workgroups of a chain group are neighbours on ONE XCD: one fetch, D - 1 hits
     * in that L2 (round 5: [65536, 4000, 3] 3.3 -> see DESIGN 5.2).  The launcher gives a multiple of 8 chain groups whenever
     * there are that many chains. */
    unsigned int d, wg;
    if (n_wg % 8u == 0u) {
        const unsigned int xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        d = i % D;
        wg = (i / D) * 8u + xcd;
    } else {
        d = blockIdx.x % D;
        wg = blockIdx.x / D;
    }
    for (unsigned int i = threadIdx.x; i < N; i += 64u * WAVES)
        S[i] = 0.f;
    if constexpr (WAVES > 1)
        __syncthreads();
    else
        MM_WAVE_LDS_SYNC();
    auto tw1_of = [&](int b) -> mm_cx { return tw[b * 64 + lane]; };
    auto tw2_of = [&](int g) -> mm_cx { return tw[R1 * 64 + g * 8 + (lane & 7u)]; };
    const size_t second = (size_t)(n - m) * D; /* the second half-chain: rows [n - m, n) */
    const float inv_m = 1.0f / (float)m;
    /* N1 = 2 (half-chains up to 2048 draws): a lane's R1 points of both half-chains stay in its registers, so the chain is
     * read ONCE -- the means and both residue passes take their points from there -- instead of three times (round 5:
     * [65536, 4000, 3] 4.0 -> 3.3 ms).  At N1 = 4 the 128 cached values per lane made the kernel SLOWER ([65536, 8000, 3]
     * 13.4 -> 19.4 ms): kept to N1 = 2 */
    constexpr bool CACHE = HALF == 1;
    /* N1 = 4 (half-chains of 2049 .. 4096 draws): the parameter's column of both half-chains is STAGED in LDS once per chain
     * (32 KB beside the 32 KB spectrum), by the pass that sums it for the means, and the four residue passes read it from there:
     * from the L2 every pass pulled the chain's lines again -- all D parameters of them, 75 GB at [65536, 8000, 3] -- and the
     * kernel sat at the L2's bandwidth whatever its occupancy. */
    constexpr bool STAGE = N1 == 4;
    float *const st0 = lds_raw + N + (size_t)WAVES * 2 * pl::LDS_CX, *const st1 = st0 + N / 2;
    unsigned int parity = 0;
    /* N1 = 2: the NEXT chain's points are requested before this chain's two transforms and arrive behind them (one wave per
     * SIMD has nobody to hide a memory latency behind; the second set of 64 registers is there) */
    float nx0[CACHE ? R1 : 1][CACHE ? HALF : 1], nx1[CACHE ? R1 : 1][CACHE ? HALF : 1];
    auto fetch = [&](unsigned long long cc, float (&f0)[CACHE ? R1 : 1][CACHE ? HALF : 1], float (&f1)[CACHE ? R1 : 1][CACHE ? HALF : 1])
                     __attribute__((always_inline)) {
        const T *const b = sample + (size_t)cc * n * D + d;
#pragma unroll
        for (int a = 0; a < (CACHE ? R1 : 1); ++a)
#pragma unroll
            for (int n1 = 0; n1 < (CACHE ? HALF : 1); ++n1) {
                const unsigned int t = 64u * a + lane + 2048u * n1, tc = t < m ? t : m - 1u;
                f0[a][n1] = (float)b[(size_t)tc * D];
                f1[a][n1] = (float)b[second + (size_t)tc * D];
            }
    };
    if constexpr (CACHE && MM_STATS_LONG_PREFETCH)
        if (wg < C)
            fetch(wg, nx0, nx1);
    for (unsigned long long c = wg; c < C; c += n_wg, parity ^= 1u) {
        const T *const base = sample + (size_t)c * n * D + d;
        float s0 = 0.f, s1 = 0.f;
        float c0[CACHE ? R1 : 1][CACHE ? HALF : 1], c1[CACHE ? R1 : 1][CACHE ? HALF : 1];
        if constexpr (CACHE) {
            if constexpr (MM_STATS_LONG_PREFETCH) {
#pragma unroll
                for (int a = 0; a < R1; ++a)
#pragma unroll
                    for (int n1 = 0; n1 < HALF; ++n1) {
                        c0[a][n1] = nx0[a][n1];
                        c1[a][n1] = nx1[a][n1];
                    }
                if (c + n_wg < C)
                    fetch(c + n_wg, nx0, nx1);
            } else
#pragma unroll
            for (int a = 0; a < R1; ++a)
#pragma unroll
                for (int n1 = 0; n1 < HALF; ++n1) {
                    const unsigned int t = 64u * a + lane + 2048u * n1, tc = t < m ? t : m - 1u;
                    c0[a][n1] = (float)base[(size_t)tc * D];
                    c1[a][n1] = (float)base[second + (size_t)tc * D];
                }
            /* the same summation order as the strided pass below: t = lane + 64 i ascending per lane, i.e. n1-major */
#pragma unroll
            for (int n1 = 0; n1 < HALF; ++n1)
#pragma unroll
                for (int a = 0; a < R1; ++a) {
                    const bool in = 64u * a + lane + 2048u * n1 < m;
                    s0 += in ? c0[a][n1] : 0.f;
                    s1 += in ? c1[a][n1] : 0.f;
                }
        } else {
        if constexpr (STAGE)
            __syncthreads(); /* the last chain's residue passes have read the staged column */
        for (unsigned int t0 = lane + 64u * 8u * wave; t0 < m; t0 += 64u * 8u * WAVES) {
            float u0[8], u1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned int t = t0 + 64u * u, tc = t < m ? t : m - 1u;
                u0[u] = (float)base[(size_t)tc * D];
                u1[u] = (float)base[second + (size_t)tc * D];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = t0 + 64u * u < m;
                s0 += in ? u0[u] : 0.f;
                s1 += in ? u1[u] : 0.f;
                if constexpr (STAGE) {
                    if (in) {
                        st0[t0 + 64u * u] = u0[u];
                        st1[t0 + 64u * u] = u1[u];
                    }
                }
            }
        }
        }
        float t0s = wave_sum_dpp_bcast(s0), t1s = wave_sum_dpp_bcast(s1);
        if constexpr (WAVES > 1 && !CACHE) { /* (a wave that caches the chain has summed all of it itself) */
            /* every wave summed its share of the time steps: the shares meet in LDS and are added in wave order.  One barrier
             * per chain (the buffer alternates with the chain's parity: a wave two chains ahead cannot exist, it would have had
             * to pass the barrier in between) */
            if (lane == 0u) {
                msum[parity][wave][0] = t0s;
                msum[parity][wave][1] = t1s;
            }
            __syncthreads();
            t0s = msum[parity][0][0];
            t1s = msum[parity][0][1];
#pragma unroll
            for (int w2 = 1; w2 < WAVES; ++w2) {
                t0s += msum[parity][w2][0];
                t1s += msum[parity][w2][1];
            }
        }
        const float mu0 = t0s * inv_m, mu1 = t1s * inv_m;
        float q0 = 0.f, q1 = 0.f;
        for (unsigned int k1 = wave; k1 < (unsigned int)N1; k1 += (unsigned int)WAVES) {
            /* The offsets of a lane's R1 x HALF points (and their in-range masks, and the twiddle indices) depend on neither the
             * chain nor the residue, and the compiler, left to itself, computes them all ONCE before the chain loop and keeps
             * them: hundreds of 64-bit values it has no registers for (N1 = 8: 2.4 KB of scratch per lane, 245 spill stores
             * ahead of the loop and a scratch reload in front of every load of the sample -- a residue pass took eleven times
             * what it takes at N1 = 2; 4.9 KB at N1 = 16).  An opaque copy of the lane index per residue makes them cheap
             * arithmetic again, done where it is used. */
            unsigned int lane_k = lane;
            if constexpr (N1 == 4 || N1 == 8) /* N1 = 16 is faster with the spilled table than with the arithmetic (307 vs 381 ms) */
                asm volatile("" : "+v"(lane_k));
            const mm_cx *wNk = wN;
            mm_cx w1[HALF]; /* w_N1^(n1 k1) */
#pragma unroll
            for (int n1 = 0; n1 < HALF; ++n1)
                w1[n1] = wNk[(((unsigned int)n1 * k1) & (unsigned int)(N1 - 1)) * 2048u];
            mm_cx y[R1];
#pragma unroll
            for (int a0 = 0; a0 < R1; a0 += AC) {
                float r0[AC][HALF], r1[AC][HALF];
                mm_cx wt[AC];
#pragma unroll
                for (int aa = 0; aa < AC; ++aa) {
                    const unsigned int n2 = 64u * (a0 + aa) + lane_k;
                    wt[aa] = wNk[(n2 * k1) & (N - 1u)]; /* w_N^(n2 k1) */
#pragma unroll
                    for (int n1 = 0; n1 < HALF; ++n1) {
                        if constexpr (CACHE) {
                            r0[aa][n1] = c0[a0 + aa][n1];
                            r1[aa][n1] = c1[a0 + aa][n1];
                        } else if constexpr (STAGE) {
                            r0[aa][n1] = st0[n2 + 2048u * n1]; /* t < N / 2: inside the block; beyond m masked below */
                            r1[aa][n1] = st1[n2 + 2048u * n1];
                        } else {
                            const unsigned int t = n2 + 2048u * n1, tc = t < m ? t : m - 1u;
                            r0[aa][n1] = (float)base[(size_t)tc * D];
                            r1[aa][n1] = (float)base[second + (size_t)tc * D];
                        }
                    }
                }
#pragma unroll
                for (int aa = 0; aa < AC; ++aa) {
                    const unsigned int n2 = 64u * (a0 + aa) + lane_k;
                    mm_cx acc = mm_cx{0.f, 0.f};
#pragma unroll
                    for (int n1 = 0; n1 < HALF; ++n1) {
                        const bool in = n2 + 2048u * n1 < m;
                        const mm_cx v = mm_cx{in ? r0[aa][n1] - mu0 : 0.f, in ? r1[aa][n1] - mu1 : 0.f};
                        if (k1 == 0u) { /* every point passes here exactly once per k1 */
                            q0 = fmaf(v.re, v.re, q0);
                            q1 = fmaf(v.im, v.im, q1);
                        }
                        acc = mm_cx_add(acc, mm_cx_mul(v, w1[n1].re, w1[n1].im));
                    }
                    y[a0 + aa] = mm_cx_mul(acc, wt[aa].re, wt[aa].im);
                }
            }
            mm_fft_pass1_full<R1>(y, tw1_of, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            mm_cx v2[pl::J][8];
            mm_fft_pass2_load<R1>(v2, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            mm_fft_pass2_store<R1>(v2, tw2_of, lds, (int)lane);
            MM_WAVE_LDS_SYNC();
            float Sr[pl::J][8];
#pragma unroll
            for (int j = 0; j < pl::J; ++j)
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    Sr[j][h] = 0.f;
            mm_fft_pass3<R1>(lds, (int)lane, Sr);
            MM_WAVE_LDS_SYNC();
#pragma unroll
            for (int j = 0; j < pl::J; ++j)
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const unsigned int f = k1 + (unsigned int)N1 * (unsigned int)mm_fft_bin<R1>((int)lane, j, h);
                    S[f] += Sr[j][h]; /* every (lane, j, h) owns its bin: no two lanes meet */
                }
        }
        q0 = wave_sum_dpp_bcast(q0);
        q1 = wave_sum_dpp_bcast(q1);
        if (lane == 0u && wave == 0u) { /* residue 0, where the squares are taken, is wave 0's */
            means[(size_t)c * D + d] = mu0;
            means[((size_t)c + (size_t)C) * D + d] = mu1;
            ssq[(size_t)c * D + d] = q0;
            ssq[((size_t)c + (size_t)C) * D + d] = q1;
        }
    }
    if constexpr (WAVES > 1)
        __syncthreads();
    else
        MM_WAVE_LDS_SYNC();
    float *const out = slabs + ((size_t)wg * D + d) * N;
    for (unsigned int i = threadIdx.x; i < N; i += 64u * WAVES)
        out[i] = S[i];
}

/* ---- half-chains beyond 2048 draws, round 6: one (residue, parameter) per WAVE, the spectrum in REGISTERS -------------------
 * The cut is the one above (N = N1 x 2048, t = n2 + 2048 n1, f = k1 + N1 k2), but a wave keeps ONE residue k1 of ONE
 * parameter for all the chains it walks: the 2048 bins k1 + N1 k2 of its |Z|^2 are 32 registers per lane that pass 3
 * accumulates into directly, exactly as the short path does (mm_chain_fft_kernel) -- no spectrum in LDS (round 5's
 * mm_chain_fft_long_kernel kept S[N] there: one wave per SIMD at N1 = 4 and 8, one per CU at N1 = 16, a read-modify-write of
 * 2048 LDS words per residue and chain; a 2048-point residue transform took 38 000 cycles against ~6000).  N1 is a run-time value (any count >= 3 with 2048 N1 >= 2 m: half-chains up to 131 072 draws), the sum over
 * the N1 / 2 data blocks of a point a run-time loop.
 * The half-chain means come from a streaming pass of their own (mm_half_chain_means_kernel: the sample once, flat and
 * coalesced); the sums of squares are taken by the residue-0 waves, which see every point once.
 * Workgroup = one wave; blockIdx -> (XCD, chain group, (k1, d)) so that the N1 D waves reading the same chains sit on ONE
 * XCD (one fetch from HBM, N1 D - 1 hits in its L2) and run at the same time; slabs [n_grp][D][N] as everywhere. */
template <class T>
__global__ __launch_bounds__(1024) void mm_half_chain_means_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n,
                                                                   unsigned int D, unsigned int m, float *__restrict__ means)
{
    /* block = 64 D threads over ONE half-chain taken as a flat array of m D numbers: thread i sees parameter i mod D only
     * (the stride 64 D is a multiple of D), reads are unit-stride over the block; its sum meets the 63 others of its
     * parameter in LDS, added in a fixed order */
    __shared__ float part[1024];
    const unsigned long long hc = blockIdx.x;
    const unsigned long long chain = hc < C ? hc : hc - C;
    const unsigned int row0 = hc < C ? 0u : n - m;
    const T *src = sample + (chain * n + row0) * D;
    const unsigned int W = 64u * D, tid = threadIdx.x, total = m * D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    unsigned int e = tid;
    for (; e + 3u * W < total; e += 4u * W) {
        const float a = (float)src[e], b = (float)src[e + W], c = (float)src[e + 2u * W], d = (float)src[e + 3u * W];
        s0 += a;
        s1 += b;
        s2 += c;
        s3 += d;
    }
    for (; e < total; e += W)
        s0 += (float)src[e];
    part[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < D) {
        float s = 0.f;
        for (unsigned int k = 0; k < 64u; ++k)
            s += part[tid + k * D];
        means[hc * D + tid] = s / (float)m;
    }
}

template <class T, int R1, int HP, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void mm_chain_fft_res_kernel(const T *__restrict__ sample, unsigned long long C, unsigned int n, unsigned int D, unsigned int m,
                             unsigned int N1, unsigned int n_grp, const mm_cx *__restrict__ tw, const mm_cx *__restrict__ wN,
                             const float *__restrict__ means, float *__restrict__ ssq, float *__restrict__ slabs)
{
    /* ONE wave per SIMD with the 512-register budget, and nothing but the sample's loads in the vector-memory queue: the
     * transform's twiddles and the residue's w_N^(n2 k1) sit in registers for the life of the wave, the wave-uniform
     * w_N1^(n1 k1) come through the scalar cache.  A lane's 32 points of HP data blocks of both half-chains (64 HP
     * registers) are REQUESTED for the next step -- the next group of data blocks of this chain, or the first group of the
     * next chain -- as soon as the pre-sum has consumed the current ones, so the long step, the 2048-point transform, runs
     * with the next chain's loads in flight (first version of this kernel, two waves per SIMD, loads in batches of 16 with
     * their use right behind: [65536, 8000, 3] 12.9 ms = eight exposed memory latencies per chain and residue). */
    /* R1 = 32: data blocks and inner transforms of 2048 points, one wave per SIMD (the form described above); R1 = 16: of 1024
     * points -- twice the residues, half the registers: two waves per SIMD (N1 counts blocks of 64 R1 points) */
    constexpr unsigned int BLK = 64u * R1;
    using pl = mm_fft_plan<R1>;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    mm_cx *const lds = reinterpret_cast<mm_cx *>(lds_raw);
    const unsigned int lane = threadIdx.x;
    /* N1 is ANY count of blocks with N = BLK N1 >= 2 m (the outer step is a plain sum over the data blocks: nothing in it wants
     * a power of two); only the NB = ceil(m / BLK) blocks that hold draws are walked, in G steps of HP */
    const unsigned int N = BLK * N1, NB = (m + BLK - 1u) / BLK, per = D * N1, G = (NB + (unsigned int)HP - 1u) / (unsigned int)HP;
    unsigned int kd, grp;
    if (n_grp % 8u == 0u) {
        const unsigned int xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        kd = i % per;
        grp = (i / per) * 8u + xcd;
    } else {
        kd = blockIdx.x % per;
        grp = blockIdx.x / per;
    }
    /* d fastest: the D waves that read the very same cache lines (interleaved parameters) are neighbours */
    const unsigned int d = kd % D, k1 = __builtin_amdgcn_readfirstlane(kd / D);
    mm_cx tw1[R1], tw2[8], wt[R1];
#pragma unroll
    for (int b = 1; b < R1; ++b)
        tw1[b] = tw[b * 64 + lane];
    tw1[0] = mm_cx{1.f, 0.f};
#pragma unroll
    for (int g = 1; g < 8; ++g)
        tw2[g] = tw[R1 * 64 + g * 8 + (lane & 7u)];
    tw2[0] = mm_cx{1.f, 0.f};
#pragma unroll
    for (int a = 0; a < R1; ++a)
        wt[a] = wN[((64u * (unsigned int)a + lane) * k1) % N]; /* w_N^(n2 k1), n2 = 64 a + lane */
    auto tw1_of = [&](int b) -> mm_cx { return tw1[b]; };
    auto tw2_of = [&](int g) -> mm_cx { return tw2[g]; };
    float S[pl::J][8];
#pragma unroll
    for (int j = 0; j < pl::J; ++j)
#pragma unroll
        for (int h = 0; h < 8; ++h)
            S[j][h] = 0.f;
    const size_t second = (size_t)(n - m) * D; /* the second half-chain: rows [n - m, n) */
    float pf0[R1][HP], pf1[R1][HP];
    auto request = [&](unsigned long long cc, unsigned int g) __attribute__((always_inline)) {
        const T *const b = sample + (size_t)cc * n * D + d;
        /* the 128 offsets depend on the lane only: computed once ahead of the chain loop they would take 256 registers (and did:
         * 1.5 KB of scratch); an opaque copy of the lane index per request keeps them arithmetic beside their loads */
        unsigned int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
#pragma unroll
        for (int a = 0; a < R1; ++a)
#pragma unroll
            for (int h = 0; h < HP; ++h) {
                const unsigned int t = 64u * (unsigned int)a + lane_k + BLK * (g * (unsigned int)HP + (unsigned int)h), tc = t < m ? t : m - 1u;
                pf0[a][h] = (float)b[(size_t)tc * D];
                pf1[a][h] = (float)b[second + (size_t)tc * D];
            }
    };
    unsigned long long c = grp;
    float mu0 = 0.f, mu1 = 0.f;
    if (c < C) {
        mu0 = means[(size_t)c * D + d];
        mu1 = means[((size_t)c + (size_t)C) * D + d];
        request(c, 0u);
    }
    for (; c < C; c += n_grp) {
        float q0 = 0.f, q1 = 0.f;
        const float m0 = mu0, m1 = mu1;
        mm_cx y[R1];
#pragma unroll
        for (int a = 0; a < R1; ++a)
            y[a] = mm_cx{0.f, 0.f};
        unsigned int r1k = 0u; /* n1 k1 mod N1, stepped */
        for (unsigned int g = 0; g < G; ++g) {
            mm_cx w1[HP]; /* w_N1^(n1 k1): one value for the wave, through the scalar cache */
#pragma unroll
            for (int h = 0; h < HP; ++h) {
                w1[h] = wN[r1k * BLK];
                r1k += k1;
                r1k = r1k >= N1 ? r1k - N1 : r1k;
            }
#pragma unroll
            for (int a = 0; a < R1; ++a)
#pragma unroll
                for (int h = 0; h < HP; ++h) {
                    const bool in = 64u * (unsigned int)a + lane + BLK * (g * (unsigned int)HP + (unsigned int)h) < m;
                    const mm_cx v = mm_cx{in ? pf0[a][h] - m0 : 0.f, in ? pf1[a][h] - m1 : 0.f};
                    if (k1 == 0u) { /* every point of the chain passes here exactly once */
                        q0 = fmaf(v.re, v.re, q0);
                        q1 = fmaf(v.im, v.im, q1);
                    }
                    y[a] = mm_cx_add(y[a], mm_cx_mul(v, w1[h].re, w1[h].im));
                }
            /* the registers are free again: the next step's points, in flight behind whatever comes next */
            if (g + 1u < G) {
                request(c, g + 1u);
            } else if (c + n_grp < C) {
                mu0 = means[((size_t)c + n_grp) * D + d];
                mu1 = means[((size_t)c + n_grp + (size_t)C) * D + d];
                request(c + n_grp, 0u);
            }
        }
#pragma unroll
        for (int a = 0; a < R1; ++a)
            y[a] = mm_cx_mul(y[a], wt[a].re, wt[a].im);
        mm_fft_pass1_full<R1>(y, tw1_of, lds, (int)lane);
        MM_WAVE_LDS_SYNC();
        mm_cx v2[pl::J][8];
        mm_fft_pass2_load<R1>(v2, lds, (int)lane);
        MM_WAVE_LDS_SYNC();
        mm_fft_pass2_store<R1>(v2, tw2_of, lds, (int)lane);
        MM_WAVE_LDS_SYNC();
        mm_fft_pass3<R1>(lds, (int)lane, S); /* |Z|^2 of this chain's 2048 bins onto the wave's running spectrum */
        MM_WAVE_LDS_SYNC();
        if (k1 == 0u) {
            q0 = wave_sum_dpp_bcast(q0);
            q1 = wave_sum_dpp_bcast(q1);
            if (lane == 0u) {
                ssq[(size_t)c * D + d] = q0;
                ssq[((size_t)c + (size_t)C) * D + d] = q1;
            }
        }
    }
    float *const out = slabs + ((size_t)grp * D + d) * N;
#pragma unroll
    for (int j = 0; j < pl::J; ++j)
#pragma unroll
        for (int h = 0; h < 8; ++h)
            out[k1 + N1 * (unsigned int)mm_fft_bin<R1>((int)lane, j, h)] = S[j][h];
}

/* P[f D + d] = the sum of the tail kernel's partial totals, in f64 (fixed order) */
__global__ __launch_bounds__(256) void mm_fft_psum_kernel(const float *__restrict__ parts, unsigned int n_parts, size_t total,
                                                          double *__restrict__ P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total)
        return;
    double t0 = 0.0, t1 = 0.0;
    unsigned int p = 0;
    for (; p + 1 < n_parts; p += 2) {
        t0 += (double)parts[(size_t)p * total + i];
        t1 += (double)parts[(size_t)(p + 1) * total + i];
    }
    if (p < n_parts)
        t0 += (double)parts[(size_t)p * total + i];
    P[i] = t0 + t1;
}

/* the inverse for long transforms: as mm_fft_finish_kernel, with the spectrum and the cosine table read from memory
 * (N doubles each: beyond N = 2048 the two no longer fit a workgroup's default LDS) */
__global__ __launch_bounds__(256) void mm_fft_finish_long_kernel(const double *__restrict__ P, unsigned int D, unsigned int N,
                                                                 unsigned int m, const double *__restrict__ cos_tab,
                                                                 float *__restrict__ acov)
{
    __shared__ double red[256];
    const unsigned int d = blockIdx.x % D, lag0 = (blockIdx.x / D) * 32u, tid = threadIdx.x;
    const unsigned int l = tid & 31u, q = tid >> 5, lag = lag0 + l;
    /* cos(2 pi f lag / N) for 32 consecutive f from TWO table values and the recurrence c[f + 1] = 2 cos(theta) c[f] - c[f - 1]
     * (round 6: a table gather per term -- 64 different cache lines per load instruction -- made this kernel 3.1 ms of the
     * 23 ms of [16384, 20000, 3] and half of [4096, 200000, 3]'s 490; restarted from the table every 32 terms the recurrence's
     * error stays at 32^2 ulps of a double, far below the f32 result).  An arc is N / 8 >= 512 terms: whole runs of 32. */
    double a0 = 0.0, a1 = 0.0;
    const unsigned int f_lo = q * (N / 8), f_hi = f_lo + N / 8;
    /* N = 2048 N1 with any N1: the table index f lag mod N is stepped, not masked (lag < m <= N / 2; 32 lag < 2^32) */
    const double k2 = 2.0 * cos_tab[lag];
    const unsigned int step = (32u * lag) % N;
    unsigned int i0 = (unsigned int)(((unsigned long long)f_lo * lag) % N);
    for (unsigned int f = f_lo; f < f_hi; f += 32) {
        unsigned int i1 = i0 + lag;
        i1 = i1 >= N ? i1 - N : i1;
        double c0 = cos_tab[i0], c1 = cos_tab[i1];
        i0 += step;
        i0 = i0 >= N ? i0 - N : i0;
        a0 = fma(P[(size_t)f * D + d], c0, a0);
        a1 = fma(P[(size_t)(f + 1u) * D + d], c1, a1);
#pragma unroll
        for (unsigned int j = 2; j < 32; j += 2) {
            const double c2 = fma(k2, c1, -c0), c3 = fma(k2, c2, -c1);
            a0 = fma(P[(size_t)(f + j) * D + d], c2, a0);
            a1 = fma(P[(size_t)(f + j + 1u) * D + d], c3, a1);
            c0 = c2;
            c1 = c3;
        }
    }
    red[q * 32 + l] = a0 + a1;
    __syncthreads();
    if (q == 0 && lag < m) {
        const double r = ((red[l] + red[32 + l]) + (red[64 + l] + red[96 + l])) +
                         ((red[128 + l] + red[160 + l]) + (red[192 + l] + red[224 + l]));
        acov[(size_t)lag * D + d] = (float)(r / (double)N);
    }
}

/* The inverse of the accumulated spectrum, once per parameter for all chains: block (d, group of 32 lags) adds the
 * partial totals parts[p][f D + d] of the tail kernel in f64 and evaluates
 *     acov[lag D + d] = (1 / N) sum_f P[f] cos(2 pi f lag / N),   lag < m,
 * the circle cut into eight arcs per lag that are combined in a fixed order.  cos_tab: cos(2 pi i / N), i < N, in f64
 * (host-made, stats_fft_twiddles). */
__global__ __launch_bounds__(256) void mm_fft_finish_kernel(const float *__restrict__ parts, unsigned int n_parts,
                                                            unsigned int D, unsigned int N, unsigned int m,
                                                            const double *__restrict__ cos_tab, float *__restrict__ acov)
{
    extern __shared__ __attribute__((aligned(16))) double shd[];
    double *P = shd, *ct = shd + N, *red = shd + 2 * N; /* [N], [N], [8][32] */
    const unsigned int d = blockIdx.x % D, lag0 = (blockIdx.x / D) * 32u, tid = threadIdx.x;
    const size_t total = (size_t)N * D;
    for (unsigned int f = tid; f < N; f += 256) {
        double t0 = 0.0, t1 = 0.0;
        unsigned int p = 0;
        /* eight loads in flight, added in the order of the plain loop (even parts into t0, odd into t1) */
        for (; p + 7 < n_parts; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = parts[(size_t)(p + u) * total + (size_t)f * D + d];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                t0 += (double)v[u];
                t1 += (double)v[u + 1];
            }
        }
        for (; p + 1 < n_parts; p += 2) {
            t0 += (double)parts[(size_t)p * total + (size_t)f * D + d];
            t1 += (double)parts[(size_t)(p + 1) * total + (size_t)f * D + d];
        }
        if (p < n_parts)
            t0 += (double)parts[(size_t)p * total + (size_t)f * D + d];
        P[f] = t0 + t1;
        ct[f] = cos_tab[f];
    }
    __syncthreads();
    const unsigned int l = tid & 31u, q = tid >> 5, lag = lag0 + l;
    double a0 = 0.0, a1 = 0.0;
    const unsigned int f_lo = q * (N / 8), f_hi = f_lo + N / 8;
    for (unsigned int f = f_lo; f < f_hi; f += 2) {
        a0 = fma(P[f], ct[(f * lag) & (N - 1u)], a0);
        a1 = fma(P[f + 1], ct[((f + 1u) * lag) & (N - 1u)], a1);
    }
    red[q * 32 + l] = a0 + a1;
    __syncthreads();
    if (q == 0 && lag < m) {
        const double r = ((red[l] + red[32 + l]) + (red[64 + l] + red[96 + l])) +
                         ((red[128 + l] + red[160 + l]) + (red[192 + l] + red[224 + l]));
        acov[(size_t)lag * D + d] = (float)(r / (double)N);
    }
}

/* tiles of 8 time steps x 16 lags of one parameter's (t, lag) triangle (the kernel's own count) */
static unsigned int stats_tile_count(size_t m)
{
    unsigned int n = 0;
    for (size_t kb = 0; 16 * kb < m; ++kb)
        n += (unsigned int)((m - 16 * kb + 7) / 8);
    return n;
}

/* The tail of the statistics, one launch, two kinds of blocks.
 *
 * Blocks [0, nb_red): out[lag, d] = sum over waves of slabs[w, d, lag].  One wave per group of 64 outputs would
 * serialise n_slabs loads; instead a 256-thread block owns 64 consecutive outputs, its four waves each sum a quarter
 * of the slabs (coalesced 256-byte rows), and the four partial sums are combined in a fixed order -- bitwise
 * reproducible, no atomics.
 *
 * Blocks [nb_red, nb_red + D * MM_WB_CHUNKS) (only when wb_part != NULL): withinvar's cross-chain sums
 * (stats.rs:449-465), per parameter d and chunk of half-chains: sum(mean - s), sum((mean - s)^2), sum(ssq / n) in f64,
 * s = the first half-chain's mean (any shift near the overall mean avoids the cancellation of the raw moments; the
 * host turns the three totals into sum((mean - overall)^2) = Sq - Sd^2 / c2 and the sum of the biased variances). */
#define MM_WB_CHUNKS 32u
__global__ __launch_bounds__(256) void mm_stats_tail_kernel(const float *__restrict__ slabs, unsigned int n_slabs,
                                                            unsigned int D, unsigned int m, float *__restrict__ out,
                                                            unsigned int nb_red, unsigned int n_parts,
                                                            const float *__restrict__ means,
                                                            const float *__restrict__ ssq, unsigned long long c2,
                                                            float nf, double *__restrict__ wb_part)
{
    __shared__ float psum[4][64];
    __shared__ double red[3][256];
    if (blockIdx.x >= nb_red) {
        const unsigned int b = blockIdx.x - nb_red, d = b / MM_WB_CHUNKS, ch = b % MM_WB_CHUNKS, tid = threadIdx.x;
        const float shift = means[d];
        const unsigned long long per = (c2 + MM_WB_CHUNKS - 1) / MM_WB_CHUNKS;
        const unsigned long long lo = ch * per, hi = lo + per < c2 ? lo + per : c2;
        /* four strands per thread, their loads in flight together (one dependent pair of loads per trip made these blocks the
         * longest of the launch: 16 trips of a memory latency each at 131 072 half-chains), combined in a fixed order */
        double sd = 0.0, sq = 0.0, ws = 0.0;
        {
            double sd4[4] = {0.0, 0.0, 0.0, 0.0}, sq4[4] = {0.0, 0.0, 0.0, 0.0}, ws4[4] = {0.0, 0.0, 0.0, 0.0};
            unsigned long long c = lo + tid;
            for (; c + 768 < hi; c += 1024) {
                float mu[4], qq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mu[u] = means[(c + 256ull * u) * D + d];
                    qq[u] = ssq[(c + 256ull * u) * D + d];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double df = (double)(mu[u] - shift);
                    sd4[u] += df;
                    sq4[u] += df * df;
                    ws4[u] += (double)(qq[u] / nf); /* biased per-chain variance (quirk Q8) */
                }
            }
            for (; c < hi; c += 256) {
                const double df = (double)(means[c * D + d] - shift);
                sd4[0] += df;
                sq4[0] += df * df;
                ws4[0] += (double)(ssq[c * D + d] / nf);
            }
            sd = (sd4[0] + sd4[1]) + (sd4[2] + sd4[3]);
            sq = (sq4[0] + sq4[1]) + (sq4[2] + sq4[3]);
            ws = (ws4[0] + ws4[1]) + (ws4[2] + ws4[3]);
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
            wb_part[(size_t)b * 3 + tid] = red[tid][0];
        return;
    }
    /* the slabs are cut into n_parts ranges, block (part, group of 64 outputs) sums one range: with n_parts = 1 `out` is
     * the total [m, D]; with more (the synchronous entry point) out[part] are partial totals the host adds up in order --
     * ten blocks walking 4096 slabs each were latency-bound (83 us for 10 MB) */
    const unsigned int lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const unsigned int total = D * m;
    const unsigned int groups = (total + 63u) / 64u;
    const unsigned int part = blockIdx.x / groups, grp = blockIdx.x - part * groups;
    const unsigned int i = grp * 64u + lane; /* over d * m + lag */
    const unsigned int per_part = (n_slabs + n_parts - 1u) / n_parts;
    const unsigned int p_lo = part * per_part, p_hi = min(n_slabs, p_lo + per_part);
    float s = 0.f;
    if (i < total && p_lo < p_hi) {
        const unsigned int per = (p_hi - p_lo + 3u) / 4u;
        const unsigned int lo = p_lo + w * per, hi = min(p_hi, lo + per);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        unsigned int k = lo;
        for (; k + 3 < hi; k += 4) {
            s0 += slabs[(size_t)k * total + i];
            s1 += slabs[(size_t)(k + 1) * total + i];
            s2 += slabs[(size_t)(k + 2) * total + i];
            s3 += slabs[(size_t)(k + 3) * total + i];
        }
        for (; k < hi; ++k)
            s0 += slabs[(size_t)k * total + i];
        s = (s0 + s1) + (s2 + s3);
    }
    psum[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < total) {
        const float r = (psum[0][lane] + psum[1][lane]) + (psum[2][lane] + psum[3][lane]);
        const unsigned int d = i / m, lag = i - d * m;
        out[(size_t)part * total + (size_t)lag * D + d] = r;
    }
}

/* out[i] = parts[0][i] + parts[1][i] + ... in this order (the asynchronous entry point's second stage) */
__global__ void mm_parts_sum_kernel(const float *__restrict__ parts, unsigned int n_parts, unsigned int total,
                                    float *__restrict__ out)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total)
        return;
    float t = parts[i];
    for (unsigned int p = 1; p < n_parts; ++p)
        t += parts[(size_t)p * total + i];
    out[i] = t;
}

int check_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return MMCMC_ERR_NO_DEVICE;
    if (device < 0 || device >= n)
        return MMCMC_ERR_INVALID_ARG;
    return MMCMC_OK;
}

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

} // namespace

/* number of waves (= per-wave lag-sum slabs) the half-chain kernel is launched with */
static unsigned int stats_n_slabs(size_t n_chains)
{
    /* 4 waves per SIMD hide the LDS / global latencies of the half-chain loop; more only lengthens the slab reduction
     * (measured at [65536, 400, 3]: 1024 waves 1.65 ms, 2048 0.96, 4096 0.69 + 0.08, 8192 0.82 + 0.21) */
    const char *e = mm_tuning_env("MMCMC_STATS_WAVES");
    const size_t w = e ? (size_t)atoi(e) : 4096;
    return (unsigned int)std::min<size_t>(2 * n_chains, w ? w : 4096);
}

/* ---- which kernel reduces a sample: mmcmc_stats_set_kernel (include/mmcmc.h) ---- */
static std::atomic<int> g_stats_kernel{MMCMC_STATS_KERNEL_AUTO};
/* lag products (chains x dim x (n / 2)^2) the O(m^2) from-global-memory path may be asked for; 0 = no limit */
static std::atomic<uint64_t> g_stats_direct_work_limit{1ull << 46};

/* the power-spectrum path (mm_chain_fft_kernel): half-chains longer than 100 draws, as stats.rs:549 switches */
struct StatsFftPlan {
    bool use = false;
    int r1 = 0, dt = 0, wpe = 0;
    bool twl = false;
    unsigned int n_pt = 0, n_wg = 0, N = 0;
};

/* sel < 0: read the process-wide selector; the workspace is sized with sel = MMCMC_STATS_KERNEL_FFT (and for the direct
 * kernels as well), so that a mmcmc_stats_set_kernel() from another thread between the sizing and the launch of one call
 * can change which kernel runs but never make the buffer too small */
static StatsFftPlan stats_fft_plan(size_t n_chains, size_t n, size_t dim, int device, int sel_in = -1)
{
    StatsFftPlan p;
    const size_t m = n / 2;
    const int sel = sel_in >= 0 ? sel_in : g_stats_kernel.load(std::memory_order_relaxed);
    if (sel != MMCMC_STATS_KERNEL_AUTO && sel != MMCMC_STATS_KERNEL_FFT)
        return p;
    if (m > 1024 || m < 2 || (sel == MMCMC_STATS_KERNEL_AUTO && m <= 100))
        return p;
    p.use = true;
    p.r1 = m <= 256 ? 8 : m <= 512 ? 16 : 32;
    p.N = 64u * (unsigned int)p.r1;
    /* parameters per wave: as many as the registers hold without spilling (R1 = 8: 4 at two waves per SIMD, 3 at three;
     * R1 = 16: 3; R1 = 32: 1), the tiles of a larger dimension as even as possible */
    const size_t dt_max = p.r1 == 8 ? 4 : p.r1 == 16 ? 3 : 1;
    p.n_pt = (unsigned int)((dim + dt_max - 1) / dt_max);
    p.dt = (int)((dim + p.n_pt - 1) / p.n_pt);
    p.wpe = p.r1 == 8 && p.dt <= 3 ? 3 : 2;
    p.twl = false;
    if (const char *e = mm_tuning_env("MMCMC_FFT_WPE"))
        p.wpe = atoi(e);
    if (const char *e = mm_tuning_env("MMCMC_FFT_TWL"))
        p.twl = atoi(e) != 0;
    /* resident workgroups of four waves: waves per SIMD x CUs, shared between the parameter tiles */
    /* the workgroup count fixes which chains a wave sums in f32, so it must not depend on the device (CU count, partition
     * mode): 256 compute units' worth, an MI355X's, on every device -- R-hat / ESS are then the same bits everywhere */
    (void)device;
    const unsigned int wpe = (unsigned int)p.wpe;
    const unsigned int resident = 256u * wpe;
    p.n_wg = std::max(1u, resident / p.n_pt);
    if (const char *e = mm_tuning_env("MMCMC_FFT_NWG_MULT")) /* workgroups per resident slot (percent) */
        p.n_wg = std::max(1u, (unsigned int)((unsigned long long)p.n_wg * (unsigned int)atoi(e) / 100u));
    p.n_wg = (unsigned int)std::min<size_t>(p.n_wg, (n_chains + 3) / 4);
    return p;
}

/* the long-chain path (mm_chain_fft_long_kernel): half-chains of 1025 .. 16384 draws */
struct StatsLongPlan {
    bool use = false;
    bool res = false; /* round 6: one (residue, parameter) per wave, spectrum in registers (mm_chain_fft_res_kernel): N1 >= 4 */
    unsigned int res_r1 = 32; /* its inner transform: 64 res_r1 points */
    unsigned int N1 = 0, N = 0, n_wg = 0, waves = 1;
};
constexpr size_t kStatsLongMaxM = 131072; /* N = 2^18: the inverse (mm_fft_finish_long_kernel) is O(m N) per parameter */
/* LDS of a workgroup of the long-chain kernel: the spectrum, an exchange block per wave, at N1 = 4 the staged column */
static size_t stats_long_lds(unsigned int N, unsigned int waves, unsigned int N1)
{
    return ((size_t)N + (size_t)waves * 2 * (size_t)mm_fft_plan<32>::LDS_CX + (N1 == 4 ? (size_t)N : 0)) * sizeof(float);
}
static StatsLongPlan stats_long_plan(size_t n_chains, size_t n, size_t dim, int sel_in = -1)
{
    StatsLongPlan p;
    const size_t m = n / 2;
    const int sel = sel_in >= 0 ? sel_in : g_stats_kernel.load(std::memory_order_relaxed);
    if ((sel != MMCMC_STATS_KERNEL_AUTO && sel != MMCMC_STATS_KERNEL_FFT) || m <= 1024 || m > kStatsLongMaxM)
        return p;
    p.use = true;
    /* N = 2048 N1 >= 2 m, N1 the SMALLEST such count (round 6, late: it was the next power of two -- but a residue wave reads
     * the whole chain, so time goes as N1 x the data: m = 10 000 took N1 = 16 where 10 do, [16384, 20000, 3] 20.7 -> 9.0 ms, [16384, 40000, 3] 73 -> 31.7) */
    p.N1 = (unsigned int)std::max<size_t>(2, (m + 1023) / 1024);
#ifdef MMCMC_TUNING
    bool old_kernels = false, pow2 = mm_tuning_env("MMCMC_STATS_LONG_POW2") != nullptr; /* A / B against the power-of-two cut */
    if (mm_tuning_env("MMCMC_STATS_LONG_OLD") != nullptr && m <= 16384) /* ... and against round 5's kernels */
        old_kernels = pow2 = true;
    if (pow2) {
        unsigned int q = 2;
        while (q < p.N1)
            q *= 2;
        p.N1 = q;
    }
#else
    const bool old_kernels = false;
#endif
    p.N = 2048u * p.N1;
    if (p.N1 >= 3 && !old_kernels) {
        /* chain groups: as many as keep ONE resident round of one-wave workgroups busy -- one per SIMD by their registers, 1024
         * on an MI355X; the figure is a constant, not a device query, because the group count fixes the f32 summation grouping
         * -- in multiples of 8 (the kernel's XCD mapping), at most 512 */
        p.res = true;
        /* inner transforms of 2048 points, one wave per SIMD.  (1024 points and two waves per SIMD -- twice the residues, half
         * the registers; MMCMC_STATS_RES_R1_16 in measurement builds -- is slower throughout: [65536, 8000, 3] 16.8 against 7.7 ms,
         * [16384, 20000, 3] 31.3 against 20.7: every residue reads the chain again, profiles/r6za_stats_long_timing_r1_16.log) */
        p.res_r1 = 32;
#ifdef MMCMC_TUNING
        if (mm_tuning_env("MMCMC_STATS_RES_R1_16"))
            p.res_r1 = 16;
#endif
        const size_t n1r = (size_t)p.N / (64 * p.res_r1);
        const size_t per = dim * n1r;
        size_t g = per > 0 ? (p.res_r1 == 16 ? 2048 : 1024) / per : 8;
        g = std::min<size_t>(512, std::max<size_t>(8, g / 8 * 8));
        p.n_wg = (unsigned int)std::min<size_t>(g, n_chains);
        return p;
    }
    /* device-independent (the workgroup count fixes the f32 summation grouping): at most 512 chain groups, and no more than fit
     * the one-wave workgroups an MI355X holds at once with this kernel's LDS (1024 at N1 = 2: a launch of 1536 ran a second,
     * half-empty round), a multiple of 8 (the kernel's XCD mapping); fewer for few chains */
    /* waves per workgroup: one where the chain is cached in registers (N1 = 2) and where a second exchange block no longer
     * fits beside the spectrum (N1 = 16: 128 KB); four at N1 = 8 ([16384, 16000, 3]: 32.8 -> 9.4 ms, with the kernel's spills
     * gone) and at N1 = 4 (with the column staged in LDS: [65536, 8000, 3] 13.0 -> 12.1 ms; without it one / two / four waves
     * 15.8 / 13.0 / 14.7 ms) */
    p.waves = (p.N1 == 8 || p.N1 == 4) ? 4u : 1u;
    const size_t lds = stats_long_lds(p.N, p.waves, p.N1);
    const size_t resident = 256 * std::max<size_t>(1, (160u << 10) / lds); /* MI355X: 256 CUs x workgroups per CU by their LDS */
    size_t g = 512;
    if (dim > 0 && g * dim > resident)
        g = std::max<size_t>(8, resident / dim / 8 * 8);
    p.n_wg = (unsigned int)std::min<size_t>(g, n_chains);
    return p;
}

/* w_N^j = exp(-2 pi i j / N), j < N, as f32 pairs, then cos(2 pi j / N) as f64: one table per device and N1, never freed
 * (32 KB x N1; a process that used every N1 = 2 .. 128 once holds 260 MB of them per device) */
static const mm_cx *stats_long_tables(int device, unsigned int N1)
{
    static std::atomic<const mm_cx *> tab[64][129]; /* N1 <= kStatsLongMaxM / 1024 = 128 */
    const unsigned int k = N1 <= 128u ? N1 : 0u;
    const mm_cx *t = tab[device & 63][k].load(std::memory_order_acquire);
    if (t)
        return t;
    const unsigned int N = 2048u * N1;
    std::vector<mm_cx> h(2 * (size_t)N);
    static_assert(sizeof(mm_cx) == sizeof(double), "the cosine table shares the allocation");
    const double two_pi = 6.283185307179586476925286766559;
    for (unsigned int j = 0; j < N; ++j) {
        const double th = two_pi * (double)j / (double)N, cv = std::cos(th);
        h[j] = mm_cx{(float)cv, (float)-std::sin(th)};
        memcpy(&h[(size_t)N + j], &cv, sizeof(double));
    }
    mm_cx *d = nullptr;
    if (hipMalloc((void **)&d, h.size() * sizeof(mm_cx)) != hipSuccess)
        return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(mm_cx), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    const mm_cx *expected = nullptr;
    if (!tab[device & 63][k].compare_exchange_strong(expected, d, std::memory_order_acq_rel)) {
        (void)hipFree(d);
        return expected;
    }
    return d;
}

/* twiddles of the wave-level FFT: w_N^(lane b) as [R1][64], then w_64^(e g) as [8][8]; one table per device and
 * length, computed in f64 on the host at first use, never freed */
static const double *stats_fft_cos_table(const mm_cx *tw, int r1)
{
    return reinterpret_cast<const double *>(tw + (size_t)r1 * 64 + 64); /* behind the twiddles, 8-byte aligned */
}

static const mm_cx *stats_fft_twiddles(int device, int r1)
{
    static std::atomic<const mm_cx *> tab[64][3];
    const int k = r1 == 8 ? 0 : r1 == 16 ? 1 : 2;
    const mm_cx *t = tab[device & 63][k].load(std::memory_order_acquire);
    if (t)
        return t;
    const unsigned int N = 64u * (unsigned int)r1;
    /* [R1][64] + [8][8] complex, then cos(2 pi i / N), i < N, as doubles (two mm_cx slots... one double per slot) */
    std::vector<mm_cx> h((size_t)r1 * 64 + 64 + N);
    static_assert(sizeof(mm_cx) == sizeof(double), "the cosine table shares the allocation");
    for (unsigned int i = 0; i < N; ++i) {
        const double cv = std::cos(6.283185307179586476925286766559 * (double)i / (double)N);
        memcpy(&h[(size_t)r1 * 64 + 64 + i], &cv, sizeof(double));
    }
    const double two_pi = 6.283185307179586476925286766559;
    for (int b = 0; b < r1; ++b)
        for (int l = 0; l < 64; ++l) {
            const double th = two_pi * (double)((unsigned int)(l * b) % N) / (double)N;
            h[(size_t)b * 64 + l] = mm_cx{(float)std::cos(th), (float)-std::sin(th)};
        }
    for (int g = 0; g < 8; ++g)
        for (int e = 0; e < 8; ++e) {
            const double th = two_pi * (double)(e * g) / 64.0;
            h[(size_t)r1 * 64 + g * 8 + e] = mm_cx{(float)std::cos(th), (float)-std::sin(th)};
        }
    mm_cx *d = nullptr;
    if (hipMalloc((void **)&d, h.size() * sizeof(mm_cx)) != hipSuccess)
        return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(mm_cx), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    const mm_cx *expected = nullptr;
    if (!tab[device & 63][k].compare_exchange_strong(expected, d, std::memory_order_acq_rel)) {
        (void)hipFree(d); /* another thread was first */
        return expected;
    }
    return d;
}

template <class T, int R1, int DT, int WPE, bool TWL, bool TOP>
static int stats_fft_launch1(const StatsFftPlan &p, const void *sample, size_t n_chains, size_t n, size_t dim, size_t m,
                             const mm_cx *tw, float *means, float *ssq, float *slabs, hipStream_t stream)
{
    const size_t lds = ((size_t)4 * mm_fft_plan<R1>::LDS_CX + (TWL ? R1 * 64 + 64 : 0)) * sizeof(mm_cx);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(attr_set.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
            MM_HIP(hipFuncSetAttribute((const void *)mm_chain_fft_kernel<T, R1, DT, WPE, TWL, TOP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((mm_chain_fft_kernel<T, R1, DT, WPE, TWL, TOP>), dim3(p.n_wg * p.n_pt), dim3(256), lds, stream,
                       (const T *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim,
                       (unsigned int)m, p.n_pt, tw, means, ssq, slabs);
    return MMCMC_OK;
}

template <class T>
static int stats_fft_launch(const StatsFftPlan &p, const void *sample, size_t n_chains, size_t n, size_t dim, size_t m,
                            const mm_cx *tw, float *means, float *ssq, float *slabs, hipStream_t stream)
{
    const bool top = m > 64 * (size_t)(p.r1 / 2 - 1); /* only a lane's last point can be padding */
#define MM_FFT_CASE(R1V, DTV, WPEV, TWLV)                                                                           \
    if (p.r1 == R1V && p.dt == DTV && p.wpe == WPEV && p.twl == TWLV)                                               \
    return top ? stats_fft_launch1<T, R1V, DTV, WPEV, TWLV, true>(p, sample, n_chains, n, dim, m, tw, means, ssq, slabs, stream) \
               : stats_fft_launch1<T, R1V, DTV, WPEV, TWLV, false>(p, sample, n_chains, n, dim, m, tw, means, ssq, slabs, stream)
    MM_FFT_CASE(8, 1, 3, false);
    MM_FFT_CASE(8, 2, 3, false);
    MM_FFT_CASE(8, 3, 3, false);
    MM_FFT_CASE(8, 4, 2, false);
    MM_FFT_CASE(16, 1, 2, false);
    MM_FFT_CASE(16, 2, 2, false);
    MM_FFT_CASE(16, 3, 2, false);
    MM_FFT_CASE(32, 1, 2, false);
#ifdef MMCMC_TUNING /* experiments: MMCMC_FFT_WPE, MMCMC_FFT_TWL */
    MM_FFT_CASE(8, 3, 4, false);
    MM_FFT_CASE(8, 3, 4, true);
    MM_FFT_CASE(8, 3, 3, true);
    MM_FFT_CASE(16, 3, 3, true);
    MM_FFT_CASE(16, 3, 2, true);
#endif
#undef MM_FFT_CASE
    return MMCMC_ERR_UNSUPPORTED;
}

/* floats of device workspace stats_partials_impl wants behind `slabs_ws` */
static size_t stats_ws_floats(size_t n_chains, size_t n, size_t dim, unsigned int n_parts, int device)
{
    const size_t m = n / 2;
    size_t need = (size_t)stats_n_slabs(n_chains) * dim * m;
    const StatsFftPlan p = stats_fft_plan(n_chains, n, dim, device, MMCMC_STATS_KERNEL_FFT); /* whichever kernel runs */
    if (p.use)
        need = std::max(need, ((size_t)p.n_wg + n_parts) * dim * p.N);
    const StatsLongPlan lp = stats_long_plan(n_chains, n, dim, MMCMC_STATS_KERNEL_FFT);
    if (lp.use) /* slabs | partial totals | their f64 sum */
        need = std::max(need, ((size_t)lp.n_wg + n_parts + 2) * dim * lp.N);
    return need;
}

/* wb_part: NULL, or [dim][MM_WB_CHUNKS][3] doubles for the cross-chain sums (single-GPU path) */
static int stats_partials_impl(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, float *means,
                               float *ssq, float *acov_sum, double *wb_part, float *slabs_ws, unsigned int n_parts,
                               int device, void *stream_v, unsigned int *parts_out, float *final_out = nullptr)
{
    /* final_out (may be NULL): where a path that produces the total itself writes it; *parts_out = 0 then */
    if (parts_out)
        *parts_out = n_parts;
    if (!sample || !means || !ssq || !acov_sum || n_chains == 0 || dim == 0 ||
        (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    const size_t m = n / 2;
    if (m < 1 || n >= (1ull << 31) || dim >= (1u << 16))
        return MMCMC_ERR_SHAPE;
    int st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DevGuard g(device);
    hipStream_t stream = (hipStream_t)stream_v;
    const StatsFftPlan fp = stats_fft_plan(n_chains, n, dim, device);
    if (fp.use) {
        /* power spectrum per workgroup -> bin totals in n_parts partial sums (the tail kernel, with bins for lags) ->
         * one inverse per parameter: acov_sum[0] is the total, *parts_out = 1 */
        const mm_cx *tw = stats_fft_twiddles(device, fp.r1);
        if (!tw)
            return (int)hipErrorOutOfMemory;
        const size_t slab_floats = (size_t)fp.n_wg * dim * fp.N, part_floats = (size_t)n_parts * dim * fp.N;
        float *ws = slabs_ws;
        if (!ws)
            MM_HIP(hipMallocAsync((void **)&ws, (slab_floats + part_floats) * sizeof(float), stream));
        float *bins = ws + slab_floats;
        st = dtype == MMCMC_F32
                 ? stats_fft_launch<float>(fp, sample, n_chains, n, dim, m, tw, means, ssq, ws, stream)
                 : stats_fft_launch<double>(fp, sample, n_chains, n, dim, m, tw, means, ssq, ws, stream);
        if (st != MMCMC_OK)
            return st;
        MM_HIP(hipGetLastError());
        const unsigned int total_b = (unsigned int)(dim * fp.N);
        const unsigned int nb_red = (total_b + 63) / 64 * n_parts, nb_wb = wb_part ? (unsigned int)dim * MM_WB_CHUNKS : 0u;
        hipLaunchKernelGGL(mm_stats_tail_kernel, dim3(nb_red + nb_wb), dim3(256), 0, stream, ws, fp.n_wg,
                           (unsigned int)dim, fp.N, bins, nb_red, n_parts, means, ssq,
                           (unsigned long long)(2 * n_chains), (float)m, wb_part);
        MM_HIP(hipGetLastError());
        hipLaunchKernelGGL(mm_fft_finish_kernel, dim3((unsigned int)dim * (unsigned int)((m + 31) / 32)), dim3(256),
                           (2 * (size_t)fp.N + 256) * sizeof(double), stream, bins, n_parts, (unsigned int)dim, fp.N,
                           (unsigned int)m, stats_fft_cos_table(tw, fp.r1), final_out ? final_out : acov_sum);
        MM_HIP(hipGetLastError());
        if (!slabs_ws)
            MM_HIP(hipFreeAsync(ws, stream));
        if (parts_out)
            *parts_out = final_out ? 0 : 1;
        return MMCMC_OK;
    }
    const StatsLongPlan lp = stats_long_plan(n_chains, n, dim);
    if (lp.use) {
        const mm_cx *tw = stats_fft_twiddles(device, 32), *wN = stats_long_tables(device, lp.N1);
        if (!tw || !wN)
            return (int)hipErrorOutOfMemory;
        const size_t slab_floats = (size_t)lp.n_wg * dim * lp.N, part_floats = (size_t)n_parts * dim * lp.N,
                     p_floats = 2 * (size_t)dim * lp.N;
        float *ws = slabs_ws;
        if (!ws)
            MM_HIP(hipMallocAsync((void **)&ws, (slab_floats + part_floats + p_floats) * sizeof(float), stream));
        float *bins = ws + slab_floats;
        double *P = reinterpret_cast<double *>(ws + ((slab_floats + part_floats + 1) / 2) * 2);
        const size_t lds = stats_long_lds(lp.N, lp.waves, lp.N1);
        const unsigned int grid = lp.n_wg * (unsigned int)dim;
        hipError_t le = hipSuccess;
        if (lp.res) {
            /* means by a streaming pass (dim <= 16: 64 dim threads per half-chain; wider samples through the any-length moments
             * kernel), then one wave per (chain group, residue, parameter) */
            const unsigned int n1r = lp.N / (64u * lp.res_r1);
            if ((uint64_t)2 * n_chains >= (1ull << 31) || (uint64_t)lp.n_wg * dim * n1r >= (1ull << 31))
                return MMCMC_ERR_SHAPE;
            const unsigned int g_mean = (unsigned int)(2 * n_chains), g_res = lp.n_wg * (unsigned int)dim * n1r;
            const size_t lds_res = (size_t)(lp.res_r1 == 16 ? mm_fft_plan<16>::LDS_CX : mm_fft_plan<32>::LDS_CX) * sizeof(mm_cx);
            const mm_cx *tw_res = lp.res_r1 == 16 ? stats_fft_twiddles(device, 16) : tw;
            if (!tw_res)
                return (int)hipErrorOutOfMemory;
#define MM_RES_LAUNCH(TT)                                                                                           \
    do {                                                                                                            \
        if (dim <= 16)                                                                                              \
            hipLaunchKernelGGL(mm_half_chain_means_kernel<TT>, dim3(g_mean), dim3(64u * (unsigned int)dim), 0, stream, \
                               (const TT *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, \
                               (unsigned int)m, means);                                                             \
        else                                                                                                        \
            hipLaunchKernelGGL(mm_half_chain_moments_any_kernel<TT>, dim3(g_mean * (unsigned int)dim), dim3(64), 0, stream, \
                               (const TT *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, \
                               (unsigned int)m, means, ssq);                                                        \
        if (lp.res_r1 == 16)                                                                                        \
            hipLaunchKernelGGL((mm_chain_fft_res_kernel<TT, 16, 2, 2>), dim3(g_res), dim3(64), lds_res, stream, (const TT *)sample, \
                               (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, n1r, \
                               lp.n_wg, tw_res, wN, means, ssq, ws);                                                \
        else                                                                                                        \
            hipLaunchKernelGGL((mm_chain_fft_res_kernel<TT, 32, 2, 1>), dim3(g_res), dim3(64), lds_res, stream, (const TT *)sample, \
                               (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, n1r, \
                               lp.n_wg, tw_res, wN, means, ssq, ws);                                                \
    } while (0)
            if (dim > 16 && (uint64_t)2 * n_chains * dim >= (1ull << 31))
                return MMCMC_ERR_SHAPE;
            if (dtype == MMCMC_F32)
                MM_RES_LAUNCH(float);
            else
                MM_RES_LAUNCH(double);
#undef MM_RES_LAUNCH
        } else {
#define MM_LONG_LAUNCH(TT, NN, WW)                                                                                  \
    do {                                                                                                            \
        if (lds > 64 * 1024)                                                                                        \
            le = hipFuncSetAttribute((const void *)mm_chain_fft_long_kernel<TT, NN, WW>,                            \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        if (le == hipSuccess)                                                                                       \
            hipLaunchKernelGGL((mm_chain_fft_long_kernel<TT, NN, WW>), dim3(grid), dim3(64 * WW), lds, stream,      \
                               (const TT *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, \
                               (unsigned int)m, lp.n_wg, tw, wN, means, ssq, ws);                                   \
    } while (0)
#ifdef MMCMC_TUNING /* round 5's kernels for N1 = 4, 8, 16: measurement builds only (MMCMC_STATS_LONG_OLD) */
#define MM_LONG_PICK(TT)                                                                                            \
    do {                                                                                                            \
        switch (lp.N1) {                                                                                            \
        case 2: MM_LONG_LAUNCH(TT, 2, 1); break;                                                                    \
        case 4: MM_LONG_LAUNCH(TT, 4, 4); break;                                                                    \
        case 8: MM_LONG_LAUNCH(TT, 8, 4); break;                                                                    \
        default: MM_LONG_LAUNCH(TT, 16, 1); break;                                                                  \
        }                                                                                                           \
    } while (0)
#else
#define MM_LONG_PICK(TT) MM_LONG_LAUNCH(TT, 2, 1)
#endif
        if (dtype == MMCMC_F32)
            MM_LONG_PICK(float);
        else
            MM_LONG_PICK(double);
#undef MM_LONG_PICK
#undef MM_LONG_LAUNCH
        }
        MM_HIP(le);
        MM_HIP(hipGetLastError());
        const unsigned int total_b = (unsigned int)(dim * lp.N);
        const unsigned int nb_red = (total_b + 63) / 64 * n_parts, nb_wb = wb_part ? (unsigned int)dim * MM_WB_CHUNKS : 0u;
        hipLaunchKernelGGL(mm_stats_tail_kernel, dim3(nb_red + nb_wb), dim3(256), 0, stream, ws, lp.n_wg, (unsigned int)dim, lp.N,
                           bins, nb_red, n_parts, means, ssq, (unsigned long long)(2 * n_chains), (float)m, wb_part);
        MM_HIP(hipGetLastError());
        hipLaunchKernelGGL(mm_fft_psum_kernel, dim3((total_b + 255) / 256), dim3(256), 0, stream, bins, n_parts, (size_t)total_b, P);
        hipLaunchKernelGGL(mm_fft_finish_long_kernel, dim3((unsigned int)dim * (unsigned int)((m + 31) / 32)), dim3(256), 0, stream, P,
                           (unsigned int)dim, lp.N, (unsigned int)m, reinterpret_cast<const double *>(wN + lp.N),
                           final_out ? final_out : acov_sum);
        MM_HIP(hipGetLastError());
        if (!slabs_ws)
            MM_HIP(hipFreeAsync(ws, stream));
        if (parts_out)
            *parts_out = final_out ? 0 : 1;
        return MMCMC_OK;
    }
    unsigned int n_slabs = stats_n_slabs(n_chains);
    /* the per-wave lag sums: the caller's workspace (synchronous path) or a stream-ordered allocation */
    float *slabs = slabs_ws;
    /* lag sums on the matrix cores (one 16 x 16 tile per 256 lags) unless the half-chain is too long for LDS */
    const unsigned int n_tiles = (unsigned int)((m + 255) / 256);
    const size_t row_len = 240 + 256 * (size_t)(n_tiles - 1) + m + 36, row_pitch = row_len + row_len / 16 + 1;
    const size_t lds_mfma = ((size_t)dim * row_pitch + (size_t)dim * m) * sizeof(float);
    const int sel = g_stats_kernel.load(std::memory_order_relaxed); /* mmcmc_stats_set_kernel */
    const bool force_direct = sel == MMCMC_STATS_KERNEL_DIRECT, force_mfma = sel == MMCMC_STATS_KERNEL_MFMA,
               force_tile = sel == MMCMC_STATS_KERNEL_TILE;
    /* register tiles on the vector ALU where a lane's share of the D x tiles(m) tiles fits its registers (at most 8
     * tiles of 16 lag sums: [., 400, 3] just fits) */
    const size_t tiles_total = dim * (size_t)stats_tile_count(m);
    const unsigned int tpl = (unsigned int)((tiles_total + 63) / 64);
    const size_t pitch_t = 12 * ((m + 8 + 16 + 8 + 7) / 8) + 4; /* the kernel's row pitch */
    const size_t lds_tile = (dim * pitch_t + dim * m + 64 * 16) * sizeof(float);
    /* one parameter per wave, a parameter's tiles dealt to `parts` waves of three slots (mm_half_chain_tile1_kernel: four
     * waves per SIMD): half-chains up to 512 long -- eight column loads per lane, 16 waves' rows in a CU's LDS --;
     * MMCMC_STATS_KERNEL=tile keeps the all-parameters kernel where that one applies */
    const unsigned int tiles_1 = stats_tile_count(m);
    const unsigned int tpl_1 = tiles_1 <= 64 ? 1u : tiles_1 <= 128 ? 2u : 3u;
    const unsigned int parts = (tiles_1 + 64u * tpl_1 - 1u) / (64u * tpl_1);
    const size_t lds_1 = (pitch_t + m + 64 * 16 + 64) * sizeof(float);
    const bool no_force = !force_direct && !force_mfma;
    const bool per_param = no_force && m <= 512 && 16 * lds_1 <= 160 * 1024 && (size_t)parts * dim <= n_slabs && !force_tile;
    if (per_param || (tpl <= 8 && lds_tile <= 40 * 1024 && /* 8 tiles = 128 lag sums + operands = 218 registers: two waves per SIMD */ no_force)) {
        if (!slabs)
            MM_HIP(hipMallocAsync((void **)&slabs, (size_t)n_slabs * dim * m * sizeof(float), stream));
        if (per_param) {
            /* slabs x parts x parameters waves = the resident 16 per CU; the tail sums slabs x parts rows per (d, lag) */
            const size_t cap = mm_tuning_env("MMCMC_STATS_WAVES") ? (size_t)n_slabs : 4096 / dim;
            n_slabs = (unsigned int)std::max<size_t>(1, std::min<size_t>(n_slabs, cap) / parts);
#define MM_TILE1_LAUNCH(TT, TPLV, NPREV)                                                                            \
    hipLaunchKernelGGL((mm_half_chain_tile1_kernel<TT, TPLV, NPREV>), dim3(n_slabs * parts * (unsigned int)dim), dim3(64), lds_1, \
                       stream, (const TT *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, \
                       (unsigned int)m, parts, means, ssq, slabs)
#define MM_TILE1_PICK(TT)                                                                                           \
    do {                                                                                                            \
        if (m > 256)                                                                                                \
            MM_TILE1_LAUNCH(TT, 3, 8);                                                                              \
        else if (tpl_1 == 1)                                                                                        \
            MM_TILE1_LAUNCH(TT, 1, 4);                                                                              \
        else if (tpl_1 == 2)                                                                                        \
            MM_TILE1_LAUNCH(TT, 2, 4);                                                                              \
        else                                                                                                        \
            MM_TILE1_LAUNCH(TT, 3, 4);                                                                              \
    } while (0)
            if (dtype == MMCMC_F32)
                MM_TILE1_PICK(float);
            else
                MM_TILE1_PICK(double);
            n_slabs *= parts; /* rows the tail kernel sums */
#undef MM_TILE1_PICK
#undef MM_TILE1_LAUNCH
        } else {
        /* this kernel holds two waves per SIMD (250 registers): exactly the resident waves, each with a longer list of
         * half-chains (measured at [65 536, 400, 3]: 1024 waves 0.62 ms, 2048 0.39, 4096 0.43, 8192 0.54) */
        if (!mm_tuning_env("MMCMC_STATS_WAVES"))
            n_slabs = std::min(n_slabs, 2048u);
        /* 16-byte loads where every half-chain's [m, D] block starts on a 16-byte boundary and is whole vectors long */
        const bool vec_ok = dtype == MMCMC_F32 && (n * dim) % 4 == 0 && ((n - m) * dim) % 4 == 0 && (m * dim) % 4 == 0 &&
                            ((uintptr_t)sample % 16) == 0 && m * dim <= 4 * 256;
#define MM_TILE_LAUNCH1(TT, TPLV, VECV)                                                                             \
    hipLaunchKernelGGL((mm_half_chain_tile_kernel<TT, TPLV, VECV>), dim3(n_slabs), dim3(64), lds_tile, stream,      \
                       (const TT *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim,        \
                       (unsigned int)m, means, ssq, slabs)
#define MM_TILE_LAUNCH(TT, TPLV)                                                                                    \
    do {                                                                                                            \
        if (sizeof(TT) == 4 && vec_ok)                                                                              \
            MM_TILE_LAUNCH1(float, TPLV, true);                                                                     \
        else                                                                                                        \
            MM_TILE_LAUNCH1(TT, TPLV, false);                                                                       \
    } while (0)
#define MM_TILE_PICK(TT)                                                                                            \
    do {                                                                                                            \
        if (tpl <= 2)                                                                                               \
            MM_TILE_LAUNCH(TT, 2);                                                                                  \
        else if (tpl <= 4)                                                                                          \
            MM_TILE_LAUNCH(TT, 4);                                                                                  \
        else                                                                                                        \
            MM_TILE_LAUNCH(TT, 8);                                                                                  \
    } while (0)
        if (dtype == MMCMC_F32)
            MM_TILE_PICK(float);
        else
            MM_TILE_PICK(double);
#undef MM_TILE_PICK
#undef MM_TILE_LAUNCH
#undef MM_TILE_LAUNCH1
        }
    } else if (lds_mfma <= 64 * 1024 && !force_direct) {
        if (!slabs)
            MM_HIP(hipMallocAsync((void **)&slabs, (size_t)n_slabs * dim * m * sizeof(float), stream));
        if (dtype == MMCMC_F32)
            hipLaunchKernelGGL(mm_half_chain_mfma_kernel<float>, dim3(n_slabs), dim3(64), lds_mfma, stream,
                               (const float *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim,
                               (unsigned int)m, n_tiles, means, ssq, slabs);
        else
            hipLaunchKernelGGL(mm_half_chain_mfma_kernel<double>, dim3(n_slabs), dim3(64), lds_mfma, stream,
                               (const double *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim,
                               (unsigned int)m, n_tiles, means, ssq, slabs);
    } else {
        const unsigned int m_pad = (unsigned int)(m + 64 + ((m + 64) % 2 == 0 ? 1 : 0)); /* odd row pitch */
        const size_t lds = ((size_t)dim * m_pad + (size_t)dim * m) * sizeof(float);
        if (lds > 160 * 1024 || m > 16384) {
            /* nothing stages such a half-chain (or only one wave per half-chain would: [2, 32770, 1] 46 ms against 1 ms here):
             * moments and lag sums straight from global memory, any length */
            /* 32-bit quantities of this path: the moments kernel's grid (2 C D workgroups) and dim * m (the tail kernel's
             * element count); and a WORK bound -- the lag sums here cost C * D * m^2 products like the reference's
             * brute-force branch (stats.rs:622-654), ~6e12 per second on this device: past the limit (default 2^46, about ten
             * seconds; mmcmc_stats_set_direct_work_limit moves or removes it) the call is refused instead of occupying the device
             * for minutes.  Round 6: under AUTO / FFT this path is reached only beyond 131 072 draws per half-chain -- up to
             * there the power spectrum is taken residue by residue (mm_chain_fft_res_kernel), e.g. [65536, 40000, 3] */
            if ((uint64_t)dim * m >= (1ull << 32) || (uint64_t)2 * n_chains * dim >= (1ull << 31))
                return MMCMC_ERR_SHAPE;
            {
                const uint64_t limit = g_stats_direct_work_limit.load(std::memory_order_relaxed); /* mmcmc_stats_set_direct_work_limit */
                if (limit != 0 && (long double)n_chains * (long double)dim * (long double)m * (long double)m > (long double)limit)
                    return MMCMC_ERR_UNSUPPORTED;
            }
            n_slabs = std::min(n_slabs, 16u);
            if (!slabs)
                MM_HIP(hipMallocAsync((void **)&slabs, (size_t)n_slabs * dim * m * sizeof(float), stream));
            const unsigned int g_mom = (unsigned int)(2 * n_chains * dim);
            const dim3 g_lag((unsigned int)((m + 255) / 256), (unsigned int)dim, n_slabs);
            if (dtype == MMCMC_F32) {
                hipLaunchKernelGGL(mm_half_chain_moments_any_kernel<float>, dim3(g_mom), dim3(64), 0, stream, (const float *)sample,
                                   (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, means, ssq);
                hipLaunchKernelGGL(mm_lag_sums_any_kernel<float>, g_lag, dim3(256), 0, stream, (const float *)sample,
                                   (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, means, n_slabs,
                                   slabs);
            } else {
                hipLaunchKernelGGL(mm_half_chain_moments_any_kernel<double>, dim3(g_mom), dim3(64), 0, stream, (const double *)sample,
                                   (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, means, ssq);
                hipLaunchKernelGGL(mm_lag_sums_any_kernel<double>, g_lag, dim3(256), 0, stream, (const double *)sample,
                                   (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, means, n_slabs,
                                   slabs);
            }
        } else {
        if (!slabs)
            MM_HIP(hipMallocAsync((void **)&slabs, (size_t)n_slabs * dim * m * sizeof(float), stream));
        if (lds > 64 * 1024) {
            const void *fn = dtype == MMCMC_F32 ? (const void *)mm_half_chain_kernel<float>
                                                : (const void *)mm_half_chain_kernel<double>;
            MM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (dtype == MMCMC_F32)
            hipLaunchKernelGGL(mm_half_chain_kernel<float>, dim3(n_slabs), dim3(64), lds, stream, (const float *)sample,
                               (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim, (unsigned int)m, m_pad,
                               means, ssq, slabs);
        else
            hipLaunchKernelGGL(mm_half_chain_kernel<double>, dim3(n_slabs), dim3(64), lds, stream,
                               (const double *)sample, (unsigned long long)n_chains, (unsigned int)n, (unsigned int)dim,
                               (unsigned int)m, m_pad, means, ssq, slabs);
        }
    }
    MM_HIP(hipGetLastError());
    const unsigned int total = (unsigned int)(dim * m);
    /* acov_sum: [n_parts][m, D] partial totals (n_parts = 1: the total) */
    const unsigned int nb_red = (total + 63) / 64 * n_parts, nb_wb = wb_part ? (unsigned int)dim * MM_WB_CHUNKS : 0u;
    hipLaunchKernelGGL(mm_stats_tail_kernel, dim3(nb_red + nb_wb), dim3(256), 0, stream, slabs, n_slabs,
                       (unsigned int)dim, (unsigned int)m, acov_sum, nb_red, n_parts, means, ssq,
                       (unsigned long long)(2 * n_chains), (float)m, wb_part);
    MM_HIP(hipGetLastError());
    if (!slabs_ws)
        MM_HIP(hipFreeAsync(slabs, stream));
    return MMCMC_OK;
}

/* Device workspace of the synchronous entry point (it returns after its kernels have finished, so one buffer per host
 * thread and device can be reused call after call): grown on demand, never freed. */
static float *stats_workspace(int device, size_t n_floats)
{
    struct Ws {
        float *p = nullptr;
        size_t cap = 0;
    };
    static thread_local Ws ws[64];
    Ws &w = ws[device & 63];
    if (n_floats > w.cap) {
        if (w.p)
            (void)hipFree(w.p);
        w.p = nullptr;
        w.cap = 0;
        if (hipMalloc((void **)&w.p, n_floats * sizeof(float)) != hipSuccess)
            return nullptr;
        w.cap = n_floats;
    }
    return w.p;
}

/* wait for the work queued so far by polling an event: a blocking hipStreamSynchronize sleeps on an interrupt and
 * wakes up tens of microseconds after a sub-millisecond reduction has finished */
static hipError_t stats_wait(int device, hipStream_t stream)
{
    /* one event per host thread AND device (an event belongs to the device that was current when it was created: a
     * single-threaded caller that reduces on device 0 and then on device 1 must not record a device-0 event on a
     * device-1 stream); the caller holds a DevGuard on `device` */
    static thread_local hipEvent_t evs[64] = {};
    hipEvent_t &ev = evs[device & 63];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        return hipStreamSynchronize(stream);
    hipError_t e = hipEventRecord(ev, stream);
    if (e != hipSuccess)
        return e;
    for (unsigned int i = 0; i < (1u << 24); ++i) {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady)
            return e;
    }
    return hipEventSynchronize(ev);
}

/* pinned staging buffer for the few KB that come back, one per host thread, grown on demand, never freed */
static float *stats_pinned(size_t n_floats)
{
    static thread_local float *buf = nullptr;
    static thread_local size_t cap = 0;
    if (n_floats > cap) {
        if (buf)
            (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
        if (hipHostMalloc((void **)&buf, n_floats * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return nullptr;
        cap = n_floats;
    }
    return buf;
}

extern "C" {

int mmcmc_stats_set_direct_work_limit(uint64_t max_lag_products)
{
    g_stats_direct_work_limit.store(max_lag_products, std::memory_order_relaxed);
    return MMCMC_OK;
}

int mmcmc_stats_set_kernel(int kind)
{
    if (kind < MMCMC_STATS_KERNEL_AUTO || kind > MMCMC_STATS_KERNEL_DIRECT)
        return MMCMC_ERR_INVALID_ARG;
    g_stats_kernel.store(kind, std::memory_order_relaxed);
    return MMCMC_OK;
}

int mmcmc_stats_partials(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, float *means,
                         float *ssq, float *acov_sum, int device, void *stream_v)
{
    /* Asynchronous: the caller's stream orders everything.  The work buffer (per-wave slabs + 16 partial totals) is
     * kept per host thread and reused as long as the calls come on the same stream of the same device -- then the
     * previous call's kernels are done with it before this call's start; any change of stream or device, or growth,
     * synchronises the device first.  (Two stream-ordered pool allocations per call were 0.25 ms.) */
    constexpr unsigned int kParts = 16;
    struct Ws {
        float *p = nullptr;
        size_t cap = 0;
        int device = -1;
        void *stream = nullptr;
    };
    static thread_local Ws w;
    if (!sample || n_chains == 0 || dim == 0 || n / 2 < 1)
        return MMCMC_ERR_INVALID_ARG;
    int st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DevGuard g(device);
    const size_t m = n / 2, total = m * dim;
    const size_t need = stats_ws_floats(n_chains, n, dim, kParts, device) + (size_t)kParts * total;
    if (w.p && (w.device != device || w.stream != stream_v || need > w.cap)) {
        DevGuard gw(w.device);
        (void)hipDeviceSynchronize();
        if (need > w.cap || w.device != device) {
            (void)hipFree(w.p);
            w.p = nullptr;
            w.cap = 0;
        }
    }
    if (!w.p) {
        MM_HIP(hipMalloc((void **)&w.p, need * sizeof(float)));
        w.cap = need;
    }
    w.device = device;
    w.stream = stream_v;
    float *parts = w.p, *slabs = w.p + (size_t)kParts * total;
    unsigned int n_written = kParts;
    st = stats_partials_impl(sample, dtype, n_chains, n, dim, means, ssq, parts, nullptr, slabs, kParts, device, stream_v,
                             &n_written, acov_sum);
    if (st != MMCMC_OK)
        return st;
    if (n_written == 0) /* the power-spectrum path wrote acov_sum itself */
        return MMCMC_OK;
    hipLaunchKernelGGL(mm_parts_sum_kernel, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_v, parts, n_written, (unsigned int)total, acov_sum);
    MM_HIP(hipGetLastError());
    return MMCMC_OK;
}

/* stats.rs:449-465 (withinvar), :425-427 (rhat), :509-545 (ess) on the gathered sufficient statistics, f32.
 * means, ssq: [c2, dim] in splitcat order (first halves of all chains, then second halves); acov_sum [m, dim]. */
/* from the cross-chain sums to R-hat and ESS of parameter d (stats.rs:459-465, 425-427, 509-545), f32 */
static float stats_ess_one(float w, float v, const float *acov_sum, size_t c2, size_t m, size_t dim, size_t d,
                           std::vector<float> &rho);
static void stats_finish_one(float dsum, float wsum, const float *acov_sum, size_t c2, size_t m, size_t dim, size_t d,
                             std::vector<float> &rho, float *rhat, float *ess)
{
    const float nf = (float)m, cf = (float)c2;
    const float b = dsum * (nf / (cf - 1.0f));
    const float w = wsum / cf;
    const float v = ((nf - 1.0f) / nf) * w + b / nf;
    rhat[d] = std::sqrt(w / v); /* sqrt(W / var+): the reference's definition (quirk Q7) */
    ess[d] = stats_ess_one(w, v, acov_sum, c2, m, dim, d, rho);
}

/* ess (stats.rs:496-546) of parameter d from within / var and the lag sums over `c2` chains of length m */
static float stats_ess_one(float w, float v, const float *acov_sum, size_t c2, size_t m, size_t dim, size_t d,
                           std::vector<float> &rho)
{
    const float nf = (float)m, cf = (float)c2;
    for (size_t t = 0; t < m; ++t) {
        const float avg_rho = (acov_sum[t * dim + d] / nf) / cf; /* mean over chains of autocov_c(t) */
        const float diff = -avg_rho + w;
        rho[t] = -(diff / v) + 1.0f;
    }
    float mn = (m >= 2) ? rho[0] + rho[1] : 0.0f;
    float out = 0.0f;
    for (size_t t = 0; t + 1 < m; t += 2) {
        float p_t = rho[t] + rho[t + 1];
        if (p_t <= 0.0f)
            break;
        if (p_t > mn)
            p_t = mn;
        mn = p_t;
        out += p_t;
    }
    const float tau = -1.0f + 2.0f * out;
    return (1.0f / tau) * cf * nf;
}

int mmcmc_stats_finish(const float *means, const float *ssq, const float *acov_sum, size_t c2, size_t m, size_t dim,
                       float *rhat, float *ess)
{
    if (!means || !ssq || !acov_sum || !rhat || !ess || c2 < 2 || m < 1 || dim == 0)
        return MMCMC_ERR_INVALID_ARG;
    const float nf = (float)m, cf = (float)c2;
    std::vector<float> rho(m);
    /* every parameter's sums run over the half-chains in index order, as before; the loops are nested chain-outer so that the
     * [c2, dim] arrays are walked once, contiguously (a device group's finish over 131 072 half-chains: 0.4 -> 0.1 ms) */
    std::vector<float> msum(dim, 0.f), overall(dim), dsum(dim, 0.f), wsum(dim, 0.f);
    for (size_t c = 0; c < c2; ++c)
        for (size_t d = 0; d < dim; ++d)
            msum[d] += means[c * dim + d];
    for (size_t d = 0; d < dim; ++d)
        overall[d] = msum[d] / cf;
    for (size_t c = 0; c < c2; ++c)
        for (size_t d = 0; d < dim; ++d) {
            const float df = means[c * dim + d] - overall[d];
            dsum[d] += df * df;
            wsum[d] += ssq[c * dim + d] / nf; /* biased per-chain variance (quirk Q8) */
        }
    for (size_t d = 0; d < dim; ++d)
        stats_finish_one(dsum[d], wsum[d], acov_sum, c2, m, dim, d, rho, rhat, ess);
    return MMCMC_OK;
}

int mmcmc_split_rhat_mean_ess(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n,
                              size_t dim, float *rhat, float *ess, int device, void *stream_v)
{
    if (!sample || !rhat || !ess || n_chains == 0 || dim == 0)
        return MMCMC_ERR_INVALID_ARG;
    const size_t m = n / 2;
    if (m < 1)
        return MMCMC_ERR_SHAPE;
    int st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    DevGuard g(device);
    hipStream_t stream = (hipStream_t)stream_v;
    const size_t esz = dtype == MMCMC_F32 ? 4 : 8;
    const size_t c2 = 2 * n_chains;
    void *d_sample = nullptr;
    float *d_buf = nullptr;
    /* device layout: means | ssq | acov_sum [m, D] (+ pad to 8 bytes) | cross-chain partial sums [D][CHUNKS][3] f64:
     * the last two blocks are what the host needs */
    constexpr unsigned int kParts = 16; /* partial lag-sum totals, added up on the host */
    const size_t n_acov = (kParts * m * dim + 1) / 2 * 2, n_wb = dim * MM_WB_CHUNKS * 3;
    const size_t nb = 2 * c2 * dim + n_acov + 2 * n_wb;
    int rc = MMCMC_OK;
    hipError_t e = hipSuccess;
    do {
        if (c2 < 2) {
            rc = MMCMC_ERR_INVALID_ARG;
            break;
        }
        float *h = stats_pinned(n_acov + 2 * n_wb);
        if (!h) {
            e = hipErrorOutOfMemory;
            break;
        }
        if (!sample_is_device) {
            if ((e = hipMalloc(&d_sample, n_chains * n * dim * esz)) != hipSuccess)
                break;
            if ((e = hipMemcpyAsync(d_sample, sample, n_chains * n * dim * esz, hipMemcpyHostToDevice, stream)) !=
                hipSuccess)
                break;
        }
        const size_t n_slab_floats = stats_ws_floats(n_chains, n, dim, kParts, device);
        if (!(d_buf = stats_workspace(device, nb + n_slab_floats))) {
            e = hipErrorOutOfMemory;
            break;
        }
        float *d_means = d_buf, *d_ssq = d_buf + c2 * dim;
        /* The few KB the host needs -- the lag sums (one total, or kParts partial totals) and the cross-chain sums -- are
         * WRITTEN BY THE KERNELS straight into the pinned host buffer (device-visible like all hipHostMalloc memory; visible to
         * the host once the event behind the last kernel has completed): the two or three copy commands that used to follow the
         * kernels cost more than the reduction's tail (round 5: 0.150 -> see DESIGN.md 5.2) */
        float *d_acov = h;
        double *d_wb = reinterpret_cast<double *>(h + n_acov);
        unsigned int n_written = kParts;
        rc = stats_partials_impl(sample_is_device ? sample : d_sample, dtype, n_chains, n, dim, d_means, d_ssq, d_acov,
                                 d_wb, d_buf + nb, kParts, device, stream_v, &n_written);
        if (rc != MMCMC_OK)
            break;
        if ((e = stats_wait(device, stream)) != hipSuccess)
            break;
        for (size_t i = 0; i < m * dim; ++i) { /* lag sums: the partial totals in their fixed order */
            float t = h[i];
            for (unsigned int part = 1; part < n_written; ++part)
                t += h[(size_t)part * m * dim + i];
            h[i] = t;
        }
        std::vector<float> rho(m);
        const double *wb = reinterpret_cast<const double *>(h + n_acov);
        for (size_t d = 0; d < dim; ++d) {
            double sd = 0.0, sq = 0.0, ws = 0.0;
            for (size_t ch = 0; ch < MM_WB_CHUNKS; ++ch) {
                const double *q = wb + (d * MM_WB_CHUNKS + ch) * 3;
                sd += q[0];
                sq += q[1];
                ws += q[2];
            }
            const double dsum = sq - sd * sd / (double)c2; /* sum of (mean - overall mean)^2 */
            stats_finish_one((float)(dsum > 0.0 ? dsum : 0.0), (float)ws, h, c2, m, dim, d, rho, rhat, ess);
        }
    } while (0);
    if (d_sample)
        (void)hipFree(d_sample);
    if (e != hipSuccess)
        return (int)e;
    return rc;
}

int mmcmc_stats_finish_sums(const double *dsum, const double *wsum, const float *acov_sum, size_t c2, size_t m, size_t dim,
                            float *rhat, float *ess)
{
    if (!dsum || !wsum || !acov_sum || !rhat || !ess || c2 < 2 || m < 1 || dim == 0)
        return MMCMC_ERR_INVALID_ARG;
    std::vector<float> rho(m);
    for (size_t d = 0; d < dim; ++d)
        stats_finish_one((float)dsum[d], (float)wsum[d], acov_sum, c2, m, dim, d, rho, rhat, ess);
    return MMCMC_OK;
}

/* ess_from_chainstats (stats.rs:668-671): the UN-split ESS of sample [n_chains, n, dim] with within / var taken from the
 * per-chain trackers' ChainStats (withinvar_from_cs, stats.rs:155-178).  The lag sums over the n_chains whole chains come
 * from the same kernels as the split diagnostics: [n_chains, n, dim] read as [n_chains / 2, 2 n, dim] has chain 2k as the
 * first and chain 2k + 1 as the second "half" of row k, each centred on its own mean; an odd last chain is run doubled
 * and counted half. */
extern "C" int mmcmc_tracker_within_var(struct mmcmc_tracker *h, float *within, float *var, void *stream);
extern "C" int mmcmc_tracker_shape(struct mmcmc_tracker *h, size_t *n_chains, size_t *dim, int *device);
int mmcmc_ess_from_chainstats(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n, size_t dim,
                              struct mmcmc_tracker *tracker, float *ess, int device, void *stream_v)
{
    if (!sample || !tracker || !ess || n_chains == 0 || dim == 0 || n < 1 || (dtype != MMCMC_F32 && dtype != MMCMC_F64))
        return MMCMC_ERR_INVALID_ARG;
    size_t tc = 0, td = 0;
    int tdev = 0;
    int st = mmcmc_tracker_shape(tracker, &tc, &td, &tdev);
    if (st != MMCMC_OK)
        return st;
    if (tc != n_chains || td != dim)
        return MMCMC_ERR_SHAPE;
    st = check_device(device);
    if (st != MMCMC_OK)
        return st;
    std::vector<float> within(dim), var(dim);
    st = mmcmc_tracker_within_var(tracker, within.data(), var.data(), tdev == device ? stream_v : nullptr);
    if (st != MMCMC_OK)
        return st;
    DevGuard g(device);
    hipStream_t stream = (hipStream_t)stream_v;
    const size_t esz = dtype == MMCMC_F32 ? 4 : 8, row = n * dim * esz;
    const size_t pairs = n_chains / 2;
    void *d_sample = nullptr, *d_tail = nullptr;
    float *d_ws = nullptr;
    int rc = MMCMC_OK;
    hipError_t e = hipSuccess;
    std::vector<float> acov(n * dim, 0.f), tmp(n * dim);
    do {
        const char *src = (const char *)sample;
        if (!sample_is_device) {
            if ((e = hipMalloc(&d_sample, n_chains * row)) != hipSuccess)
                break;
            if ((e = hipMemcpyAsync(d_sample, sample, n_chains * row, hipMemcpyHostToDevice, stream)) != hipSuccess)
                break;
            src = (const char *)d_sample;
        }
        /* means | ssq of the pseudo half-chains (unused) | lag sums */
        const size_t n_ms = 2 * std::max<size_t>(pairs, 1) * dim;
        if ((e = hipMalloc((void **)&d_ws, (2 * n_ms + n * dim) * sizeof(float))) != hipSuccess)
            break;
        float *d_means = d_ws, *d_ssq = d_ws + n_ms, *d_acov = d_ws + 2 * n_ms;
        if (pairs) {
            rc = mmcmc_stats_partials(src, dtype, pairs, 2 * n, dim, d_means, d_ssq, d_acov, device, stream_v);
            if (rc != MMCMC_OK)
                break;
            if ((e = hipMemcpyAsync(acov.data(), d_acov, n * dim * sizeof(float), hipMemcpyDeviceToHost, stream)) != hipSuccess)
                break;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess)
                break;
        }
        if (n_chains & 1) {
            if ((e = hipMalloc(&d_tail, 2 * row)) != hipSuccess)
                break;
            const char *last = src + (n_chains - 1) * row;
            if ((e = hipMemcpyAsync(d_tail, last, row, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
                break;
            if ((e = hipMemcpyAsync((char *)d_tail + row, last, row, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
                break;
            rc = mmcmc_stats_partials(d_tail, dtype, 1, 2 * n, dim, d_means, d_ssq, d_acov, device, stream_v);
            if (rc != MMCMC_OK)
                break;
            if ((e = hipMemcpyAsync(tmp.data(), d_acov, n * dim * sizeof(float), hipMemcpyDeviceToHost, stream)) != hipSuccess)
                break;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess)
                break;
            for (size_t i = 0; i < n * dim; ++i)
                acov[i] += 0.5f * tmp[i];
        }
        std::vector<float> rho(n);
        for (size_t d = 0; d < dim; ++d)
            ess[d] = stats_ess_one(within[d], var[d], acov.data(), n_chains, n, dim, d, rho);
    } while (0);
    (void)hipStreamSynchronize(stream);
    if (d_sample)
        (void)hipFree(d_sample);
    if (d_tail)
        (void)hipFree(d_tail);
    if (d_ws)
        (void)hipFree(d_ws);
    if (e != hipSuccess)
        return (int)e;
    return rc;
}

/* stats.rs:310-336 basic_stats: sorted descending; min = last, median = [len/2], max = first, std with ddof 1 */
int mmcmc_basic_stats_from(const float *data, size_t len, mmcmc_basic_stats *out)
{
    if (!data || !out || len == 0)
        return MMCMC_ERR_INVALID_ARG;
    std::vector<float> s(data, data + len);
    std::stable_sort(s.begin(), s.end(), [](float a, float b) { return b < a; });
    out->min = s[len - 1];
    out->median = s[len / 2];
    out->max = s[0];
    float sum = 0.f;
    for (float v : s)
        sum += v;
    out->mean = sum / (float)len;
    float mu = 0.f, sq = 0.f;
    for (size_t i = 0; i < len; ++i) {
        const float delta = s[i] - mu;
        mu += delta / (float)(i + 1);
        sq = (s[i] - mu) * delta + sq;
    }
    out->std = std::sqrt(sq / ((float)len - 1.0f));
    return MMCMC_OK;
}

/* RunStats::from (stats.rs:360-371) */
int mmcmc_run_stats_from(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n, size_t dim,
                    mmcmc_run_stats *out, int device, void *stream)
{
    if (!out)
        return MMCMC_ERR_INVALID_ARG;
    std::vector<float> rhat(dim), ess(dim);
    int rc = mmcmc_split_rhat_mean_ess(sample, sample_is_device, dtype, n_chains, n, dim, rhat.data(), ess.data(),
                                       device, stream);
    if (rc != MMCMC_OK)
        return rc;
    rc = mmcmc_basic_stats_from(ess.data(), dim, &out->ess);
    if (rc != MMCMC_OK)
        return rc;
    return mmcmc_basic_stats_from(rhat.data(), dim, &out->rhat);
}

} /* extern "C" */
