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
 *   kernel 2 (mm_slab_reduce_kernel): sums the per-wave slabs in a fixed order (bitwise reproducible, no atomics).
 *   host finish (mm_stats_finish): W, B, var+, the reference's "R-hat" sqrt(W/var+) (quirk Q7), rho_t, Geyer's
 *      initial monotone sequence and ESS exactly as stats.rs:449-465, :509-545 order them, in f32.
 *
 * The autocovariance is evaluated directly (the reference does the same for m <= 100 and switches to an FFT above;
 * both compute the same biased, mean-removed estimator).  Only sufficient statistics leave the GPU: 2*C*D means
 * and sums of squares and m*D lag sums -- which is also all a multi-GPU run has to exchange (all-gather of the
 * former, all-reduce of the latter; mini_mcmc_amd/stats.py).
 */
#include "../../include/mmcmc.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
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

/* The same half-chain statistics with the lag sums on the matrix cores.  For one parameter of one half-chain the
 * biased autocovariance sums c_k = sum_t y[t] y[t+k] are a product of two Hankel/Toeplitz operands:
 *     A[i][t] = y[t + i]  (16 x K),   B[t][j] = y[t - 16 j - 256 tau]  (K x 16)   =>   (A B)[i][j] = c_(i + 16 j + 256 tau)
 * (y = 0 outside [0, m)), so one 16 x 16 accumulator tile of v_mfma_f32_16x16x4_f32 holds 256 consecutive lags and
 * K = m suffices: m / 4 MFMAs per tile instead of ~2 m LDS reads and m FMAs per lane -- the direct kernel above is
 * LDS-issue bound (two reads per FMA).  y sits in LDS with 240 + 256 (tiles - 1) zeros in front and 20 behind, and
 * with one pad word after every 16 (address p + p / 16): the B operand's lanes read addresses 16 apart, which would
 * otherwise land on two banks.  Means, centred sums of squares and the carried per-wave slab are as above. */
typedef float mm_f4 __attribute__((ext_vector_type(4)));

template <class T>
__global__ __launch_bounds__(64) void mm_half_chain_mfma_kernel(const T *__restrict__ sample, unsigned long long C,
                                                                unsigned int n, unsigned int D, unsigned int m,
                                                                unsigned int n_tiles, float *__restrict__ means,
                                                                float *__restrict__ ssq, float *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned int padl = 240u + 256u * (n_tiles - 1u);
    const unsigned int row_len = padl + m + 20u;            /* logical elements per parameter */
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

    const unsigned int k_steps = (m + 3u) / 4u;
    for (unsigned long long hc = blockIdx.x; hc < n_half; hc += gridDim.x) {
        const unsigned long long chain = hc < C ? hc : hc - C;
        const unsigned int row0 = hc < C ? 0u : n - m;
        const T *src = sample + (chain * n + row0) * D;
        {
            /* (t, d) of element e = lane + 64 i without a division per element: advance by 64 = q64 * D + r64 */
            const unsigned int q64 = 64u / D, r64 = 64u - q64 * D;
            unsigned int t = lane / D, d = lane - t * D;
            for (unsigned int e = lane; e < m * D; e += 64) {
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
            float s = 0.f;
            for (unsigned int t = lane; t < m; t += 64)
                s += yd[at(padl + t)];
            const float mean = wave_sum(s) / (float)m;
            float q = 0.f;
            for (unsigned int t = lane; t < m; t += 64) {
                const float v = yd[at(padl + t)] - mean;
                yd[at(padl + t)] = v;
                q += v * v;
            }
            q = wave_sum(q);
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
                unsigned int aa[4], ba[4];
#pragma unroll
                for (unsigned int u = 0; u < 4; ++u) {
                    aa[u] = at(padl + kq + li + 4u * u);
                    ba[u] = at(padl + kq - 16u * li - 256u * tau + 4u * u);
                }
                unsigned int sidx = 0;
                for (; sidx + 4 <= k_steps; sidx += 4) {
#pragma unroll
                    for (unsigned int u = 0; u < 4; ++u) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yd[aa[u]], yd[ba[u]], c0, 0, 0, 0);
                        aa[u] += 17u;
                        ba[u] += 17u;
                    }
                }
                for (unsigned int u = 0; sidx < k_steps; ++sidx, ++u)
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yd[aa[u]], yd[ba[u]], c0, 0, 0, 0);
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

/* out[lag, d] = sum over waves of slabs[w, d, lag].  One wave per group of 64 outputs would serialise n_slabs loads;
 * instead a 256-thread block owns 64 consecutive outputs, its four waves each sum a quarter of the slabs (coalesced
 * 256-byte rows), and the four partial sums are combined in a fixed order -- bitwise reproducible, no atomics. */
__global__ __launch_bounds__(256) void mm_slab_reduce_kernel(const float *__restrict__ slabs, unsigned int n_slabs,
                                                             unsigned int D, unsigned int m, float *__restrict__ out)
{
    __shared__ float part[4][64];
    const unsigned int lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const unsigned int i = blockIdx.x * 64u + lane; /* over d * m + lag */
    const unsigned int total = D * m;
    float s = 0.f;
    if (i < total) {
        const unsigned int per = (n_slabs + 3u) / 4u;
        const unsigned int lo = w * per, hi = min(n_slabs, lo + per);
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
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < total) {
        const float r = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        const unsigned int d = i / m, lag = i - d * m;
        out[(size_t)lag * D + d] = r;
    }
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

/* withinvar's cross-chain sums (stats.rs:449-465) on the device: per parameter d the sum of squared deviations of the
 * half-chain means from their mean, and the sum of the biased half-chain variances.  One block per parameter, f64
 * accumulators; out[d] = {dsum, wsum}.  Spares the single-GPU path the copy of 2 * 2C * D numbers to the host. */
__global__ __launch_bounds__(1024) void mm_within_between_kernel(const float *__restrict__ means,
                                                                 const float *__restrict__ ssq, unsigned long long c2,
                                                                 unsigned int D, float nf, float *__restrict__ out)
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
    double s = 0.0;
    for (unsigned long long c = tid; c < c2; c += 1024)
        s += (double)means[c * D + d];
    const float overall = (float)(block_sum(s) / (double)c2);
    double ds = 0.0, ws = 0.0;
    for (unsigned long long c = tid; c < c2; c += 1024) {
        const float df = means[c * D + d] - overall;
        ds += (double)(df * df);
        ws += (double)(ssq[c * D + d] / nf); /* biased per-chain variance (quirk Q8) */
    }
    ds = block_sum(ds);
    ws = block_sum(ws);
    if (tid == 0) {
        out[2 * d] = (float)ds;
        out[2 * d + 1] = (float)ws;
    }
}

} // namespace

/* number of waves (= per-wave lag-sum slabs) the half-chain kernel is launched with */
static unsigned int stats_n_slabs(size_t n_chains)
{
    /* 4 waves per SIMD hide the LDS / global latencies of the half-chain loop; more only lengthens the slab reduction
     * (measured at [65536, 400, 3]: 1024 waves 1.65 ms, 2048 0.96, 4096 0.69 + 0.08, 8192 0.82 + 0.21) */
    const char *e = getenv("MMCMC_STATS_WAVES");
    const size_t w = e ? (size_t)atoi(e) : 4096;
    return (unsigned int)std::min<size_t>(2 * n_chains, w ? w : 4096);
}

extern "C" {

int mmcmc_stats_partials(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, float *means,
                         float *ssq, float *acov_sum, int device, void *stream_v)
{
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
    const unsigned int n_slabs = stats_n_slabs(n_chains);
    float *slabs = nullptr;
    /* lag sums on the matrix cores (one 16 x 16 tile per 256 lags) unless the half-chain is too long for LDS */
    const unsigned int n_tiles = (unsigned int)((m + 255) / 256);
    const size_t row_len = 240 + 256 * (size_t)(n_tiles - 1) + m + 20, row_pitch = row_len + row_len / 16 + 1;
    const size_t lds_mfma = ((size_t)dim * row_pitch + (size_t)dim * m) * sizeof(float);
    const char *force_direct = getenv("MMCMC_STATS_DIRECT"); /* measurement aid: the direct kernel */
    if (lds_mfma <= 64 * 1024 && !(force_direct && force_direct[0] == '1')) {
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
        if (lds > 160 * 1024)
            return MMCMC_ERR_UNSUPPORTED;
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
    MM_HIP(hipGetLastError());
    const unsigned int total = (unsigned int)(dim * m);
    hipLaunchKernelGGL(mm_slab_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, slabs, n_slabs,
                       (unsigned int)dim, (unsigned int)m, acov_sum);
    MM_HIP(hipGetLastError());
    MM_HIP(hipFreeAsync(slabs, stream));
    return MMCMC_OK;
}

/* stats.rs:449-465 (withinvar), :425-427 (rhat), :509-545 (ess) on the gathered sufficient statistics, f32.
 * means, ssq: [c2, dim] in splitcat order (first halves of all chains, then second halves); acov_sum [m, dim]. */
/* from the cross-chain sums to R-hat and ESS of parameter d (stats.rs:459-465, 425-427, 509-545), f32 */
static void stats_finish_one(float dsum, float wsum, const float *acov_sum, size_t c2, size_t m, size_t dim, size_t d,
                             std::vector<float> &rho, float *rhat, float *ess)
{
    const float nf = (float)m, cf = (float)c2;
    const float b = dsum * (nf / (cf - 1.0f));
    const float w = wsum / cf;
    const float v = ((nf - 1.0f) / nf) * w + b / nf;
    rhat[d] = std::sqrt(w / v); /* sqrt(W / var+): the reference's definition (quirk Q7) */
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
    ess[d] = (1.0f / tau) * cf * nf;
}

int mmcmc_stats_finish(const float *means, const float *ssq, const float *acov_sum, size_t c2, size_t m, size_t dim,
                       float *rhat, float *ess)
{
    if (!means || !ssq || !acov_sum || !rhat || !ess || c2 < 2 || m < 1 || dim == 0)
        return MMCMC_ERR_INVALID_ARG;
    const float nf = (float)m, cf = (float)c2;
    std::vector<float> rho(m);
    for (size_t d = 0; d < dim; ++d) {
        float msum = 0.f;
        for (size_t c = 0; c < c2; ++c)
            msum += means[c * dim + d];
        const float overall = msum / cf;
        float dsum = 0.f, wsum = 0.f;
        for (size_t c = 0; c < c2; ++c) {
            const float df = means[c * dim + d] - overall;
            dsum += df * df;
            wsum += ssq[c * dim + d] / nf; /* biased per-chain variance (quirk Q8) */
        }
        stats_finish_one(dsum, wsum, acov_sum, c2, m, dim, d, rho, rhat, ess);
    }
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
    const size_t nb = (2 * c2 * dim + m * dim + 2 * dim);
    std::vector<float> h(m * dim + 2 * dim);
    int rc = MMCMC_OK;
    hipError_t e = hipSuccess;
    do {
        if (!sample_is_device) {
            if ((e = hipMalloc(&d_sample, n_chains * n * dim * esz)) != hipSuccess)
                break;
            if ((e = hipMemcpyAsync(d_sample, sample, n_chains * n * dim * esz, hipMemcpyHostToDevice, stream)) !=
                hipSuccess)
                break;
        }
        if ((e = hipMallocAsync((void **)&d_buf, nb * sizeof(float), stream)) != hipSuccess)
            break;
        /* layout: means | ssq | acov_sum [m, D] | {dsum, wsum} [D]: the last two blocks are what the host needs */
        float *d_means = d_buf, *d_ssq = d_buf + c2 * dim, *d_acov = d_buf + 2 * c2 * dim, *d_wb = d_acov + m * dim;
        rc = mmcmc_stats_partials(sample_is_device ? sample : d_sample, dtype, n_chains, n, dim, d_means, d_ssq,
                                  d_acov, device, stream_v);
        if (rc != MMCMC_OK)
            break;
        hipLaunchKernelGGL(mm_within_between_kernel, dim3((unsigned int)dim), dim3(1024), 0, stream, d_means, d_ssq,
                           (unsigned long long)c2, (unsigned int)dim, (float)m, d_wb);
        if ((e = hipGetLastError()) != hipSuccess)
            break;
        if ((e = hipMemcpyAsync(h.data(), d_acov, h.size() * sizeof(float), hipMemcpyDeviceToHost, stream)) != hipSuccess)
            break;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess)
            break;
        if (c2 < 2) {
            rc = MMCMC_ERR_INVALID_ARG;
            break;
        }
        std::vector<float> rho(m);
        const float *wb = h.data() + m * dim;
        for (size_t d = 0; d < dim; ++d)
            stats_finish_one(wb[2 * d], wb[2 * d + 1], h.data(), c2, m, dim, d, rho, rhat, ess);
    } while (0);
    if (d_sample)
        (void)hipFree(d_sample);
    if (d_buf)
        (void)hipFreeAsync(d_buf, stream);
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
