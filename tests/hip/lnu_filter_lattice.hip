// Checker for mm_ratio_exceeds_lnu_f32 (csrc/mm_rng.h), compiled and run by tests/test_gpu_parity.py on the GPU box:
// over ALL 2^24 accept uniforms u = (s + 1) 2^-24 of the f32 stream
//   (a) the table-free logarithm (v_log_f32 * ln 2) deviates from the table's ln u by at most a fraction of the band
//       MM_LNU_BAND (1 + |ln u|) -- printed as max_dev_over_band (must be < 0.5: the filter's margin);
//   (b) the filtered decision equals the plain comparison `ratio > mm_lnu_f32(u)` for ratios at, next to and around the
//       table's value: ln u itself, its f32 neighbours, +- half a band, +- one band, +- two bands, 0, -inf, +inf, NaN.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>

#include "mm_rng.h"

__global__ void lattice(unsigned long long *mismatch, unsigned int *max_dev_bits, unsigned long long *in_band)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (1u << 24))
        return;
    const float u = fmaf((float)s, 0x1.0p-24f, 0x1.0p-24f);
    const mm_icdf_global tab;
    const float L = mm_lnu_f32(u, tab);
    const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
    const float band = fmaf(fabsf(l), MM_LNU_BAND, MM_LNU_BAND);
    const float dev = fabsf(l - L) / band;
    atomicMax(max_dev_bits, __float_as_uint(dev));
    const float probes[] = {L, nextafterf(L, INFINITY), nextafterf(L, -INFINITY), L + 0.5f * band, L - 0.5f * band, L + band, L - band,
                            L + 2.0f * band, L - 2.0f * band, l, nextafterf(l, INFINITY), nextafterf(l, -INFINITY), 0.0f, -INFINITY, INFINITY,
                            __builtin_nanf("")};
    unsigned long long bad = 0, nb = 0;
    for (float r : probes) {
        bad += mm_ratio_exceeds_lnu_f32(r, u, tab) != (r > L);
        nb += fabsf(r - l) <= band;
    }
    if (bad)
        atomicAdd(mismatch, bad);
    atomicAdd(in_band, nb);
}

// (c) the paired MH stream's filter (mm_ratio_exceeds_lnu_mhp): over all 2^16 values of the uniform's high bits s16 and all
//     2^8 low bytes s8 -- i.e. the same 2^24 uniforms u = (s16 256 + s8 + 1) 2^-24 -- the decision taken from
//     u_hi = (s16 + 1) 2^-16 with the low byte supplied only on demand equals `ratio > mm_lnu_f32(u)` for probe ratios at the
//     logarithms of u, of the interval's two ends, their neighbours and +- bands; `demanded` counts how often the low
//     byte was asked for.
__global__ void lattice_mhp(unsigned long long *mismatch, unsigned long long *demanded)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (1u << 24))
        return;
    const uint32_t s16 = s >> 8, s8 = s & 255u;
    const mm_icdf_global tab;
    const float u = mm_mhp_u(s16, s8), u_hi = mm_mhp_u_hi(s16), u_lo = mm_mhp_u(s16, 0u);
    const float L = mm_lnu_f32(u, tab), L_hi = mm_lnu_f32(mm_mhp_u(s16, 255u), tab), L_lo = mm_lnu_f32(u_lo, tab);
    const float l_hi = __builtin_amdgcn_logf(u_hi) * 0.6931471805599453f, l_lo = __builtin_amdgcn_logf(u_lo) * 0.6931471805599453f;
    const float b_hi = fmaf(fabsf(l_hi), MM_LNU_BAND, MM_LNU_BAND), b_lo = fmaf(fabsf(l_lo), MM_LNU_BAND, MM_LNU_BAND);
    const float probes[] = {L, nextafterf(L, INFINITY), nextafterf(L, -INFINITY), L_hi, nextafterf(L_hi, INFINITY), L_lo, nextafterf(L_lo, -INFINITY),
                            l_hi + b_hi, nextafterf(l_hi + b_hi, INFINITY), l_lo - b_lo, nextafterf(l_lo - b_lo, -INFINITY), l_hi, l_lo,
                            0.5f * (L_lo + L_hi), l_hi + 2.0f * b_hi, l_lo - 2.0f * b_lo, 0.0f, -INFINITY, INFINITY, __builtin_nanf("")};
    unsigned long long bad = 0, dem = 0;
    for (float r : probes) {
        bool asked = false;
        const bool got = mm_ratio_exceeds_lnu_mhp(r, u_hi, [&]() { asked = true; return s8; }, tab);
        bad += got != (r > L);
        dem += asked;
    }
    if (bad)
        atomicAdd(mismatch, bad);
    atomicAdd(demanded, dem);
}

int main()
{
    unsigned long long *d_mis, *d_inb, h_mis = 0, h_inb = 0;
    unsigned int *d_dev, h_dev = 0;
    if (hipMalloc(&d_mis, 8) != hipSuccess || hipMalloc(&d_inb, 8) != hipSuccess || hipMalloc(&d_dev, 4) != hipSuccess)
        return 2;
    hipMemset(d_mis, 0, 8);
    hipMemset(d_inb, 0, 8);
    hipMemset(d_dev, 0, 4);
    hipLaunchKernelGGL(lattice, dim3((1u << 24) / 256), dim3(256), 0, 0, d_mis, d_dev, d_inb);
    if (hipDeviceSynchronize() != hipSuccess)
        return 3;
    hipMemcpy(&h_mis, d_mis, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&h_inb, d_inb, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&h_dev, d_dev, 4, hipMemcpyDeviceToHost);
    float dev;
    memcpy(&dev, &h_dev, 4);
    unsigned long long *d_mis2, *d_dem, h_mis2 = 0, h_dem = 0;
    if (hipMalloc(&d_mis2, 8) != hipSuccess || hipMalloc(&d_dem, 8) != hipSuccess)
        return 2;
    hipMemset(d_mis2, 0, 8);
    hipMemset(d_dem, 0, 8);
    hipLaunchKernelGGL(lattice_mhp, dim3((1u << 24) / 256), dim3(256), 0, 0, d_mis2, d_dem);
    if (hipDeviceSynchronize() != hipSuccess)
        return 3;
    hipMemcpy(&h_mis2, d_mis2, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&h_dem, d_dem, 8, hipMemcpyDeviceToHost);
    printf("{\"uniforms\": %u, \"probes_per_uniform\": 16, \"mismatches\": %llu, \"max_dev_over_band\": %.6f, \"probes_in_band\": %llu, "
           "\"paired_probes_per_uniform\": 20, \"paired_mismatches\": %llu, \"paired_low_byte_demanded\": %llu}\n", 1u << 24,
           h_mis, dev, h_inb, h_mis2, h_dem);
    return (h_mis == 0 && h_mis2 == 0) ? 0 : 1;
}
