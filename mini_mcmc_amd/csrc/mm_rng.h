/*
 * mm_rng.h -- the engine's counter-based random stream (host + device, one definition).
 *
 * Replaces, for the GPU path, the per-chain `SmallRng` + ziggurat stack of the reference
 * (metropolis_hastings.rs:109,190,310; distributions.rs:347,364-372; nuts.rs:388,554-563,579,660,910;
 * hmc.rs:309-313,359-363 use burn's global RNG).  A lane cannot carry the reference's sequential
 * generator state across launches or GPUs without serialising, so every draw is a pure function
 *
 *        philox4x32-10( key = seed, counter = (chain, iteration, block) )
 *
 * which makes results independent of launch partitioning, iterations-per-launch and GPU count.
 *
 * Draw schedule (normative; oracle/orng.c restates it independently in plain C):
 *   counter = { chain & 0xffffffff, chain >> 32, iteration, block },  key = { seed & 0xffffffff, seed >> 32 }
 *   u53(hi, lo)  = (double)((((u64)hi << 21) | (lo >> 11)) + 1) * 2^-53   in (0,1]
 *   f32 block b: words w0..w3
 *       normals z[4b+i] = icdf24(w_i), i = 0..3: one normal per word by inversion of the normal CDF on a 2^24-point
 *                                 lattice, symmetric about 0:  n = (w >> 8) | 1 (odd, 24 bits), sign = bit 8 of w,
 *                                 z = sign * -Phi^-1(n 2^-25), with -Phi^-1 the piecewise cubic of mm_icdf_table.h
 *                                 (16 segments per binade of n, |error| <= 1.2e-7 max(1, |z|); |z| <= 5.42)
 *       spare uniform           = u24-style from the low bytes: s = (w0&255) | (w1&255)<<8 | (w2&255)<<16,
 *                                 (float)(s + 1) * 2^-24;  its logarithm (the accept rules compare with ln u) is
 *                                 mm_lnu_f32: exponent * ln 2 + a cubic in the mantissa on 32 segments
 *   f64 block b: z[2b+0], z[2b+1] = box_muller(u53(w0,w1), u53(w2,w3))
 *   box_muller(u1, u2) = ( r cos(2 pi u2), r sin(2 pi u2) ),  r = sqrt(-2 ln u1)
 *   (f32 normals were Box-Muller pairs too until the inversion table replaced them: ~13 instructions and one 16-byte
 *    table read per normal instead of ~30 instructions of software log / sqrt / sincos -- the sampling kernels are
 *    instruction-issue bound and the noise was 60 % of the HMC kernel, tools/hmc_split.hip.)
 *   MH / HMC iteration t of a chain:
 *       noise z[0..D) from blocks 0..;   accept uniform: f32 -> spare of block 0;  f64 -> u53(w0,w1) of block AUX
 *   NUTS iteration: momentum z[0..D) as above; auxiliary draw k (k = 0: Exp(1) = -ln u, then the uniforms in
 *       program order) = u53 of words (2(k&1), 2(k&1)+1) of block AUX + (k >> 1)  -- always 53-bit.
 */
#ifndef MM_RNG_H
#define MM_RNG_H

#include "mm_math.h"
#include "mm_icdf_table.h"

#define MM_AUX_BLOCK 0x40000000u

/* ---- the f32 normal: table-driven inverse CDF ----
 * The table lives in three places with identical contents: a host array, a device array in global memory (default
 * on the device: 16-byte gathers that hit L1) and, for the kernels that draw noise every iteration, a copy in LDS
 * (mm_icdf_lds).  `Tab` only says where row r is read from; the arithmetic is the same everywhere. */
typedef float mm_v4f __attribute__((vector_size(16)));
static const float mm_icdf_tab_h[MM_ICDF_ROWS * 4] __attribute__((aligned(16))) = MM_ICDF_COEFFS;
static const float mm_lnm_tab_h[MM_LNM_ROWS * 4] __attribute__((aligned(16))) = MM_LNM_COEFFS;
#if defined(__HIPCC__)
static __device__ const float mm_icdf_tab_d[MM_ICDF_ROWS * 4] __attribute__((aligned(16))) = MM_ICDF_COEFFS;
static __device__ const float mm_lnm_tab_d[MM_LNM_ROWS * 4] __attribute__((aligned(16))) = MM_LNM_COEFFS;
#endif

struct mm_icdf_global {
    MM_HD void row(uint32_t r, float *c) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const mm_v4f v = *reinterpret_cast<const mm_v4f *>(mm_icdf_tab_d + 4 * r);
#else
        const mm_v4f v = *reinterpret_cast<const mm_v4f *>(mm_icdf_tab_h + 4 * r);
#endif
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
    MM_HD void row_of_bits(uint32_t b, float *c) const { row((b >> 19) & 511u, c); }
    MM_HD void row_ln(uint32_t r, float *c) const /* the logarithm's table (mm_lnu_f32) */
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const mm_v4f v = *reinterpret_cast<const mm_v4f *>(mm_lnm_tab_d + 4 * r);
#else
        const mm_v4f v = *reinterpret_cast<const mm_v4f *>(mm_lnm_tab_h + 4 * r);
#endif
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
};

#if defined(__HIPCC__)
typedef __attribute__((address_space(3))) const mm_v4f *mm_lds_f4_ptr;
/* The sampling kernels keep the tables at the very start of their (dynamic) LDS and have no static LDS, i.e. at LDS
 * address 0: mm_icdf_lds_at0() hands out that address as a literal, so a row's address is its byte offset and nothing
 * else.  With the array's symbol as the base every table read carried a `v_add_u32 v, <base>, v` (the base is resolved
 * after instruction selection and cannot be folded): one instruction per normal and per logarithm.  The kernels check
 * the assumption once (mm_icdf_lds_check) and trap if it does not hold. */
struct mm_icdf_lds {
    mm_lds_f4_ptr tab; /* MM_ICDF_ROWS rows of the normal's table, then MM_LNM_ROWS of the logarithm's (mm_icdf_lds_fill) */
    __device__ __forceinline__ void row(uint32_t r, float *c) const
    {
        const mm_v4f v = tab[r];
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
    __device__ __forceinline__ void row_of_bits(uint32_t b, float *c) const
    {
        /* the row's byte offset straight from the bits: (b >> 15) & 0x1ff0 (tab is LDS address 0, mm_icdf_lds_at0) */
        const unsigned int off = (b >> 15) & 0x1ff0u;
        const mm_v4f v = *(mm_lds_f4_ptr)((unsigned int)(unsigned long long)tab + off);
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
    __device__ __forceinline__ void row_ln(uint32_t r, float *c) const
    {
        const mm_v4f v = tab[MM_ICDF_ROWS + r];
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];
    }
};
__device__ __forceinline__ mm_lds_f4_ptr mm_icdf_lds_at0() { return (mm_lds_f4_ptr)(unsigned int)0u; }
__device__ __forceinline__ void mm_icdf_lds_check(const void *lds_tab)
{
    if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) const void *)lds_tab != 0u)
        __builtin_trap();
}
/* every thread of the block copies its share; the caller synchronises the block before the first draw */
__device__ __forceinline__ void mm_icdf_lds_fill(float *lds_tab, unsigned int tid, unsigned int nthreads)
{
    for (unsigned int i = tid; i < MM_ICDF_ROWS; i += nthreads)
        reinterpret_cast<mm_v4f *>(lds_tab)[i] = *reinterpret_cast<const mm_v4f *>(mm_icdf_tab_d + 4 * i);
    for (unsigned int i = tid; i < MM_LNM_ROWS; i += nthreads)
        reinterpret_cast<mm_v4f *>(lds_tab)[MM_ICDF_ROWS + i] = *reinterpret_cast<const mm_v4f *>(mm_lnm_tab_d + 4 * i);
}
#endif
#define MM_NOISE_TABLE_BYTES ((MM_ICDF_ROWS + MM_LNM_ROWS) * 16)

/* an opaque use-and-redefine of a value on the device: keeps the SLP vectoriser from fusing the scalar operations that
 * produced it with a neighbour's (no instruction is emitted) */
/* Only in translation units that ask for it (MM_NOISE_SCALAR_CUBICS: the MH / HMC sampling kernels, mm_inst.inc): in the
 * NUTS kernels the same constraint sends this compiler into "Illegal instruction detected: Subtarget requires even
 * aligned vector registers". */
#if defined(__HIP_DEVICE_COMPILE__) && defined(MM_NOISE_SCALAR_CUBICS)
#define MM_KEEP_SCALAR(v) asm("" : "+v"(v)) /* not volatile: free to move, so neighbouring table reads still issue together */
#else
#define MM_KEEP_SCALAR(v) ((void)0)
#endif

template <class Tab> MM_HD float mm_icdf_f32(uint32_t w, const Tab &tab)
{
    const uint32_t n = (w >> 8) | 1u;           /* odd, < 2^24: exact in f32 */
    const uint32_t b = mm_f2u((float)n);
    float c[4];
    tab.row_of_bits(b, c);                      /* row (b >> 19) & 511: low 5 exponent bits, top 4 mantissa bits */
    const float t = mm_u2f((b & 0x7ffffu) | 0x3f800000u) - 1.0f; /* [0, 1/16), exact */
    const float m = fmaf(fmaf(fmaf(c[3], t, c[2]), t, c[1]), t, c[0]);
    /* magnitude of m, sign = bit 8 of w: (m & 0x7fffffff) | ((w << 23) & 0x80000000) written as a copysign -- one v_bfi_b32 */
    float z = __builtin_copysignf(m, mm_u2f(w << 23));
    /* On the device the result is made opaque (no instruction): otherwise the vectoriser evaluates the cubics of two
     * neighbouring normals as three packed fmas and pays six register moves to pair the coefficients of the two table
     * rows -- nine instructions and twelve issue slots where six scalar fmas take six. */
    MM_KEEP_SCALAR(z);
    return z;
}

/* ln u of the f32 accept uniform (u = (s + 1) 2^-24, any positive normal f32 works): e ln 2 + ln m with the mantissa's
 * logarithm a cubic on 32 segments (mm_icdf_table.h); |error| <= 1e-7 max(1, |ln u|) -- far inside the accept rule's own
 * f32 rounding -- in ~14 instructions and one table read where the polynomial mm_logf takes ~25 */
template <class Tab> MM_HD float mm_lnu_f32(float u, const Tab &tab)
{
    const uint32_t b = mm_f2u(u);
    const float e = (float)((int32_t)(b >> 23) - 127);
    float c[4];
    tab.row_ln((b >> 18) & 31u, c); /* top five mantissa bits */
    const float t = mm_u2f((b & 0x3ffffu) | 0x3f800000u) - 1.0f; /* [0, 1/32), exact */
    const float lnm = fmaf(fmaf(fmaf(c[3], t, c[2]), t, c[1]), t, c[0]);
    float r = fmaf(e, 0.6931471805599453f, lnm);
    MM_KEEP_SCALAR(r); /* as in mm_icdf_f32: two logarithms side by side are not worth pairing */
    return r;
}
/* ---- the Metropolis-Hastings sampler's f32 stream at D <= 2: TWO transitions per Philox block (round 5) -------------
 * A transition of the random walk at D <= 2 needs two normals (24 bits each) and a 24-bit accept uniform: 72 of a block's
 * 128 bits -- and the ten rounds were half of the MH step's issue slots (profiles/r5e_split_mh_stream_probe.log: one block per
 * two transitions, timing only: -14 %).  So, for sampler = MH, T = f32, D <= 2 (mm_mh_paired) and ONLY there:
 *     k = iteration >> 1, h = iteration & 1
 *     W = philox4x32-10(key = seed, counter = (chain, k, MM_MHP_BLOCK))            one block for iterations 2k and 2k + 1
 *     z_i = icdf24(W[2h + i]), i < D                                                 (as everywhere: mm_icdf_f32)
 *     s16 = low byte of W[2h] | low byte of W[2h + 1] << 8                           the HIGH 16 bits of the accept uniform
 *     s8  = byte h of word 0 of philox(key = seed, counter = (chain, k, MM_MHP_BLOCK | 1))   its LOW 8 bits
 *     u   = (s16 * 256 + s8 + 1) 2^-24                                               uniform on the same 2^24 lattice as before
 * u is a pure function of (seed, chain, iteration) like every other draw, so results stay independent of launch partition,
 * continuation and sharding.  Its low byte comes from a SECOND block on purpose: `ratio > ln u` is decided by the high 16
 * bits alone unless ratio falls between ln(u_lo) and ln(u_hi) (u_lo / u_hi = the smallest / largest u with that s16) --
 * about 3 wave-steps in 1000 -- so the sampling kernel evaluates the second block only then (mm_accept_mhp_f32 below); the
 * host build, the other kernels and the oracle evaluate it always.  Same decisions either way.  HMC, NUTS, f64 and D > 2
 * keep the stream described at the top of this file. */
#define MM_MHP_BLOCK 0x20000000u
template <class T, int D, int SAMPLER> struct mm_mh_paired {
    static constexpr bool value = SAMPLER == 0 /* MM_SAMPLER_MH */ && sizeof(T) == 4 && D <= 2;
};
MM_HD uint32_t mm_mhp_s16(uint32_t wa, uint32_t wb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(wb, wa, 0x0c0c0400u); /* [wa.b0, wb.b0, 0, 0] */
#else
    return (wa & 255u) | ((wb & 255u) << 8);
#endif
}
MM_HD float mm_mhp_u_hi(uint32_t s16) { return fmaf((float)s16, 0x1.0p-16f, 0x1.0p-16f); }        /* (s16 + 1) 2^-16: exact */
MM_HD float mm_mhp_u(uint32_t s16, uint32_t s8) { return fmaf((float)(s16 * 256u + s8), 0x1.0p-24f, 0x1.0p-24f); } /* exact */

/* `ratio > ln u` of the f32 accept rule (MH: metropolis_hastings.rs:311, strict), ln u = mm_lnu_f32(u) by definition, decided
 * WITHOUT the table whenever possible (round 5; the f32 sibling of mm_ratio_exceeds_ln_u below): on the device the
 * hardware's base-2 logarithm (v_log_f32, 1 ulp) times ln 2 is within MM_LNU_BAND (1 + |ln u|) of the table's value over
 * the whole lattice u = (s + 1) 2^-24 (tests/test_gpu_parity.py::test_lnu_filter_bound_holds_over_the_whole_lattice: the
 * largest deviation is a tenth of the band), so a ratio outside that band around it is decided there -- two instructions
 * and no LDS gather -- and only a ratio inside it, a few lanes in a million, evaluates the table logarithm.  The DECISION
 * is that of the plain comparison in every case (NaN compares false everywhere and rejects); only its cost changes: the
 * 16-byte gather of the logarithm's row was 3.5 % of the MH step (profiles/r4u_mh_table_probe3.log).  Host builds take
 * the plain comparison: same decisions, so the twins stay bit-identical. */
#define MM_LNU_BAND 2e-6f
template <class Tab> MM_HD bool mm_ratio_exceeds_lnu_f32(float ratio, float u, const Tab &tab)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
    const float band = fmaf(__builtin_fabsf(l), MM_LNU_BAND, MM_LNU_BAND);
    const float d = ratio - l;
    bool acc = d > band;
    if (__builtin_expect(__builtin_fabsf(d) <= band, 0))
        acc = ratio > mm_lnu_f32(u, tab);
    return acc;
#else
    return ratio > mm_lnu_f32(u, tab);
#endif
}
/* `ratio > ln u` for the MH sampler's paired stream (below: u = (s16 256 + s8 + 1) 2^-24 with s8 from a second block): decided
 * from u_hi = (s16 + 1) 2^-16 alone whenever ratio lies outside [ln u_lo, ln u_hi] widened by the filter's band -- `low`
 * (s8, the second Philox block) is called only inside, then the comparison is the plain one on the exact u.  Bounds:
 * mm_lnu_f32 is within a tenth of the band of v_log_f32 * ln 2 (lattice test) and u_lo <= u <= u_hi. */
template <class Tab, class Low> MM_HD bool mm_ratio_exceeds_lnu_mhp(float ratio, float u_hi, const Low &low, const Tab &tab)
{
    const uint32_t s16 = (uint32_t)(u_hi * 0x1.0p16f) - 1u; /* exact */
#if defined(__HIP_DEVICE_COMPILE__)
    const float u_lo = u_hi - 255.0f * 0x1.0p-24f; /* (s16 256 + 1) 2^-24: exact */
    const float l_hi = __builtin_amdgcn_logf(u_hi) * 0.6931471805599453f, l_lo = __builtin_amdgcn_logf(u_lo) * 0.6931471805599453f;
    const float b_hi = fmaf(__builtin_fabsf(l_hi), MM_LNU_BAND, MM_LNU_BAND), b_lo = fmaf(__builtin_fabsf(l_lo), MM_LNU_BAND, MM_LNU_BAND);
    bool acc = ratio - l_hi > b_hi;
    if (__builtin_expect(!acc && !(ratio - l_lo <= -b_lo), 0)) /* NaN lands here and is rejected by the plain comparison */
        acc = ratio > mm_lnu_f32(mm_mhp_u(s16, low()), tab);
    return acc;
#else
    return ratio > mm_lnu_f32(mm_mhp_u(s16, low()), tab);
#endif
}
/* the accept uniform's logarithm per element type: f32 the table above, f64 mm_log */
template <class Tab> MM_HD float mm_ln_accept(float u, const Tab &tab) { return mm_lnu_f32(u, tab); }
template <class Tab> MM_HD double mm_ln_accept(double u, const Tab &) { return mm_log(u); }

/* `ratio > ln u` for an f64 uniform, decided without the f64 logarithm whenever possible: the f32 table logarithm of
 * (float)u is within 1e-6 (1 + |ln u|) of ln u (generous: the conversion moves ln u by 6e-8, the table is good to
 * ~1e-7 relative; tests/test_engine_stream.py samples the bound), so a ratio outside that band around it is decided
 * there and only a ratio inside the band -- one proposal in a million -- pays for mm_log.  The DECISION is the one of
 * the full comparison in every case (NaN falls through to it and rejects), only its cost changes: the f64 logarithm
 * with its division was a third of the integer-state MH step. */
/* ... given lf = mm_lnu_f32((float)u): the integer-state kernel's noise waves compute it (a pure function of the draw) */
MM_HD bool mm_ratio_exceeds_ln_u_given(double ratio, double u, float lf)
{
    const double mid = (double)lf, band = 1e-6 + 1e-6 * (mid < 0 ? -mid : mid);
    if (ratio > mid + band)
        return true;
    if (ratio <= mid - band)
        return false;
    return ratio > mm_log(u);
}
MM_HD bool mm_ratio_exceeds_ln_u(double ratio, double u)
{
    return mm_ratio_exceeds_ln_u_given(ratio, u, mm_lnu_f32((float)u, mm_icdf_global()));
}

typedef struct {
    uint32_t w[4];
} mm_u32x4;

MM_HD void mm_mulhilo32(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *lo = (uint32_t)p;
    *hi = (uint32_t)(p >> 32);
}

/* a ^ b ^ c.  gfx950 has a three-input bit operation (v_bitop3_b32, truth table 0x96 = odd parity); this compiler does
 * not form it from two xors on its own, and the Philox rounds are two such terms each: 20 vector instructions per block,
 * a fifth of an MH transition.  Same bits either way. */
MM_HD uint32_t mm_xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__) && defined(__has_builtin)
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
#else
    return a ^ b ^ c;
#endif
}

/* Philox4x32-10 (Salmon et al., SC'11).  MM_PHILOX_ROUNDS exists for ONE purpose: timing probes (tools/split_probe.hip
 * -DMM_PHILOX_ROUNDS=7: the fewest rounds the authors report as passing BigCrush) -- the engine's stream IS ten rounds
 * (the oracle, the numpy statement in the tests and every bit-exact comparison say so); a build with another count fails
 * those tests. */
#ifndef MM_PHILOX_ROUNDS
#define MM_PHILOX_ROUNDS 10
#endif
MM_HD mm_u32x4 mm_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < MM_PHILOX_ROUNDS; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mm_mulhilo32(0xD2511F53u, c0, &hi0, &lo0);
        mm_mulhilo32(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = mm_xor3(hi1, c1, k0);
        uint32_t n2 = mm_xor3(hi0, c3, k1);
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    mm_u32x4 o;
    o.w[0] = c0;
    o.w[1] = c1;
    o.w[2] = c2;
    o.w[3] = c3;
    return o;
}

MM_HD mm_u32x4 mm_block(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t block)
{
    return mm_philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iteration, block, (uint32_t)seed,
                            (uint32_t)(seed >> 32));
}
/* the MH sampler's paired stream (above): the block of iterations 2k, 2k + 1 and the low byte of an iteration's uniform */
MM_HD mm_u32x4 mm_mhp_block(uint64_t seed, uint64_t chain, uint32_t iteration) { return mm_block(seed, chain, iteration >> 1, MM_MHP_BLOCK); }
MM_HD uint32_t mm_mhp_low_byte(uint64_t seed, uint64_t chain, uint32_t iteration)
{
    const mm_u32x4 a = mm_block(seed, chain, iteration >> 1, MM_MHP_BLOCK | 1u);
    return (a.w[0] >> (8u * (iteration & 1u))) & 255u;
}

/* s = low bytes of w0, w1, w2 side by side.  On the device two byte permutes (v_perm_b32: selector bytes 0-3 take bytes
 * of the second operand, 4-7 of the first, 12 is the constant 0) instead of three masks, two shifts and two ors; and
 * (float)(s + 1) 2^-24 as fma((float)s, 2^-24, 2^-24): s < 2^24 converts exactly and (s + 1) 2^-24 is representable, so the
 * fma returns exactly the product of the plain form -- same bits, one instruction less. */
MM_HD uint32_t mm_spare_bytes(uint32_t w0, uint32_t w1, uint32_t w2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __builtin_amdgcn_perm(w1, w0, 0x0c0c0400u); /* [w0.b0, w1.b0, 0, 0] */
    return __builtin_amdgcn_perm(w2, t, 0x0c040100u);               /* [t.b0, t.b1, w2.b0, 0] */
#else
    return (w0 & 255u) | ((w1 & 255u) << 8) | ((w2 & 255u) << 16);
#endif
}
MM_HD float mm_spare_u24(mm_u32x4 b)
{
    const uint32_t s = mm_spare_bytes(b.w[0], b.w[1], b.w[2]);
#if defined(__HIP_DEVICE_COMPILE__)
    return fmaf((float)s, 0x1.0p-24f, 0x1.0p-24f);
#else
    return (float)(s + 1u) * 0x1.0p-24f;
#endif
}

MM_HD double mm_u53(uint32_t hi, uint32_t lo)
{
    uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)(m + 1ull) * 0x1.0p-53;
}

MM_HD void mm_box_muller_f64(double u1, double u2, double *z0, double *z1)
{
    double r = sqrt(-2.0 * mm_log(u1));
    double s, c;
    mm_sincos2pi(u2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

/* ---- two iterations at a time (f32): lane 0 = (chain, iteration), lane 1 = (chain, iteration + 1) ----
 * The same functions of (seed, chain, iteration, block) as above, evaluated for two consecutive iterations with the
 * integer rounds interleaved and the float work packed (mm_math.h, two-lane section). */
typedef struct {
    mm_u2 w[4];
} mm_u32x4x2;

/* the pair evaluation in three steps, so that a kernel can spread the ten rounds over other work (mm_kernels.h) */
typedef struct {
    mm_u2 c0, c1, c2, c3;
    uint32_t k0, k1;
} mm_philox_pair;

MM_HD mm_philox_pair mm_philox_pair_init(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t block)
{
    mm_philox_pair s;
    s.c0 = mm_u2{(uint32_t)chain, (uint32_t)chain};
    s.c1 = mm_u2{(uint32_t)(chain >> 32), (uint32_t)(chain >> 32)};
    s.c2 = mm_u2{iteration, iteration + 1u};
    s.c3 = mm_u2{block, block};
    s.k0 = (uint32_t)seed;
    s.k1 = (uint32_t)(seed >> 32);
    return s;
}

MM_HD void mm_philox_pair_round(mm_philox_pair &s)
{
    mm_ul2 p0 = __builtin_convertvector(s.c0, mm_ul2) * 0xD2511F53ull;
    mm_ul2 p1 = __builtin_convertvector(s.c2, mm_ul2) * 0xCD9E8D57ull;
    mm_u2 hi0 = __builtin_convertvector(p0 >> 32, mm_u2), lo0 = __builtin_convertvector(p0, mm_u2);
    mm_u2 hi1 = __builtin_convertvector(p1 >> 32, mm_u2), lo1 = __builtin_convertvector(p1, mm_u2);
    mm_u2 n0, n2;
    n0[0] = mm_xor3(hi1[0], s.c1[0], s.k0);
    n0[1] = mm_xor3(hi1[1], s.c1[1], s.k0);
    n2[0] = mm_xor3(hi0[0], s.c3[0], s.k1);
    n2[1] = mm_xor3(hi0[1], s.c3[1], s.k1);
    s.c0 = n0;
    s.c1 = lo1;
    s.c2 = n2;
    s.c3 = lo0;
    s.k0 += 0x9E3779B9u;
    s.k1 += 0xBB67AE85u;
}

MM_HD mm_u32x4x2 mm_philox_pair_words(const mm_philox_pair &s)
{
    mm_u32x4x2 o;
    o.w[0] = s.c0;
    o.w[1] = s.c1;
    o.w[2] = s.c2;
    o.w[3] = s.c3;
    return o;
}

MM_HD mm_u32x4x2 mm_block_pair(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t block)
{
    mm_philox_pair s = mm_philox_pair_init(seed, chain, iteration, block);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < MM_PHILOX_ROUNDS; ++r)
        mm_philox_pair_round(s);
    return mm_philox_pair_words(s);
}

MM_HD mm_f2 mm_spare_u24x2(mm_u32x4x2 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    mm_f2 r;
    r[0] = fmaf((float)mm_spare_bytes(b.w[0][0], b.w[1][0], b.w[2][0]), 0x1.0p-24f, 0x1.0p-24f);
    r[1] = fmaf((float)mm_spare_bytes(b.w[0][1], b.w[1][1], b.w[2][1]), 0x1.0p-24f, 0x1.0p-24f);
    return r;
#else
    mm_u2 s = (b.w[0] & 255u) | ((b.w[1] & 255u) << 8) | ((b.w[2] & 255u) << 16);
    return mm_u2_to_f2(s + 1u) * mm_splat2(0x1.0p-24f);
#endif
}

/* auxiliary 53-bit uniform k of an iteration (NUTS; f64 accept uniform uses k = 0) */
MM_HD double mm_aux_u53(uint64_t seed, uint64_t chain, uint32_t iteration, uint32_t k)
{
    mm_u32x4 b = mm_block(seed, chain, iteration, MM_AUX_BLOCK + (k >> 1));
    return (k & 1u) ? mm_u53(b.w[2], b.w[3]) : mm_u53(b.w[0], b.w[1]);
}

#endif /* MM_RNG_H */
