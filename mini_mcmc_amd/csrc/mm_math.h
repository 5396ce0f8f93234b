/*
 * mm_math.h -- elementary functions with ONE definition for host and device.
 *
 * The engine's accept masks must be reproducible on the host bit for bit, so log / exp / sincos are not
 * taken from a libm (glibc and the ROCm device library round differently) but are spelled out here with
 * explicit fma()/fmaf() and IEEE +,-,*,/ only.  Everything is compiled with -ffp-contract=off on both
 * sides, so the only fused operations are the ones written below.  sqrt and division are IEEE correctly
 * rounded on both sides (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt).
 *
 * Plain C99 / C++ / HIP.  No reference code is involved: the reference (Rust) calls the platform libm.
 */
#ifndef MM_MATH_H
#define MM_MATH_H

#if defined(__HIPCC_RTC__)
/* hipRTC (run-time compiled user targets, mm_rtc.hip): no host headers; the compiler's built-in runtime header declares
 * the device math functions, these are the fixed-width types this code uses */
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef __SIZE_TYPE__ size_t;
#else
#include <math.h>
#include <stdint.h>
#include <string.h>
#endif
/* spelled as builtins so that the header needs no libc declarations */
#define MM_INFINITY_F (__builtin_huge_valf())
#define MM_NAN_F (__builtin_nanf(""))

#if defined(__cplusplus)
/* std::conditional without <type_traits> (the header also compiles under hipRTC, which has no host headers) */
template <bool C, class A, class B> struct mm_cond { typedef A type; };
template <class A, class B> struct mm_cond<false, A, B> { typedef B type; };
#endif

#if defined(__HIPCC__)
#define MM_HD __host__ __device__ __forceinline__
#elif defined(__cplusplus)
#define MM_HD inline
#else
#define MM_HD static inline
#endif

#if defined(__clang__)
#define MM_UNROLL _Pragma("unroll")
#else
#define MM_UNROLL
#endif

MM_HD uint32_t mm_f2u(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
MM_HD float mm_u2f(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
MM_HD uint64_t mm_d2u(double f)
{
    uint64_t u;
    __builtin_memcpy(&u, &f, 8);
    return u;
}
MM_HD double mm_u2d(uint64_t u)
{
    double f;
    __builtin_memcpy(&f, &u, 8);
    return f;
}

/* ------------------------------------------------------------------ f32 */

/* natural log for normal positive finite x (callers pass uniforms in (0,1] or small positive numbers).
 * Mantissa in [sqrt(1/2), sqrt(2)), degree-8 polynomial in f = m - 1 (Cephes logf scheme), ~1 ulp. */
MM_HD float mm_logf(float x)
{
    uint32_t ix = mm_f2u(x);
    int e = (int)(ix >> 23) - 127;
    float m = mm_u2f((ix & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (p * f) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    return fmaf(fe, 0.693359375f, r);
}

/* exp(x) for f32; clamps to 0 / +inf outside [-104, 88.7]. ~1 ulp. */
MM_HD float mm_expf(float x)
{
    if (!(x == x))
        return x;
    if (x > 88.72283905206835f)
        return MM_INFINITY_F;
    if (x < -103.97208f)
        return 0.0f;
    float kf = rintf(x * 1.44269504088896341f);
    int k = (int)kf;
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float rr = r * r;
    float y = fmaf(p, rr, r) + 1.0f;
    /* scale by 2^k in two steps so that subnormal results round once */
    int k1 = k / 2, k2 = k - k1;
    float s1 = mm_u2f((uint32_t)(k1 + 127) << 23);
    float s2 = mm_u2f((uint32_t)(k2 + 127) << 23);
    return (y * s1) * s2;
}

/* sin(2*pi*u), cos(2*pi*u) for u in [0,1]: exact quadrant reduction, Taylor kernels on |y| <= pi/4 */
MM_HD void mm_sincos2pif(float u, float *s, float *c)
{
    float t = u * 4.0f;      /* exact */
    float qf = rintf(t);     /* nearest quadrant boundary, 0..4 */
    float r = (t - qf);      /* exact, in [-0.5, 0.5] */
    float y = r * 1.57079632679489662f; /* |y| <= pi/4 */
    float y2 = y * y;
    float sp = 2.7557313707e-6f; /* 1/9! */
    sp = fmaf(sp, y2, -1.9841269841e-4f);
    sp = fmaf(sp, y2, 8.3333333333e-3f);
    sp = fmaf(sp, y2, -1.6666666667e-1f);
    float sy = fmaf(sp * y2, y, y);
    float cp = -2.7557319224e-7f; /* -1/10! */
    cp = fmaf(cp, y2, 2.4801587302e-5f);
    cp = fmaf(cp, y2, -1.3888888889e-3f);
    cp = fmaf(cp, y2, 4.1666666667e-2f);
    cp = fmaf(cp, y2, -0.5f);
    float cy = fmaf(cp, y2, 1.0f);
    int q = (int)qf & 3;
    float ss = (q & 1) ? cy : sy;
    float cc = (q & 1) ? sy : cy;
    if (q == 1 || q == 2)
        cc = -cc;
    if (q >= 2)
        ss = -ss;
    *s = ss;
    *c = cc;
}

/* ------------------------------------------------------------------ f32, two lanes at a time
 *
 * A dependent chain of scalar VALU instructions issues one instruction per ~5 cycles on gfx950, a packed
 * instruction (v_pk_fma_f32, ...) processes two elements in the same time (tools/valu_rate.hip): the two-element
 * vector types and the few helpers on them that the kernels use (Philox pair, RosenbrockND gradient).  Element k of
 * every result is computed by EXACTLY the scalar operations.  GCC vector extensions: accepted by g++ (host build) and
 * clang (hipcc). */
typedef float mm_f2 __attribute__((vector_size(8)));
typedef int32_t mm_i2 __attribute__((vector_size(8)));
typedef uint32_t mm_u2 __attribute__((vector_size(8)));
typedef uint64_t mm_ul2 __attribute__((vector_size(16)));

MM_HD mm_f2 mm_splat2(float v)
{
    mm_f2 r = {v, v};
    return r;
}
MM_HD mm_f2 mm_fma2(mm_f2 a, mm_f2 b, mm_f2 c)
{
#if defined(__clang__)
    return __builtin_elementwise_fma(a, b, c);
#else
    mm_f2 r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    return r;
#endif
}
MM_HD mm_f2 mm_u2_to_f2(mm_u2 a) { return __builtin_convertvector(a, mm_f2); }
/* ------------------------------------------------------------------ f64 */

/* natural log, normal positive finite x. fdlibm-style: s = f/(2+f), odd polynomial in s, ~1 ulp. */
MM_HD double mm_log(double x)
{
    uint64_t ix = mm_d2u(x);
    int e = (int)(ix >> 52) - 1023;
    double m = mm_u2d((ix & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                              2.857142874366239149e-01),
                        6.666666666666735130e-01);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    /* log(1+f) = f - hfsq + s*(hfsq+R) */
    double lo = fma(dk, 1.90821492927058770002e-10, s * (hfsq + R));
    return fma(dk, 6.93147180369123816490e-01, f - (hfsq - lo));
}

/* exp(x), f64, ~1 ulp; clamps outside [-745.2, 709.78]. */
MM_HD double mm_exp(double x)
{
    if (!(x == x))
        return x;
    if (x > 709.782712893384)
        return (double)MM_INFINITY_F;
    if (x < -745.1332191019411)
        return 0.0;
    double kf = rint(x * 1.44269504088896338700e+00);
    int k = (int)kf;
    double hi = fma(kf, -6.93147180369123816490e-01, x);
    double lo = kf * 1.90821492927058770002e-10;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * fma(t, fma(t, fma(t, fma(t, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                          6.61375632143793436117e-05),
                                  -2.77777777770155933842e-03),
                           1.66666666666666019037e-01);
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    int k1 = k / 2, k2 = k - k1;
    double s1 = mm_u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = mm_u2d((uint64_t)(k2 + 1023) << 52);
    return (y * s1) * s2;
}

/* mm_expf with its three early returns as final selects (see mm_exp_sel below): same bits, no exec-mask regions; device only */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float mm_expf_sel(float x)
{
    const float kf = rintf(x * 1.44269504088896341f);
    const int k = (int)kf;
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float rr = r * r;
    const float y = fmaf(p, rr, r) + 1.0f;
    const int k1 = k / 2, k2 = k - k1;
    const float s1 = mm_u2f((uint32_t)(k1 + 127) << 23);
    const float s2 = mm_u2f((uint32_t)(k2 + 127) << 23);
    float e = (y * s1) * s2;
    e = x < -103.97208f ? 0.0f : e;
    e = x > 88.72283905206835f ? MM_INFINITY_F : e;
    e = x == x ? e : x;
    return e;
}
#endif

/* mm_exp with its three early returns as final selects: the same operations on every input that takes none of them, the
 * same constants otherwise -- the same bits -- and no exec-mask regions.  Device code only (out of range the unused
 * conversion to int would be undefined behaviour on the host; the GPU's saturates). */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double mm_exp_sel(double x)
{
    const double kf = rint(x * 1.44269504088896338700e+00);
    const int k = (int)kf;
    const double hi = fma(kf, -6.93147180369123816490e-01, x);
    const double lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * fma(t, fma(t, fma(t, fma(t, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                                6.61375632143793436117e-05),
                                        -2.77777777770155933842e-03),
                                 1.66666666666666019037e-01);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k1 = k / 2, k2 = k - k1;
    const double s1 = mm_u2d((uint64_t)(k1 + 1023) << 52);
    const double s2 = mm_u2d((uint64_t)(k2 + 1023) << 52);
    double e = (y * s1) * s2;
    e = x < -745.1332191019411 ? 0.0 : e;
    e = x > 709.782712893384 ? (double)MM_INFINITY_F : e;
    e = x == x ? e : x;
    return e;
}
#endif

/* sin(2*pi*u), cos(2*pi*u) for u in [0,1], f64 */
MM_HD void mm_sincos2pi(double u, double *s, double *c)
{
    double t = u * 4.0;
    double qf = rint(t);
    double r = t - qf;
    double y = r * 1.57079632679489661923;
    double z = y * y;
    /* fdlibm __kernel_sin / __kernel_cos polynomials on |y| <= pi/4 */
    double sr = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                  2.75573137070700676789e-06),
                           -1.98412698298579493134e-04),
                    8.33333333332248946124e-03);
    double sy = fma(z * y, fma(z, sr, -1.66666666666666324348e-01), y);
    double cr = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                         -2.75573143513906633035e-07),
                                  2.48015872894767294178e-05),
                           -1.38888888888741095749e-03),
                    4.16666666666666019037e-02);
    double cy = fma(z * z, cr, fma(-0.5, z, 1.0));
    int q = (int)qf & 3;
    double ss = (q & 1) ? cy : sy;
    double cc = (q & 1) ? sy : cy;
    if (q == 1 || q == 2)
        cc = -cc;
    if (q >= 2)
        ss = -ss;
    *s = ss;
    *c = cc;
}

#endif /* MM_MATH_H */
