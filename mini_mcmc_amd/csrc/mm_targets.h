/*
 * mm_targets.h -- built-in target log-densities and analytic gradients (host + device, one definition).
 *
 * GPU-side counterpart of the reference's `Target::unnorm_logp` (distributions.rs:105-108),
 * `BatchedGradientTarget::unnorm_logp_batch` (:65-76) and `GradientTarget::unnorm_logp_and_grad` (:78-88)
 * for the closed set of built-ins on the hot path:
 *     Gaussian2D (:193-205)   DiffableGaussian2D (:227-315)   IsotropicGaussian (:394-402)
 *     Rosenbrock2D (:497-524) RosenbrockND (:536-546)         StandardNormal (nuts.rs:1027-1037)
 *     GaussianND (dense precision matrix; not in the reference, BASELINE.json config 5)
 * Gradients are analytic (the reference differentiates the same expressions with burn's autodiff).
 * Arithmetic: explicit fma()/fmaf() only, compiled with -ffp-contract=off, so a host build of this header
 * reproduces the device results bit for bit.  A new target is one more functor with the same two members;
 * that is the C++ analogue of implementing the reference's traits.
 */
#ifndef MM_TARGETS_H
#define MM_TARGETS_H

#include "mm_math.h"

enum mm_target_kind {
    MM_GAUSSIAN2D = 0,
    MM_DIFFABLE_GAUSSIAN2D = 1,
    MM_ISOTROPIC_GAUSSIAN = 2,
    MM_ROSENBROCK2D = 3,
    MM_ROSENBROCK_ND = 4,
    MM_STANDARD_NORMAL = 5,
    MM_GAUSSIAN_ND = 6,
    MM_N_TARGET_KINDS = 7
};

/* Device-side parameter block, passed to kernels by value (lands in SGPRs / scalar cache).
 *   Gaussian2D, DiffableGaussian2D : p = mean0, mean1, inv00, inv01, inv10, inv11, norm_const
 *   IsotropicGaussian              : p = 1/std^2
 *   Rosenbrock2D                   : p = a, b
 *   GaussianND                     : mat -> precision matrix A, row-major [dim x dim], element type T */
template <class T> struct mm_tparams {
    T p[8];
    const T *mat;
};

MM_HD float mm_fma(float a, float b, float c) { return fmaf(a, b, c); }
MM_HD double mm_fma(double a, double b, double c) { return fma(a, b, c); }

template <class T, int KIND, int D> struct mm_target;

/* Gaussian2D / DiffableGaussian2D: lp = nc - 1/2 d^T Sinv d ; grad = -1/2 (Sinv + Sinv^T) d */
template <class T, int KIND> struct mm_gauss2d_impl {
    static constexpr int dim = 2;
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T d0 = x[0] - P.p[0], d1 = x[1] - P.p[1];
        T z0 = mm_fma(d1, P.p[4], d0 * P.p[2]);
        T z1 = mm_fma(d1, P.p[5], d0 * P.p[3]);
        T q = mm_fma(z1, d1, z0 * d0);
        return mm_fma(T(-0.5), q, P.p[6]);
    }
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T d0 = x[0] - P.p[0], d1 = x[1] - P.p[1];
        T z0 = mm_fma(d1, P.p[4], d0 * P.p[2]);
        T z1 = mm_fma(d1, P.p[5], d0 * P.p[3]);
        T q = mm_fma(z1, d1, z0 * d0);
        T y0 = mm_fma(P.p[3], d1, P.p[2] * d0);
        T y1 = mm_fma(P.p[5], d1, P.p[4] * d0);
        g[0] = T(-0.5) * (z0 + y0);
        g[1] = T(-0.5) * (z1 + y1);
        return mm_fma(T(-0.5), q, P.p[6]);
    }
};
template <class T> struct mm_target<T, MM_GAUSSIAN2D, 2> : mm_gauss2d_impl<T, MM_GAUSSIAN2D> {};
template <class T> struct mm_target<T, MM_DIFFABLE_GAUSSIAN2D, 2> : mm_gauss2d_impl<T, MM_DIFFABLE_GAUSSIAN2D> {};

/* IsotropicGaussian as target: -1/2 sum(x^2) / std^2 ; p[0] = 1/std^2 */
template <class T, int D> struct mm_target<T, MM_ISOTROPIC_GAUSSIAN, D> {
    static constexpr int dim = D;
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T s = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            s = mm_fma(x[i], x[i], s);
        return (T(-0.5) * s) * P.p[0];
    }
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T s = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            s = mm_fma(x[i], x[i], s);
            g[i] = -(x[i] * P.p[0]);
        }
        return (T(-0.5) * s) * P.p[0];
    }
};

/* Rosenbrock2D: -[(a - x)^2 + b (y - x^2)^2] */
template <class T> struct mm_target<T, MM_ROSENBROCK2D, 2> {
    static constexpr int dim = 2;
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T u = P.p[0] - x[0];
        T w = mm_fma(-x[0], x[0], x[1]);
        return -mm_fma(P.p[1] * w, w, u * u);
    }
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T u = P.p[0] - x[0];
        T w = mm_fma(-x[0], x[0], x[1]);
        T bw = P.p[1] * w;
        g[0] = mm_fma(T(4) * x[0], bw, T(2) * u);
        g[1] = T(-2) * bw;
        return -mm_fma(bw, w, u * u);
    }
};

/* RosenbrockND: -sum_{i<D-1} [100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2] */
template <class T, int D> struct mm_target<T, MM_ROSENBROCK_ND, D> {
    static constexpr int dim = D;
    MM_HD static T logp(const mm_tparams<T> &, const T *x)
    {
        T acc = 0;
        MM_UNROLL
        for (int i = 0; i + 1 < D; ++i) {
            T t = mm_fma(-x[i], x[i], x[i + 1]);
            T u = T(1) - x[i];
            acc = mm_fma(T(100) * t, t, acc);
            acc = mm_fma(u, u, acc);
        }
        return -acc;
    }
    MM_HD static T logp_grad(const mm_tparams<T> &, const T *x, T *g)
    {
        T acc = 0;
        T tprev = 0;
        MM_UNROLL
        for (int i = 0; i + 1 < D; ++i) {
            T t = mm_fma(-x[i], x[i], x[i + 1]);
            T u = T(1) - x[i];
            acc = mm_fma(T(100) * t, t, acc);
            acc = mm_fma(u, u, acc);
            T a = mm_fma(T(400) * x[i], t, T(2) * u);
            g[i] = (i > 0) ? mm_fma(T(-200), tprev, a) : a;
            tprev = t;
        }
        if (D >= 2)
            g[D - 1] = T(-200) * tprev;
        else
            g[0] = 0;
        return -acc;
    }
};

/* f32 RosenbrockND gradient with coordinates processed in pairs (v_pk_*_f32 on the device): every element goes
 * through the same operations as in the generic template above -- 2(1 - x) is formed as fma(-2, x, 2), which is the
 * same number because scaling by two commutes with rounding -- so the results are bit-identical to it; the packed
 * form just issues ~4 fewer instructions per leapfrog step at D = 3 (the kernels are instruction-issue bound). */
template <int D> struct mm_target<float, MM_ROSENBROCK_ND, D> {
    static constexpr int dim = D;
    MM_HD static float logp(const mm_tparams<float> &, const float *x)
    {
        float acc = 0;
        MM_UNROLL
        for (int i = 0; i + 1 < D; ++i) {
            float t = fmaf(-x[i], x[i], x[i + 1]);
            float u = 1.0f - x[i];
            acc = fmaf(100.0f * t, t, acc);
            acc = fmaf(u, u, acc);
        }
        return -acc;
    }
    MM_HD static float logp_grad(const mm_tparams<float> &, const float *x, float *g)
    {
        float t[D > 1 ? D - 1 : 1];
        float acc = 0;
        MM_UNROLL
        for (int i = 0; i + 1 < D; ++i) {
            t[i] = fmaf(-x[i], x[i], x[i + 1]);
            float u = 1.0f - x[i];
            acc = fmaf(100.0f * t[i], t[i], acc);
            acc = fmaf(u, u, acc);
        }
        /* a_i = fma(400 x_i, t_i, 2 - 2 x_i) for i < D-1, two coordinates per packed instruction */
        MM_UNROLL
        for (int i = 0; i + 1 < D; i += 2) {
            if (i + 2 < D) {
                mm_f2 xx = {x[i], x[i + 1]};
                mm_f2 tt = {t[i], t[i + 1]};
                mm_f2 u2 = mm_fma2(mm_splat2(-2.0f), xx, mm_splat2(2.0f));
                mm_f2 a = mm_fma2(xx * mm_splat2(400.0f), tt, u2);
                g[i] = (i > 0) ? fmaf(-200.0f, t[i - 1], a[0]) : a[0];
                g[i + 1] = fmaf(-200.0f, t[i], a[1]);
            } else {
                float u2 = fmaf(-2.0f, x[i], 2.0f);
                float a = fmaf(400.0f * x[i], t[i], u2);
                g[i] = (i > 0) ? fmaf(-200.0f, t[i - 1], a) : a;
            }
        }
        if (D >= 2)
            g[D - 1] = -200.0f * t[D - 2];
        else
            g[0] = 0;
        return -acc;
    }
};

/* StandardNormal: -1/2 sum(x^2) */
template <class T, int D> struct mm_target<T, MM_STANDARD_NORMAL, D> {
    static constexpr int dim = D;
    MM_HD static T logp(const mm_tparams<T> &, const T *x)
    {
        T s = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            s = mm_fma(x[i], x[i], s);
        return T(-0.5) * s;
    }
    MM_HD static T logp_grad(const mm_tparams<T> &, const T *x, T *g)
    {
        T s = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            s = mm_fma(x[i], x[i], s);
            g[i] = -x[i];
        }
        return T(-0.5) * s;
    }
};

/* GaussianND: -1/2 x^T A x, grad = -A x.  A is the same for every chain: its elements are wave-uniform
 * scalar loads, the products run on the vector ALU with one scalar operand. */
template <class T, int D> struct mm_target<T, MM_GAUSSIAN_ND, D> {
    static constexpr int dim = D;
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T q = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            T y = 0;
            MM_UNROLL
            for (int j = 0; j < D; ++j)
                y = mm_fma(P.mat[i * D + j], x[j], y);
            q = mm_fma(x[i], y, q);
        }
        return T(-0.5) * q;
    }
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T q = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            T y = 0;
            MM_UNROLL
            for (int j = 0; j < D; ++j)
                y = mm_fma(P.mat[i * D + j], x[j], y);
            q = mm_fma(x[i], y, q);
            g[i] = -y;
        }
        return T(-0.5) * q;
    }
};

/* Order of the D-term dot products inside a transition (HMC kinetic energy, NUTS criteria, the dense Gaussian's x.y).
 *   mm_red_seq : one fma chain over i = 0..D-1 (what one lane per chain does).
 *   mm_red_grp4: four interleaved chains -- chain q sums the terms i = q (mod 4) in ascending order -- combined as
 *                (c0 + c1) + (c2 + c3).  This is the order the lane-group kernel (mm_nuts_lg.h: four lanes per chain,
 *                lane q owning the coordinates i = q mod 4, butterfly over the four lanes) produces, so the host build
 *                of mm_nuts_step with this policy is its bit-exact twin. */
template <class TT, int D> struct mm_red_seq {
    MM_HD static TT dot(const TT *a, const TT *b)
    {
        TT s = 0;
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            s = mm_fma(a[i], b[i], s);
        return s;
    }
};
template <class TT, int D> struct mm_red_grp4 {
    MM_HD static TT dot(const TT *a, const TT *b)
    {
        TT c[4] = {0, 0, 0, 0};
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            c[i & 3] = mm_fma(a[i], b[i], c[i & 3]);
        return (c[0] + c[1]) + (c[2] + c[3]);
    }
};

/* GaussianND with the grouped reduction for x.y (the matrix-vector product itself stays the in-order fma chain over
 * the columns, which is exactly what v_mfma_f64_16x16x4_f64 computes -- tools/mfma_f64_check.hip). */
template <class T, int D> struct mm_target_gnd_grp4 {
    static constexpr int dim = D;
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T y[D];
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            T acc = 0;
            MM_UNROLL
            for (int j = 0; j < D; ++j)
                acc = mm_fma(P.mat[i * D + j], x[j], acc);
            y[i] = acc;
            g[i] = -acc;
        }
        return T(-0.5) * mm_red_grp4<T, D>::dot(x, y);
    }
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T g[D];
        return logp_grad(P, x, g);
    }
};


/* The f32 lane-group kernel (mm_hmc_lg.h) on v_mfma_f32_16x16x4_f32: the matrix instruction returns its 16 x 16 result in
 * row blocks (lane group q holds rows 4 q .. 4 q + 3 of a tile), so lane q of a chain owns the coordinates
 * coord(q, s) = 16 (s >> 2) + 4 q + (s & 3), s = 0 .. D/4 - 1.  Two consequences for the order of sums, which the host
 * twin reproduces:
 *   mm_red_blk4:        partial sum c_q over the coordinates i with (i >> 2) & 3 == q, ascending; total (c0 + c1) + (c2 + c3)
 *   mm_target_gnd_blk4: row i of A x is the fma chain over the columns in k-step order -- step s takes the columns
 *                       coord(0, s), coord(1, s), coord(2, s), coord(3, s) (tools/mfma_f32_check.hip: the hardware's
 *                       accumulation is exactly this chain). */
MM_HD int mm_blk4_coord(int q, int s) { return 16 * (s >> 2) + 4 * q + (s & 3); }
template <class TT, int D> struct mm_red_blk4 {
    MM_HD static TT dot(const TT *a, const TT *b)
    {
        TT c[4] = {0, 0, 0, 0};
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            c[(i >> 2) & 3] = mm_fma(a[i], b[i], c[(i >> 2) & 3]);
        return (c[0] + c[1]) + (c[2] + c[3]);
    }
};
template <class T, int D> struct mm_target_gnd_blk4 {
    static constexpr int dim = D;
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g)
    {
        T y[D];
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            T acc = 0;
            MM_UNROLL
            for (int s = 0; s < D / 4; ++s) {
                MM_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const int j = mm_blk4_coord(k, s);
                    acc = mm_fma(P.mat[i * D + j], x[j], acc);
                }
            }
            y[i] = acc;
            g[i] = -acc;
        }
        return T(-0.5) * mm_red_blk4<T, D>::dot(x, y);
    }
    MM_HD static T logp(const mm_tparams<T> &P, const T *x)
    {
        T g[D];
        return logp_grad(P, x, g);
    }
};

#endif /* MM_TARGETS_H */
