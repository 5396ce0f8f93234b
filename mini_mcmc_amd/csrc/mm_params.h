/*
 * mm_params.h -- host-side derivation of the device parameter block from a target description.
 *
 * Gaussian2D / DiffableGaussian2D: Sigma^-1 = adj(Sigma)/det and the normalising constant are derived once, in
 * the element type T, the way DiffableGaussian2D::new does (distributions.rs:227-251); the reference's Gaussian2D
 * (distributions.rs:193-205) re-derives the same inverse on every call and has no constant.
 * Host only (plain C++), shared by the C ABI and by the host build of the samplers used in parity tests.
 */
#ifndef MM_PARAMS_H
#define MM_PARAMS_H

#include <cmath>
#include <cstring>

#include "mm_targets.h"

/* returns 0 on success, -1 on an invalid description (singular covariance, non-positive std) */
template <class T> int mm_fill_params(int kind, const double params[8], mm_tparams<T> *P)
{
    std::memset(P, 0, sizeof *P);
    switch (kind) {
    case MM_GAUSSIAN2D:
    case MM_DIFFABLE_GAUSSIAN2D: {
        T c00 = (T)params[2], c01 = (T)params[3], c10 = (T)params[4], c11 = (T)params[5];
        T det = c00 * c11 - c01 * c10;
        if (!(det != (T)0) || !std::isfinite((double)det))
            return -1;
        T inv_det = (T)1 / det;
        P->p[0] = (T)params[0];
        P->p[1] = (T)params[1];
        P->p[2] = c11 * inv_det;
        P->p[3] = -c01 * inv_det;
        P->p[4] = -c10 * inv_det;
        P->p[5] = c00 * inv_det;
        if (kind == MM_DIFFABLE_GAUSSIAN2D) {
            T two = (T)2;
            P->p[6] = -(two * (T)std::log((double)(two * (T)M_PI)) + (T)std::log((double)det)) / two;
        }
        return 0;
    }
    case MM_ISOTROPIC_GAUSSIAN: {
        T s = (T)params[0];
        if (!(s > (T)0))
            return -1;
        P->p[0] = (T)1 / (s * s);
        return 0;
    }
    case MM_ROSENBROCK2D:
        P->p[0] = (T)params[0];
        P->p[1] = (T)params[1];
        return 0;
    default:
        return 0;
    }
}

#endif /* MM_PARAMS_H */
