/* oracle/targets.c -- TEST INFRASTRUCTURE ONLY. See targets.h. */
#include "targets.h"

#include <math.h>

#define REAL float
#define SUF f32
#include "targets_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF f64
#include "targets_impl.inc"
#undef REAL
#undef SUF

/*
 * distributions.rs:227-251  DiffableGaussian2D::new(mean, cov) for scalar type T:
 *   inv_det = 1/det; inv_cov = [[c11*inv_det, -c01*inv_det], [-c10*inv_det, c00*inv_det]]
 *   norm_const = -(2 ln(2 pi) + ln det) / 2
 * computed in T (f32 when t_is_f32), stored widened to double.
 */
void ot_make_diffable_gaussian2d(ot_target *t, const double mean[2], const double cov[4], int t_is_f32)
{
    t->kind = OT_DIFFABLE_GAUSSIAN2D;
    t->dim = 2;
    t->mat = 0;
    t->p[0] = mean[0];
    t->p[1] = mean[1];
    if (t_is_f32) {
        float c00 = (float)cov[0], c01 = (float)cov[1], c10 = (float)cov[2], c11 = (float)cov[3];
        float det = c00 * c11 - c01 * c10;
        float inv_det = 1.0f / det;
        t->p[0] = (float)mean[0];
        t->p[1] = (float)mean[1];
        t->p[2] = c11 * inv_det;
        t->p[3] = -c01 * inv_det;
        t->p[4] = -c10 * inv_det;
        t->p[5] = c00 * inv_det;
        float two = 2.0f;
        t->p[6] = -(two * logf(two * (float)M_PI) + logf(det)) / two;
    } else {
        double c00 = cov[0], c01 = cov[1], c10 = cov[2], c11 = cov[3];
        double det = c00 * c11 - c01 * c10;
        double inv_det = 1.0 / det;
        t->p[2] = c11 * inv_det;
        t->p[3] = -c01 * inv_det;
        t->p[4] = -c10 * inv_det;
        t->p[5] = c00 * inv_det;
        t->p[6] = -(2.0 * log(2.0 * M_PI) + log(det)) / 2.0;
    }
    t->p[7] = 0;
}
