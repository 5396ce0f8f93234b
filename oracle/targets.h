/*
 * oracle/targets.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's built-in targets on the hot path (distributions.rs), with the
 * analytic gradients that replace burn's reverse-mode autodiff:
 *   Gaussian2D            distributions.rs:193-205   (MH target; adj(cov)/det re-derived per call)
 *   DiffableGaussian2D    distributions.rs:227-251, 262-288, 296-315
 *   IsotropicGaussian     distributions.rs:394-402   (as a target)
 *   Rosenbrock2D          distributions.rs:497-524
 *   RosenbrockND          distributions.rs:536-546
 *   StandardNormal        nuts.rs:1027-1037          (test-only target of the reference)
 *   GaussianND            NOT in the reference: dense zero-mean Gaussian with precision matrix A,
 *                         the synthetic target BASELINE.json config 5 needs.
 *
 * A target is a POD descriptor; `kind` selects the formula, `p` holds its parameters, `mat` an optional
 * dense matrix (GaussianND precision, row-major dim x dim, kept as double and cast per use).
 */
#ifndef ORACLE_TARGETS_H
#define ORACLE_TARGETS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    OT_GAUSSIAN2D = 0,          /* p = mean0, mean1, c00, c01, c10, c11 */
    OT_DIFFABLE_GAUSSIAN2D = 1, /* p = mean0, mean1, inv00, inv01, inv10, inv11, norm_const (see ot_make_diffable_gaussian2d) */
    OT_ISOTROPIC_GAUSSIAN = 2,  /* p = std */
    OT_ROSENBROCK2D = 3,        /* p = a, b */
    OT_ROSENBROCK_ND = 4,       /* no params */
    OT_STANDARD_NORMAL = 5,     /* no params */
    OT_GAUSSIAN_ND = 6          /* mat = precision A [dim x dim] */
};

typedef struct {
    int kind;
    int dim;
    double p[8];
    const double *mat;
} ot_target;

/* log-density only */
float ot_logp_f32(const ot_target *t, const float *x);
double ot_logp_f64(const ot_target *t, const double *x);
/* log-density and gradient (analytic) */
float ot_logp_grad_f32(const ot_target *t, const float *x, float *g);
double ot_logp_grad_f64(const ot_target *t, const double *x, double *g);

/* distributions.rs:227-251 DiffableGaussian2D::new, evaluated in f32 (t_is_f32) or f64 */
void ot_make_diffable_gaussian2d(ot_target *t, const double mean[2], const double cov[4], int t_is_f32);

#ifdef __cplusplus
}
#endif
#endif
