/*
 * mm_nuts_dims.h -- the (target kind, dimension) pairs the one-chain-per-lane NUTS kernels are instantiated for (an
 * X-macro: csrc/mm_nuts_inst.inc builds the dispatch tables from it, oracle/engine_host.cpp the host twin's).  The
 * reference's NUTS takes any target that implements GradientTarget (nuts.rs:123-129); its N-dimensional built-ins
 * (distributions.rs:350-402, 531-547; nuts.rs:1027-1037) are covered at every dimension up to 8 and at 10, 16, 20
 * (24 / 32 for the dense Gaussian): a NUTS transition keeps both trajectory edges, the proposal and the working leaf
 * in registers, so the dimension is a compile-time constant here; every other dimension runs the run-time-dimension
 * kernel (mm_nuts_generic.h; MH / HMC: mm_generic.h).
 */
#ifndef MM_NUTS_DIMS_H
#define MM_NUTS_DIMS_H

#define MM_NUTS_ND_SMALL(X, KIND) X(KIND, 2) X(KIND, 3) X(KIND, 4) X(KIND, 5) X(KIND, 6) X(KIND, 7) X(KIND, 8)
#define MM_NUTS_INSTANCES(X)                                                                                      \
    X(MM_DIFFABLE_GAUSSIAN2D, 2) X(MM_GAUSSIAN2D, 2) X(MM_ROSENBROCK2D, 2)                                        \
    X(MM_STANDARD_NORMAL, 1) MM_NUTS_ND_SMALL(X, MM_STANDARD_NORMAL) X(MM_STANDARD_NORMAL, 10)                    \
    X(MM_STANDARD_NORMAL, 16) X(MM_STANDARD_NORMAL, 20)                                                           \
    X(MM_ISOTROPIC_GAUSSIAN, 1) MM_NUTS_ND_SMALL(X, MM_ISOTROPIC_GAUSSIAN) X(MM_ISOTROPIC_GAUSSIAN, 10)           \
    X(MM_ISOTROPIC_GAUSSIAN, 16) X(MM_ISOTROPIC_GAUSSIAN, 20)                                                     \
    MM_NUTS_ND_SMALL(X, MM_ROSENBROCK_ND) X(MM_ROSENBROCK_ND, 10) X(MM_ROSENBROCK_ND, 16)                         \
    MM_NUTS_ND_SMALL(X, MM_GAUSSIAN_ND) X(MM_GAUSSIAN_ND, 10) X(MM_GAUSSIAN_ND, 16) X(MM_GAUSSIAN_ND, 20)         \
    X(MM_GAUSSIAN_ND, 24) X(MM_GAUSSIAN_ND, 32)

#endif /* MM_NUTS_DIMS_H */
