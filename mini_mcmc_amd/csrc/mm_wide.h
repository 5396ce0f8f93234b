/* mm_wide.h -- HMC with one chain per WORKGROUP (mm_wide.hip): few chains of a huge dimension */
#ifndef MM_WIDE_H
#define MM_WIDE_H

#include <hip/hip_runtime.h>

#include "mm_targets.h"

#define MM_WIDE_MAX_DIM (32 * 1024) /* 32 coordinates per thread x 1024 threads */

template <class T> struct mm_wide_args {
    mm_tparams<T> P;
    int kind, dim;
    T eps;
    int n_leapfrog;
    T *state;   /* [n_chains, dim] */
    T *out;     /* [n_chains, n_total, dim] or NULL */
    unsigned long long *accept, *accept_total;
    unsigned long long n_chains, seed, chain_offset, n_total;
    unsigned int iter0, n_discard, n_collect, out_t0;
};

inline bool mm_wide_kind_ok(int kind)
{
    return kind == MM_ISOTROPIC_GAUSSIAN || kind == MM_ROSENBROCK_ND || kind == MM_STANDARD_NORMAL;
}
hipError_t mm_launch_hmc_wide_f32(const mm_wide_args<float> &a, hipStream_t stream);
hipError_t mm_launch_hmc_wide_f64(const mm_wide_args<double> &a, hipStream_t stream);

#endif /* MM_WIDE_H */
