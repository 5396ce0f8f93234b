/* run-time-dimension kernels (mm_generic.h): instances and launchers */
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mm_generic.h"

template <class T> static hipError_t launch_run_generic(const mm_gen_args<T> &a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 63ull) / 64ull);
    const size_t lds = mm_generic_store_bytes<T>(a.sampler, a.dim);
    if (!a.scratch) {
        if (lds > MM_GENERIC_LDS_MAX)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL((mm_run_generic_kernel<T, true>), dim3(grid), dim3(64), lds, stream, a);
    } else {
        hipLaunchKernelGGL((mm_run_generic_kernel<T, false>), dim3(grid), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}
hipError_t mm_launch_run_generic_f32(const mm_gen_args<float> &a, hipStream_t stream) { return launch_run_generic<float>(a, stream); }
hipError_t mm_launch_run_generic_f64(const mm_gen_args<double> &a, hipStream_t stream) { return launch_run_generic<double>(a, stream); }

template <class T>
static hipError_t launch_logp_grad_generic(const mm_tparams<T> &P, int kind, int dim, const T *x, T *logp, T *grad, T *scratch,
                                           unsigned long long n, unsigned long long n_pad, hipStream_t stream)
{
    const unsigned int block = 64, grid = (unsigned int)((n + block - 1) / block);
    hipLaunchKernelGGL((mm_logp_grad_generic_kernel<T>), dim3(grid), dim3(block), 0, stream, P, kind, dim, x, logp, grad, scratch, n,
                       n_pad);
    return hipGetLastError();
}
hipError_t mm_launch_logp_grad_generic_f32(const mm_tparams<float> &P, int kind, int dim, const float *x, float *logp, float *grad,
                                           float *scratch, unsigned long long n, unsigned long long n_pad, hipStream_t stream)
{
    return launch_logp_grad_generic<float>(P, kind, dim, x, logp, grad, scratch, n, n_pad, stream);
}
hipError_t mm_launch_logp_grad_generic_f64(const mm_tparams<double> &P, int kind, int dim, const double *x, double *logp,
                                           double *grad, double *scratch, unsigned long long n, unsigned long long n_pad,
                                           hipStream_t stream)
{
    return launch_logp_grad_generic<double>(P, kind, dim, x, logp, grad, scratch, n, n_pad, stream);
}

template <class T>
static hipError_t launch_noise_generic(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                       unsigned long long n, T *z, T *u, hipStream_t stream, int mh)
{
    const unsigned int block = 256, grid = (unsigned int)((n + block - 1) / block);
    hipLaunchKernelGGL((mm_noise_generic_kernel<T>), dim3(grid), dim3(block), 0, stream, seed, chain_offset, iteration, dim, n, z, u, mh);
    return hipGetLastError();
}
hipError_t mm_launch_noise_generic_f32(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                       unsigned long long n, float *z, float *u, hipStream_t stream, int mh)
{
    return launch_noise_generic<float>(seed, chain_offset, iteration, dim, n, z, u, stream, mh);
}
hipError_t mm_launch_noise_generic_f64(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration, int dim,
                                       unsigned long long n, double *z, double *u, hipStream_t stream, int mh)
{
    return launch_noise_generic<double>(seed, chain_offset, iteration, dim, n, z, u, stream, mh);
}
