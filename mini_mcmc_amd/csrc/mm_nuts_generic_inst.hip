/* run-time-dimension NUTS (mm_nuts_generic.h): instances and launchers for the three type modes */
#include <hip/hip_runtime.h>

#include "mm_nuts_generic.h"

template <class TT, class ST> static hipError_t launch_nuts_generic(const mm_gen_nuts_args<TT, ST> &a, int init, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 63ull) / 64ull);
    if (init)
        hipLaunchKernelGGL((mm_nuts_generic_init_kernel<TT, ST>), dim3(grid), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((mm_nuts_generic_run_kernel<TT, ST>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}
hipError_t mm_launch_nuts_generic_m0(const mm_gen_nuts_args<float, double> &a, int init, hipStream_t stream)
{
    return launch_nuts_generic<float, double>(a, init, stream);
}
hipError_t mm_launch_nuts_generic_m1(const mm_gen_nuts_args<float, float> &a, int init, hipStream_t stream)
{
    return launch_nuts_generic<float, float>(a, init, stream);
}
hipError_t mm_launch_nuts_generic_m2(const mm_gen_nuts_args<double, double> &a, int init, hipStream_t stream)
{
    return launch_nuts_generic<double, double>(a, init, stream);
}
