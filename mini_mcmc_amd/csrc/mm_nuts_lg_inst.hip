/* NUTS lane-group / MFMA kernel instances (mm_nuts_lg.h): GaussianND, f64, D in {16, 32} */
#include "mm_nuts_kernels.h"
#include "mm_nuts_lg.h"
#define MM_HMC_LG_KERNELS 1
#include "mm_hmc_lg.h"

#define MM_LGENTRY(DIM)                                                                                           \
    {                                                                                                             \
        DIM, &mm_launch_nuts_init<double, double, mm_target_gnd_grp4<double, DIM>, mm_red_grp4<double, DIM>>,     \
            &mm_launch_nuts_lg<DIM>, &mm_launch_nuts_lgc_transition<DIM>,                                         \
            &mm_launch_nuts_lgq<DIM>,                                                                             \
            /* sized for either build of the scheduler */                                                         \
            (mm_lg_cfg<DIM, 2>::scratch_doubles_per_wave > mm_lg_cfg<DIM, 1>::scratch_doubles_per_wave ? mm_lg_cfg<DIM, 2>::scratch_doubles_per_wave : mm_lg_cfg<DIM, 1>::scratch_doubles_per_wave), mm_lg_cfg<DIM>::rec_doubles(1)                              \
    }

static const mm_nuts_lg_entry g_lg_table[] = {MM_LGENTRY(16), MM_LGENTRY(32)};

const mm_nuts_lg_entry *mm_nuts_lg_table(int *n)
{
    *n = (int)(sizeof(g_lg_table) / sizeof(g_lg_table[0]));
    return g_lg_table;
}

/* HMC on the same mapping (mm_hmc_lg.h) */
hipError_t mm_launch_hmc_lg(int dim, const mm_hmc_lg_args &a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 15) / 16);
    if (dim == 16)
        hipLaunchKernelGGL((mm_hmc_lg_kernel<16>), dim3(grid), dim3(64), 0, stream, a);
    else if (dim == 32)
        hipLaunchKernelGGL((mm_hmc_lg_kernel<32>), dim3(grid), dim3(64), 0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t mm_launch_hmc_lg32(int dim, const mm_hmc_lg32_args &a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 15) / 16);
    if (dim == 16)
        hipLaunchKernelGGL((mm_hmc_lg32_kernel<16>), dim3(grid), dim3(64), 0, stream, a);
    else if (dim == 32)
        hipLaunchKernelGGL((mm_hmc_lg32_kernel<32>), dim3(grid), dim3(64), 0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
