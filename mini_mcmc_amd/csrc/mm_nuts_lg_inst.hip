/* NUTS lane-group / MFMA kernel instances (mm_nuts_lg.h): GaussianND, f64, D in {16, 32} */
#include "mm_nuts_kernels.h"
#include "mm_nuts_lg.h"

#define MM_LGENTRY(DIM)                                                                                           \
    {                                                                                                             \
        DIM, &mm_launch_nuts_init<double, double, mm_target_gnd_grp4<double, DIM>, mm_red_grp4<double, DIM>>,     \
            &mm_launch_nuts_lg<DIM>, &mm_launch_nuts_lgc_transition<DIM>,                                         \
            &mm_launch_nuts_lgq<DIM>,                                                                             \
            mm_lg_cfg<DIM>::scratch_doubles_per_wave, mm_lg_cfg<DIM>::rec_doubles(1)                              \
    }

static const mm_nuts_lg_entry g_lg_table[] = {MM_LGENTRY(16), MM_LGENTRY(32)};

const mm_nuts_lg_entry *mm_nuts_lg_table(int *n)
{
    *n = (int)(sizeof(g_lg_table) / sizeof(g_lg_table[0]));
    return g_lg_table;
}
