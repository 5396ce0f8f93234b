/* f64 kernel instances (see mm_inst.inc) */
#define MM_INST_T double
#include "mm_inst.inc"

const mm_kernel_entry<double> *mm_kernel_table_f64(int *n)
{
    *n = (int)(sizeof(g_table) / sizeof(g_table[0]));
    return g_table;
}
const mm_noise_entry<double> *mm_noise_table_f64(int *n)
{
    *n = (int)(sizeof(g_noise) / sizeof(g_noise[0]));
    return g_noise;
}
