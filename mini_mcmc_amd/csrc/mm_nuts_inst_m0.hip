/* NUTS kernel instances: tensors float, scalars double (see mm_nuts_inst.inc) */
#define MM_TT float
#define MM_ST double
#include "mm_nuts_inst.inc"

const mm_nuts_entry<float, double> *mm_nuts_table_m0(int *n)
{
    *n = (int)(sizeof(g_nuts_table) / sizeof(g_nuts_table[0]));
    return g_nuts_table;
}
