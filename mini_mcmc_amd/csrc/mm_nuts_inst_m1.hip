/* NUTS kernel instances: tensors float, scalars float (see mm_nuts_inst.inc) */
#define MM_TT float
#define MM_ST float
#include "mm_nuts_inst.inc"

const mm_nuts_entry<float, float> *mm_nuts_table_m1(int *n)
{
    *n = (int)(sizeof(g_nuts_table) / sizeof(g_nuts_table[0]));
    return g_nuts_table;
}
