/* NUTS kernel instances: tensors double, scalars double (see mm_nuts_inst.inc) */
#define MM_TT double
#define MM_ST double
#include "mm_nuts_inst.inc"

const mm_nuts_entry<double, double> *mm_nuts_table_m2(int *n)
{
    *n = (int)(sizeof(g_nuts_table) / sizeof(g_nuts_table[0]));
    return g_nuts_table;
}
