/* f32 kernel instances (see mm_inst.inc) */
#define MM_INST_T float
#include "mm_inst.inc"

const mm_kernel_entry<float> *mm_kernel_table_f32(int *n)
{
    *n = (int)(sizeof(g_table) / sizeof(g_table[0]));
    return g_table;
}
const mm_noise_entry<float> *mm_noise_table_f32(int *n)
{
    *n = (int)(sizeof(g_noise) / sizeof(g_noise[0]));
    return g_noise;
}

/* dynamic LDS of the split-role kernel for (float, dim, sampler): what a run-time compiled user-target kernel of the same
 * shape is launched with (csrc/mm_rtc.hip; dim <= 8, else 0) */
size_t mm_split_lds_bytes_f32(int dim, int mh)
{
#define MM_SPLIT_LDS_CASE(DD)                                                                                       \
    case DD:                                                                                                        \
        return mh ? mm_split_plan<float, DD, true>::lds_bytes : mm_split_plan<float, DD, false>::lds_bytes;
    switch (dim) {
        MM_SPLIT_LDS_CASE(1)
        MM_SPLIT_LDS_CASE(2)
        MM_SPLIT_LDS_CASE(3)
        MM_SPLIT_LDS_CASE(4)
        MM_SPLIT_LDS_CASE(5)
        MM_SPLIT_LDS_CASE(6)
        MM_SPLIT_LDS_CASE(7)
        MM_SPLIT_LDS_CASE(8)
    default:
        return 0;
    }
#undef MM_SPLIT_LDS_CASE
}
