/*
 * mm_tuning.h -- measurement-only environment knobs.
 *
 * The library reads no environment variable unless it is built with -DMMCMC_TUNING (make TUNING=1): a shipped
 * libmmcmc.so selects its kernels from the arguments alone.  With the flag, the MMCMC_* variables named at their
 * call sites (slab counts, scheduler patience, forced kernel variants) are honoured -- for the sweeps under tools/.
 */
#ifndef MM_TUNING_H
#define MM_TUNING_H
#include <cstdlib>
static inline const char *mm_tuning_env(const char *name)
{
#ifdef MMCMC_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
#endif
