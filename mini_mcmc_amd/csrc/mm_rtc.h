/* mm_rtc.h -- run-time compiled user targets (mm_rtc.hip): what the rest of the library sees */
#ifndef MM_RTC_H
#define MM_RTC_H

#include <hip/hip_runtime.h>

#define MM_USER_KIND_BASE 1000 /* = MMCMC_USER_KIND_BASE: kinds handed out by mmcmc_target_register_source */

struct mm_user_target; /* opaque */
const mm_user_target *mm_rtc_find(int kind);
int mm_rtc_dim(const mm_user_target *t);
/* a MODEL (target + user proposal, mmcmc_proposal_register_source): MH kernels only; base kind = the target it wraps
 * (a built-in kind, or a user kind); -1 for a plain user target */
int mm_rtc_base_kind(const mm_user_target *t);
int mm_rtc_is_model(const mm_user_target *t);
/* an integer-state model (mmcmc_discrete_register_source); `args` = mm_discrete_user_args, one wave per workgroup */
int mm_rtc_is_discrete(const mm_user_target *t);
/* a unit the library built for itself (mm_rtc_builtin / mm_rtc_builtin_nuts): not a kind a caller may pass in a description */
int mm_rtc_is_internal(const mm_user_target *t);
/* MMCMC_RTC_COMPILER_HIPCC / _HIPRTC: which compiler built the unit */
int mm_rtc_compiler(const mm_user_target *t);
/* a built-in target's register-resident MH / HMC kernels for a dimension <= 32 without a compiled instance, built on first
 * use (NULL: no hipRTC, or the unit failed to build) */
const mm_user_target *mm_rtc_builtin(int kind, int dim);
const mm_user_target *mm_rtc_builtin_nuts(int kind, int dim); /* the same for NUTS: init + run, three type modes */
hipError_t mm_rtc_launch_discrete(const mm_user_target *t, void *args, size_t args_bytes, unsigned int grid, hipStream_t stream);
/* kernels of a registered target: sampler 0 MH / 1 HMC, dtype 0 f32 / 1 f64; `args` = the mm_run_args<T> block */
hipError_t mm_rtc_launch_run_split(const mm_user_target *t, int sampler, void *args, size_t args_bytes, unsigned int grid, size_t lds,
                                   hipStream_t stream);
hipError_t mm_rtc_launch_run(const mm_user_target *t, int sampler, int dtype, void *args, size_t args_bytes, unsigned int grid,
                             unsigned int block, size_t lds, hipStream_t stream);
/* NUTS kernels of a registered target: mode 0 / 1 / 2 (mm_nuts_api.hip), init = 1: init_chain (`args` = mm_nuts_init_args<TT, ST>),
 * 0: the run with the lanes in step (`args` = mm_nuts_args<TT, ST>, stack in the HBM scratch area), 2: the run with asynchronous
 * lanes and leaves in pairs (mm_nuts_pair_body; dynamic LDS = the ring of uniforms; hipErrorNotFound if the unit lacks it: it is
 * compiled up to MM_RTC_NUTS_PAIR_MAX_DIM; the caller decides by the size of a chain's vectors); one wave per workgroup */
#define MM_RTC_NUTS_PAIR_MAX_DIM 32
/* init_chain and the asynchronous-lane pair kernel of `mode` exist on the current device (the only NUTS run kernel of a
 * run-time compiled unit that is launched: mm_nuts_api.hip) */
int mm_rtc_nuts_usable(const mm_user_target *t, int mode);
hipError_t mm_rtc_launch_nuts(const mm_user_target *t, int mode, int init, void *args, size_t args_bytes, unsigned int grid, size_t lds,
                              hipStream_t stream);
/* unnorm_logp / unnorm_logp_and_grad of n rows; `args` = {mm_tparams<T> P, const T *x, T *logp, T *grad, u64 n} packed */
hipError_t mm_rtc_launch_logp_grad(const mm_user_target *t, int dtype, void *args, size_t args_bytes, unsigned long long n,
                                   hipStream_t stream);

#endif /* MM_RTC_H */
