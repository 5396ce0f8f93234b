/*
 * mmcmc.h -- C ABI of libmmcmc.so: the MI355X-native many-chain engine for mini-mcmc's per-chain inner loop.
 *
 * This is the drop-in boundary.  The reference (mini-mcmc v0.8.3, Rust) has no FFI of its own: its boundary is
 * the trait + constructor surface of the crate.  Each entry point below names the reference item it stands in
 * for (file:line in the reference tree); INTEGRATION.md shows the `extern "C"` block and the safe Rust wrapper
 * (`GpuMetropolisHastings`, `GpuHmc`, `GpuNuts`) a maintainer would add on top.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types; no exceptions cross the boundary
 *   - every function returns an int status: MMCMC_OK (0), negative = invalid argument / unsupported,
 *     positive = the hipError_t of the failing HIP call.  mmcmc_status_string() names them.
 *   - sample layout is the reference's: [n_chains, n_collect, dim] row-major (core.rs:176-186, hmc.rs:157)
 *   - a handle owns its device state (current positions, iteration counter, seed), so calling run() twice
 *     continues the chains exactly as the Rust sampler structs do
 *   - handles are not thread-safe; one handle = one device; `stream` arguments are hipStream_t passed as
 *     void* (NULL = the handle's own stream, a BLOCKING stream: it orders itself with work on the legacy default
 *     stream, so a caller working on the default stream -- e.g. torch's -- sees the results without a host
 *     synchronisation); run() is asynchronous when `out` is device memory and no host-side result (accept
 *     counts) is requested -- call mmcmc_*_sync() or synchronise the stream.
 *   - the library has NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef MMCMC_H
#define MMCMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101 (round 6): + mmcmc_{mh,hmc}_group_run_async, mmcmc_*_group_stats_phases, mmcmc_device_pci_bus_id,
 *   MMCMC_ERR_GROUP_BROKEN; mmcmc_*_group_run blocks again whatever its arguments (100 inferred "asynchronous" from
 *   out_host == NULL && accept_counts == NULL); mmcmc_nuts_set_repacking (100, measured slower, removed) stays removed.
 * 102 (round 6): + mmcmc_rtc_compiler_info; long-chain diagnostics take N1 = ceil(n/2 / 1024) residues (any count).
 * A binding checks mmcmc_version() >= the version it was generated from (rust/mini-mcmc-hip: assert_abi, in every constructor). */
#define MMCMC_VERSION 102 /* 0.1.2 */

/* ---- status ---- */
#define MMCMC_OK 0
#define MMCMC_ERR_INVALID_ARG (-1)  /* NULL pointer, zero chains, negative count, ...            */
#define MMCMC_ERR_UNSUPPORTED (-2)  /* target kind / dim / dtype combination has no kernel        */
#define MMCMC_ERR_SHAPE (-3)        /* the analogue of ndarray::ShapeError from core.rs:184       */
#define MMCMC_ERR_NO_DEVICE (-4)    /* no HIP device: the engine never falls back to the CPU      */
#define MMCMC_ERR_STATE (-5)        /* call made in the wrong state (e.g. timing before any run)  */
#define MMCMC_ERR_GROUP_BROKEN (-6) /* a device group whose shards stand at different iterations (a run failed on some) */

/* ---- element types ---- */
#define MMCMC_F32 0
#define MMCMC_F64 1

/* ---- built-in targets: the closed registry standing in for `impl Target` / `impl BatchedGradientTarget`
 *      / `impl GradientTarget` (distributions.rs:65-108).  params[] meaning per kind:
 *   MMCMC_GAUSSIAN2D           mean0, mean1, cov00, cov01, cov10, cov11     distributions.rs:158-206
 *   MMCMC_DIFFABLE_GAUSSIAN2D  mean0, mean1, cov00, cov01, cov10, cov11     distributions.rs:212-316
 *   MMCMC_ISOTROPIC_GAUSSIAN   std                                          distributions.rs:344-402
 *   MMCMC_ROSENBROCK2D         a, b                                         distributions.rs:490-524
 *   MMCMC_ROSENBROCK_ND        (none)                                       distributions.rs:528-547
 *   MMCMC_STANDARD_NORMAL      (none)                                       nuts.rs:1024-1037 (test target)
 *   MMCMC_GAUSSIAN_ND          matrix = precision A, row-major dim x dim    (not in the reference; config 5) */
#define MMCMC_GAUSSIAN2D 0
#define MMCMC_DIFFABLE_GAUSSIAN2D 1
#define MMCMC_ISOTROPIC_GAUSSIAN 2
#define MMCMC_ROSENBROCK2D 3
#define MMCMC_ROSENBROCK_ND 4
#define MMCMC_STANDARD_NORMAL 5
#define MMCMC_GAUSSIAN_ND 6

typedef struct mmcmc_target_desc {
    int32_t kind;
    int32_t dim;
    double params[8];
    const double *matrix; /* host pointer, only for MMCMC_GAUSSIAN_ND */
} mmcmc_target_desc;

/* `impl Proposal` (distributions.rs:92-101): the reference's IsotropicGaussian (distributions.rs:344-392), or a kind
 * handed out by mmcmc_proposal_register_source (>= MMCMC_USER_PROPOSAL_BASE): a proposal of the caller's own, `std` = its
 * one run-time parameter */
#define MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN 0
#define MMCMC_USER_PROPOSAL_BASE 1000
typedef struct mmcmc_proposal_desc {
    int32_t kind;
    int32_t reserved;
    double std;
} mmcmc_proposal_desc;

/* summary of one diagnostic (stats.rs:373-381 BasicStats) */
typedef struct mmcmc_basic_stats {
    float min, median, max, mean, std;
} mmcmc_basic_stats;
/* stats.rs:338-342 RunStats */
typedef struct mmcmc_run_stats {
    mmcmc_basic_stats ess;
    mmcmc_basic_stats rhat;
} mmcmc_run_stats;

/* timing of the most recent run(): device time of the sampling kernel(s), measured with HIP events on the
 * stream the kernels were launched on */
typedef struct mmcmc_timing {
    float kernel_ms;      /* sum over launches */
    uint32_t n_launches;
    uint32_t reserved;
    uint64_t out_bytes;   /* sample bytes written by the kernels */
    uint64_t state_bytes; /* chain state bytes loaded + stored */
} mmcmc_timing;

/* ---- library ---- */
int mmcmc_version(void);
const char *mmcmc_status_string(int status);
int mmcmc_device_count(int *count);
/* hipDeviceGetPCIBusId of `device` as "dddd:bb:dd.f" into buf (len >= 16): a scaling run prints it per shard so that two
 * "devices" that are one GPU cannot pass for two */
int mmcmc_device_pci_bus_id(int device, char *buf, size_t len);

/* ---- core.rs:394-435  init_with_seed / init_det (seed 42): n*d StandardNormal draws of the reference's own
 *      stream (rand 0.9 SmallRng = xoshiro256++, rand_distr 0.5 ziggurat), row-major, as double. Host only. */
int mmcmc_init_with_seed(size_t n, size_t d, uint64_t seed, double *out);

/* ---- Metropolis-Hastings -----------------------------------------------------------------------------
 * MetropolisHastings::new(target, proposal, initial_states)   metropolis_hastings.rs:149-159
 * init: host, [n_chains, dim] of `dtype` elements */
typedef struct mmcmc_mh mmcmc_mh;
int mmcmc_mh_create(mmcmc_mh **out, const mmcmc_target_desc *target, const mmcmc_proposal_desc *proposal,
                    const void *init, size_t n_chains, int dtype, int device);
/* MetropolisHastings::seed   metropolis_hastings.rs:187-193 */
int mmcmc_mh_seed(mmcmc_mh *h, uint64_t seed);
/* multi-GPU sharding: global id of this handle's first chain (stream key = chain_offset + local index) */
int mmcmc_mh_set_chain_offset(mmcmc_mh *h, uint64_t chain_offset);
/* ChainRunner::run(n_collect, n_discard) -> Array3 [n_chains, n_collect, dim]   core.rs:176-186 (+ :55-73)
 * out: [n_chains, n_collect, dim] of dtype, device memory if out_is_device else host (may be NULL: no samples).
 * accept_counts: host, [n_chains], accepted proposals per chain over the whole call (may be NULL). */
int mmcmc_mh_run(mmcmc_mh *h, size_t n_collect, size_t n_discard, void *out, int out_is_device,
                 uint64_t *accept_counts, void *stream);
/* the same into rows [row0, row0 + n_rows) of a DEVICE buffer out [n_chains, n_total_rows, dim]: one sample filled by
 * several launches (run_progress' segments); n_rows = 0 with n_discard > 0 only advances the chains */
int mmcmc_mh_run_rows(mmcmc_mh *h, size_t n_rows, size_t n_discard, void *out_device, size_t n_total_rows, size_t row0,
                      void *stream);
/* what the handle was created with (any pointer may be NULL) */
int mmcmc_mh_shape(mmcmc_mh *h, size_t *n_chains, int *dim, int *dtype, int *device);
/* MarkovChain::current_state for every chain   core.rs:43-44 ; host [n_chains, dim] */
int mmcmc_mh_state(mmcmc_mh *h, void *out);
int mmcmc_mh_sync(mmcmc_mh *h);
int mmcmc_mh_timing(mmcmc_mh *h, mmcmc_timing *t);
/* run() brackets its launches with two HIP events for mmcmc_*_timing (default on); off = no event packets on the
 * stream between back-to-back runs (mmcmc_*_timing then returns MMCMC_ERR_STATE) */
int mmcmc_mh_enable_timing(mmcmc_mh *h, int on);
int mmcmc_mh_destroy(mmcmc_mh *h);

/* ---- HMC ---------------------------------------------------------------------------------------------
 * HMC::new(target, initial_positions, step_size, n_leapfrog)   hmc.rs:87-109 */
typedef struct mmcmc_hmc mmcmc_hmc;
int mmcmc_hmc_create(mmcmc_hmc **out, const mmcmc_target_desc *target, const void *init, size_t n_chains,
                     double step_size, int n_leapfrog, int dtype, int device);
/* HMC::set_seed   hmc.rs:118-121 (inert in the reference, quirk Q6; here it keys the stream) */
int mmcmc_hmc_seed(mmcmc_hmc *h, uint64_t seed);
int mmcmc_hmc_set_chain_offset(mmcmc_hmc *h, uint64_t chain_offset);
/* HMC::run(n_collect, n_discard) -> Tensor [n_chains, n_collect, dim]   hmc.rs:137-158 */
int mmcmc_hmc_run(mmcmc_hmc *h, size_t n_collect, size_t n_discard, void *out, int out_is_device,
                  uint64_t *accept_counts, void *stream);
int mmcmc_hmc_run_rows(mmcmc_hmc *h, size_t n_rows, size_t n_discard, void *out_device, size_t n_total_rows, size_t row0,
                       void *stream);
int mmcmc_hmc_shape(mmcmc_hmc *h, size_t *n_chains, int *dim, int *dtype, int *device);
/* HMC::step   hmc.rs:304-377 : one transition of every chain */
int mmcmc_hmc_step(mmcmc_hmc *h, void *stream);
/* HMC::positions   hmc.rs:49 ; host [n_chains, dim] */
int mmcmc_hmc_state(mmcmc_hmc *h, void *out);
int mmcmc_hmc_sync(mmcmc_hmc *h);
int mmcmc_hmc_timing(mmcmc_hmc *h, mmcmc_timing *t);
int mmcmc_hmc_enable_timing(mmcmc_hmc *h, int on);
int mmcmc_hmc_destroy(mmcmc_hmc *h);

/* ---- NUTS ------------------------------------------------------------------------------------------------
 * NUTS::new(target, initial_positions, target_accept_p)   nuts.rs:123-129 over NUTSChain::new nuts.rs:410-434.
 * init: host [n_chains, dim] doubles (the reference's Vec<Vec<T>>).  mode selects the reference's type split:
 *   0 = f32 tensors + f64 scalars  (`NUTS<f64, Autodiff<NdArray>, _>`: NdArray elements are f32)
 *   1 = f32 tensors + f32 scalars  (`NUTS<f32, Autodiff<NdArray>, _>`)
 *   2 = f64 tensors + f64 scalars  (BASELINE.json config 5 "f64"; no default reference backend has f64 elements)
 * Samples are of the tensor type (f32 for modes 0/1, f64 for mode 2). */
typedef struct mmcmc_nuts mmcmc_nuts;
int mmcmc_nuts_create(mmcmc_nuts **out, const mmcmc_target_desc *target, const double *init, size_t n_chains,
                      double target_accept_p, int mode, int device);
/* NUTS::set_seed   nuts.rs:347-353 (keys the counter-based stream; chains differ by their global index) */
int mmcmc_nuts_seed(mmcmc_nuts *h, uint64_t seed);
int mmcmc_nuts_set_chain_offset(mmcmc_nuts *h, uint64_t chain_offset);
/* maximum tree depth (doublings per transition), 1..12, default 10.  The reference's `while s` (nuts.rs:578) is
 * unbounded; a lane that never terminates would stall its whole wave, so the engine caps it. */
int mmcmc_nuts_set_max_depth(mmcmc_nuts *h, int max_depth);
/* Kernel mapping (not in the reference).  0 = one chain per lane, the lanes of a wave taking their transitions in step
 * (every target / mode); 4 = one chain per lane, every lane advancing through its transitions on its own, one leaf per
 * tick (dim <= 8; results identical to 0); 5 = the same with the leaves taken in pairs, level 0 of the pending-subtree
 * stack in registers and the auxiliary uniforms in an LDS ring (dim <= 8; the default there; results identical to 0
 * and 4).  6 = one chain per lane at RUN-TIME dimension, the
 * transition's vectors in an HBM store (csrc/mm_nuts_generic.h): the N-dimensional built-in targets (isotropic / standard
 * / dense Gaussian, RosenbrockND) at ANY dimension, all three modes; the default where there is no compiled instance of
 * 0 (those exist for every dim <= 8 and 10, 16, 20, 24 / 32 for the dense Gaussian), selectable everywhere, results
 * identical to 0 where both exist.  1 = lane-group mapping with
 * the gradient on the matrix cores: 16 chains per wave, four lanes per chain, v_mfma_f64_16x16x4 for A x, the whole
 * run in one launch, every wave keeping its 16 chains.  2 and 3 = the same arithmetic with tree-depth compaction
 * (BASELINE.json config 5): a transition is cut at the doubling boundaries and the chains that still double are
 * re-packed into full waves -- 2: one launch per tree level and transition; 3: one persistent kernel whose waves take
 * work units from per-level queues, chains advancing independently of each other (the fastest from a few thousand
 * chains on; below 2048 chains it runs as 1).
 * 1..3 exist for mode 2 + MMCMC_GAUSSIAN_ND with dim 16 or 32, where 3 is the default; elsewhere setting them returns
 * MMCMC_ERR_UNSUPPORTED (so do 4 and 5 above dim 8).  1, 2 and 3 give bit-identical results.  They sum the D-term dot products in a different
 * order than 0 (four interleaved partial sums vs sequential), so their samples differ from 0's in the last bits and,
 * over long trajectories of a stiff target, visibly; each is bit-exact against its own host build
 * (oracle/engine_host.cpp modes 3 and 2).  With variant 3 mmcmc_nuts_run returns after the kernel has finished.
 * 7 (the default of targets compiled at run time: mmcmc_target_register_source, and built-in targets at dimensions 9 .. 31
 * without a compiled instance): the asynchronous-lane kernel with the leaves in pairs (as 5), compiled with the target's functor
 * on first use.  A unit is checked once against the run-time-dimension kernel (6), bit for bit, before a handle relies on
 * it; if that fails a built-in target runs 6 and a user target is refused (MMCMC_ERR_UNSUPPORTED).
 * mmcmc_nuts_kernel_variant returns the mapping in use (>= 0) or a negative status.
 * mmcmc_nuts_set_compaction (variants 2, 3): doublings below `first_level` run before the first re-packing (default 5;
 * variant 3, until this is called, picks the mode of the depths seen so far minus one, after a 16-transition pilot on a
 * fresh handle); variant 2 splits the chains into `n_groups` contiguous groups (1..16; 0 = one per 16 384 chains), each
 * with its own launch sequence on its own stream.  Neither setting changes a result. */
int mmcmc_nuts_set_kernel_variant(mmcmc_nuts *h, int variant);
int mmcmc_nuts_kernel_variant(mmcmc_nuts *h);
int mmcmc_nuts_set_compaction(mmcmc_nuts *h, int first_level, int n_groups);
/* progress = 0: NUTS::run -> NUTSChain::run (nuts.rs:163-170, 457-471): n_collect + n_discard - 1 transitions, and
 *               with n_discard == 0 row 0 is the initial position (the reference's off-by-one, test_chain_1);
 * progress = 1: the stepping of run_progress (nuts.rs:491-522): all n_collect + n_discard transitions;
 * progress = 2: the same with EVERY state recorded, burn-in included (what run_progress' per-chain trackers are fed,
 *               nuts.rs:491-506): out then has n_collect + n_discard rows per chain.
 * out: [n_chains, n_collect, dim] of the tensor type, device or host memory (may be NULL). */
int mmcmc_nuts_run(mmcmc_nuts *h, size_t n_collect, size_t n_discard, void *out, int out_is_device, int progress,
                   void *stream);
int mmcmc_nuts_shape(mmcmc_nuts *h, size_t *n_chains, int *dim, int *mode, int *device);
/* NUTSChain::position for every chain (nuts.rs:369); host [n_chains, dim] of the tensor type */
int mmcmc_nuts_state(mmcmc_nuts *h, void *out);
/* per-chain adaptation state: host [n_chains, 4] doubles = epsilon, epsilon_bar, h_bar, mu (nuts.rs:374-386) */
int mmcmc_nuts_adapt_state(mmcmc_nuts *h, double *out);
/* diagnostics (not in the reference): gradient evaluations per chain so far; histogram[13] of tree depths */
int mmcmc_nuts_leapfrog_counts(mmcmc_nuts *h, uint64_t *out);
int mmcmc_nuts_depth_histogram(mmcmc_nuts *h, uint32_t *out);
int mmcmc_nuts_sync(mmcmc_nuts *h);
int mmcmc_nuts_timing(mmcmc_nuts *h, mmcmc_timing *t);
int mmcmc_nuts_destroy(mmcmc_nuts *h);

/* knobs shared by the samplers:
 * iterations per kernel launch (0 = the whole run in one launch, the default) -- never changes a result;
 * kernel variant: 2 = noise of two iterations packed and software-pipelined (default up to dim 16 in f64, at dim 16 in
 *   f32), 0 = plain (default at dim 32), 5 = workgroups of four transition waves fed by noise waves through an LDS ring
 *   (f32: three noise waves each, 1024 threads, four waves per SIMD; f64: one, 512 threads, two per SIMD) (csrc/mm_split_kernels.h; up to dim 8, the default there in f32, elsewhere
 *   MMCMC_ERR_UNSUPPORTED) -- all bit-identical;
 *   HMC only: 3 = the lane-group mapping with the gradient on the matrix cores (16 chains per wave, four lanes per
 *   chain, v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4; csrc/mm_hmc_lg.h), for MMCMC_GAUSSIAN_ND of dim 16 or 32 in f64 and
 *   f32, where it is the default (elsewhere MMCMC_ERR_UNSUPPORTED).  Like the NUTS mappings it sums the D-term dot
 *   products as four partial sums (f64: interleaved coordinates, f32: blocks of four), so its samples differ from
 *   variants 0 / 2 by rounding; it is bit-exact against its own host build (oracle/engine_host.cpp:
 *   eh_hmc_grouped_run, eh_hmc_grouped_run_f32);
 *   6 = one chain per lane at RUN-TIME dimension (csrc/mm_generic.h), the default for a dimension without a compiled
 *   instance; HMC only: 8 = one chain per WORKGROUP, the coordinates across up to 1024 threads (csrc/mm_wide.hip) -- few
 *   chains of a huge dimension, the shape of the reference's own timing tests (hmc.rs:882-916: 6 chains, D = 10 000) --
 *   for RosenbrockND / IsotropicGaussian / StandardNormal at 4 <= dim <= 32 768, the default there from dim 128 on with
 *   fewer than 1024 chains.  It reduces the three sums of a transition across the workgroup, so its samples differ from
 *   variant 6's by rounding (same stream, same transition). */
int mmcmc_mh_set_iters_per_launch(mmcmc_mh *h, uint32_t iters);
int mmcmc_hmc_set_iters_per_launch(mmcmc_hmc *h, uint32_t iters);
int mmcmc_mh_set_kernel_variant(mmcmc_mh *h, int variant);
int mmcmc_hmc_set_kernel_variant(mmcmc_hmc *h, int variant);
int mmcmc_hmc_kernel_variant(mmcmc_hmc *h); /* the variant in use (>= 0) or a negative status */

/* ---- user-defined targets (csrc/mm_rtc.hip) ------------------------------------------------------------------
 * The reference's `Target` / `GradientTarget` (distributions.rs:65-108) are open traits.  The GPU analogue of
 * `impl GradientTarget for MyDensity`: the HIP source of
 *     template <class T> struct mmcmc_user_target {
 *         static constexpr int dim = <dim>;
 *         MM_HD static T logp(const mm_tparams<T> &P, const T *x);                // unnorm_logp
 *         MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g);     // unnorm_logp_and_grad, returns logp
 *     };
 * (P.p[0..8) = `params` of the mmcmc_target_desc a sampler is later created with, P.mat = its `matrix` [dim, dim] or
 * NULL; everything in csrc/mm_math.h / mm_targets.h -- mm_fma, mm_logf, mm_exp, ... -- is in scope) is compiled with
 * hipRTC into the engine's own MH / HMC kernel skeleton for f32 and f64 and its one-chain-per-lane NUTS kernel for the
 * three type modes (NUTS kernel variant 7), and registered under a new target kind (>= MMCMC_USER_KIND_BASE), which
 * mmcmc_mh_create, mmcmc_hmc_create, mmcmc_nuts_create (nuts.rs:123-129 takes any GradientTarget), mmcmc_*_group_create
 * and mmcmc_logp_grad_batch then accept like a built-in kind (dim 1..32).  `log` (may be NULL) receives the compiler's diagnostics.
 * MMCMC_ERR_INVALID_ARG: the source does not compile (see log); MMCMC_ERR_UNSUPPORTED: no libhiprtc on this machine. */
#define MMCMC_USER_KIND_BASE 1000
int mmcmc_target_register_source(const char *name, int dim, const char *hip_source, int *kind_out, char *log, size_t log_len);

/* Which compiler builds run-time compiled units (process-wide; default AUTO).  AUTO: `hipcc --genco` in a child process
 * wherever hipcc is found (PATH, then /opt/rocm/bin), hipRTC (libhiprtc.so, bound at run time) otherwise.  A hipRTC was
 * caught miscompiling one kernel of these units that hipcc compiles correctly, hence the order; it was the copy PyTorch
 * bundles (ROCm 7.0.2's compiler, which a Python process that imported torch resolves libhiprtc.so.7 to), not the system's
 * 7.2, which emits hipcc's code (csrc/mm_rtc.hip at g_compiler).
 * HIPCC / HIPRTC pin one of them (registration then returns MMCMC_ERR_UNSUPPORTED where it is missing). */
#define MMCMC_RTC_COMPILER_AUTO 0
#define MMCMC_RTC_COMPILER_HIPCC 1
#define MMCMC_RTC_COMPILER_HIPRTC 2
int mmcmc_rtc_set_compiler(int which);
/* which compiler built the unit behind a kind handed out by a mmcmc_*_register_source call: MMCMC_RTC_COMPILER_HIPCC or
 * MMCMC_RTC_COMPILER_HIPRTC (MMCMC_ERR_INVALID_ARG: no such kind) */
int mmcmc_rtc_unit_compiler(int kind);
/* (version 102) which copy of hipRTC this process would compile with -- the file the dynamic loader resolved, e.g.
 * .../torch/lib/libhiprtc.so once PyTorch is imported --, the HIP version (major * 10^7 + minor * 10^5 + patch) of the
 * runtime the process bound (hipRTC and its compiler library come from the same bundle) and of the compiler the library was
 * built with (what a `hipcc --genco` child normally is).  Any pointer may be NULL.  MMCMC_ERR_UNSUPPORTED (empty path): no
 * usable libhiprtc.  Needs no GPU.  A unit built by the hipRTC of an older runtime than the library's compiler says so in the
 * `log` of its register call. */
int mmcmc_rtc_compiler_info(char *hiprtc_path, size_t path_len, int *process_hip_version, int *built_hip_version);

/* ---- user-defined proposals (csrc/mm_rtc.hip) --------------------------------------------------------------------
 * `Proposal` is an open trait too (distributions.rs:92-101), and the reference keeps BOTH q-terms in the acceptance
 * ratio because a proposal may be asymmetric (metropolis_hastings.rs:303-315; its own integration tests define custom
 * proposals, tests/metrohast_poisson_test.rs:50-85).  The GPU analogue of `impl Proposal for MyProposal`: the HIP source of
 *     template <class T> struct mmcmc_user_proposal {
 *         MM_HD static void sample(T sigma, const T *x, const T *z, T *out);   // x' ~ q(. | x) from z[dim] i.i.d. N(0, 1)
 *         MM_HD static T logp(T sigma, const T *from, const T *to);            // log q(to | from)
 *     };
 * (sigma = `std` of the mmcmc_proposal_desc the sampler is created with; MM_USER_DIM = dim; the engine's headers are in
 * scope) is compiled for ONE target -- `target_kind` a built-in kind at dimension `dim`, or a kind handed out by
 * mmcmc_target_register_source -- into the engine's MH skeletons for f32 and f64, whose transition then is
 * MHMarkovChain::step with both q-terms in the reference's order:
 *     log_accept_ratio = (lp(x') + log q(x | x')) - (lp(x) + log q(x' | x));   accept iff log_accept_ratio > ln u
 * The noise z and u are the engine's stream (mm_rng.h), so a run is reproducible and independent of launch partition
 * and sharding like every other.  *proposal_kind_out (>= MMCMC_USER_PROPOSAL_BASE) goes into mmcmc_proposal_desc.kind;
 * mmcmc_mh_create accepts it together with a description of exactly that target kind and dimension
 * (MMCMC_ERR_UNSUPPORTED / MMCMC_ERR_SHAPE otherwise).  Status codes as for mmcmc_target_register_source. */
int mmcmc_proposal_register_source(const char *name, int target_kind, int dim, const char *hip_source, int *proposal_kind_out,
                                   char *log, size_t log_len);

/* ---- device groups: one call runs every chain on N GPUs (csrc/mm_group.hip) ------------------------------------
 * `run` of the reference executes ALL chains of the sampler (ChainRunner::run core.rs:176-186, HMC::run hmc.rs:137-158).
 * A group shards n_chains contiguously over `devices` (device i: global chains [first_i, first_i + n_i), stream keyed by
 * the global index, so the sample does not depend on the number of devices), runs the shards with no data-path
 * collective, and reduces split-R-hat / ESS (stats.rs:416-546) over all chains: per-device
 * sufficient statistics, RCCL ncclAllGather of the per-half-chain means / sums of squares + ncclAllReduce of the lag
 * sums over xGMI; the cross-chain sums of withinvar are then reduced on shard 0 (f64 partial sums of the means relative
 * to the mean of the first half of global chain 0 -- the shift the single-GPU kernel uses -- grouped per rank, so equal to
 * the single-GPU result to rounding, not bit for bit) and the finish is the single-GPU one.  (RCCL is bound at run time; if it is missing, or a
 * device is listed twice -- several shards on one GPU --, the statistics travel through the host; *used_rccl says which:
 * 1 = RCCL; 0 = the host by design (a device listed twice); -1 = the host as a FALLBACK because no RCCL library could be
 * loaded; -2 = the host because ncclCommInitAll failed -- check it in a scaling run.)
 *   init: host [n_chains, dim] of dtype.  run: the sample stays on the devices (mmcmc_hmc_group_shard gives each shard's
 *   device pointer [n_i, n_collect, dim]); out_host, if not NULL, also receives [n_chains, n_collect, dim];
 *   accept_counts: host [n_chains] or NULL.  A second run continues the chains.
 *   _run BLOCKS: it returns when every shard's results are there, also with out_host == NULL and accept_counts == NULL
 *   (the shards' device pointers may then be read from any stream).
 *   _run_async (MH / HMC; version 101 -- version 100 inferred this from two NULL arguments of _run) enqueues every shard's
 *   launch on that shard's stream and returns (as mmcmc_hmc_run does on a caller's stream), so runs issued back to back keep
 *   all devices busy; mmcmc_*_group_sync waits for them and reports their errors.  _state, _split_rhat_mean_ess and a later
 *   run are ordered behind the queued work by the streams; read a shard's device pointer only after _sync.
 *   A run (either spelling) that fails after SOME shards have advanced leaves them at different iterations: what was queued
 *   is waited for, the failing status returned, and every later run / diagnostics call of the group returns
 *   MMCMC_ERR_GROUP_BROKEN (_state, _shard, _destroy still work).
 *   _stats_phases: host wall-clock milliseconds of the last _split_rhat_mean_ess, ms3 = {every shard's local statistics,
 *   the exchange (collectives + cross-chain sums, or the host copies), the finish}.
 *   _stream_timer(g, 0, NULL) records a start event on every shard's stream, _stream_timer(g, 1, ms) the end events, waits
 *   and writes each shard's elapsed device milliseconds into ms [n_devices].
 *   _exchange: how the diagnostics' statistics will travel -- decided in _create, where the RCCL communicators are made
 *   (ncclCommInitAll; *status as *used_rccl above, *n_rccl_ranks = ranks of the communicator or 0). */
typedef struct mmcmc_hmc_group mmcmc_hmc_group;
int mmcmc_hmc_group_create(mmcmc_hmc_group **out, const mmcmc_target_desc *target, const void *init, size_t n_chains,
                           double step_size, int n_leapfrog, int dtype, const int *devices, int n_devices);
int mmcmc_hmc_group_seed(mmcmc_hmc_group *g, uint64_t seed);
int mmcmc_hmc_group_set_chain_offset(mmcmc_hmc_group *g, uint64_t first_global_chain);
int mmcmc_hmc_group_run(mmcmc_hmc_group *g, size_t n_collect, size_t n_discard, void *out_host, uint64_t *accept_counts);
int mmcmc_hmc_group_run_async(mmcmc_hmc_group *g, size_t n_collect, size_t n_discard);
int mmcmc_hmc_group_stats_phases(mmcmc_hmc_group *g, double *ms3);
int mmcmc_hmc_group_state(mmcmc_hmc_group *g, void *out); /* host [n_chains, dim] */
int mmcmc_hmc_group_split_rhat_mean_ess(mmcmc_hmc_group *g, float *rhat, float *ess, int *used_rccl); /* of the last run */
int mmcmc_hmc_group_shard(mmcmc_hmc_group *g, int i, int *device, size_t *first_chain, size_t *n_chains, void **sample_dev);
int mmcmc_hmc_group_sync(mmcmc_hmc_group *g);
int mmcmc_hmc_group_stream_timer(mmcmc_hmc_group *g, int stop, float *ms_per_device);
int mmcmc_hmc_group_exchange(mmcmc_hmc_group *g, int *status, int *n_rccl_ranks);
int mmcmc_hmc_group_destroy(mmcmc_hmc_group *g);
typedef struct mmcmc_mh_group mmcmc_mh_group;
int mmcmc_mh_group_create(mmcmc_mh_group **out, const mmcmc_target_desc *target, const mmcmc_proposal_desc *proposal,
                          const void *init, size_t n_chains, int dtype, const int *devices, int n_devices);
int mmcmc_mh_group_seed(mmcmc_mh_group *g, uint64_t seed);
int mmcmc_mh_group_set_chain_offset(mmcmc_mh_group *g, uint64_t first_global_chain);
int mmcmc_mh_group_run(mmcmc_mh_group *g, size_t n_collect, size_t n_discard, void *out_host, uint64_t *accept_counts);
int mmcmc_mh_group_run_async(mmcmc_mh_group *g, size_t n_collect, size_t n_discard);
int mmcmc_mh_group_stats_phases(mmcmc_mh_group *g, double *ms3);
int mmcmc_mh_group_state(mmcmc_mh_group *g, void *out);
int mmcmc_mh_group_split_rhat_mean_ess(mmcmc_mh_group *g, float *rhat, float *ess, int *used_rccl);
int mmcmc_mh_group_sync(mmcmc_mh_group *g);
int mmcmc_mh_group_stream_timer(mmcmc_mh_group *g, int stop, float *ms_per_device);
int mmcmc_mh_group_exchange(mmcmc_mh_group *g, int *status, int *n_rccl_ranks);
int mmcmc_mh_group_destroy(mmcmc_mh_group *g);
/* NUTS::run / run_progress (nuts.rs:163-170, 194-338) of every chain on N devices: `init` host [n_chains, dim] doubles
 * and `mode` as in mmcmc_nuts_create; out_host [n_chains, n_collect, dim] of the mode's tensor type (f32 for modes 0 and
 * 1, f64 for mode 2) or NULL; progress as in mmcmc_nuts_run.  Chain i's results do not depend on the number of devices. */
typedef struct mmcmc_nuts_group mmcmc_nuts_group;
int mmcmc_nuts_group_create(mmcmc_nuts_group **out, const mmcmc_target_desc *target, const double *init, size_t n_chains,
                            double target_accept_p, int mode, const int *devices, int n_devices);
int mmcmc_nuts_group_seed(mmcmc_nuts_group *g, uint64_t seed);
int mmcmc_nuts_group_set_chain_offset(mmcmc_nuts_group *g, uint64_t first_global_chain);
int mmcmc_nuts_group_set_max_depth(mmcmc_nuts_group *g, int max_depth);
int mmcmc_nuts_group_run(mmcmc_nuts_group *g, size_t n_collect, size_t n_discard, void *out_host, int progress);
int mmcmc_nuts_group_stats_phases(mmcmc_nuts_group *g, double *ms3);
int mmcmc_nuts_group_state(mmcmc_nuts_group *g, void *out); /* host [n_chains, dim] of the tensor type */
int mmcmc_nuts_group_leapfrog_counts(mmcmc_nuts_group *g, uint64_t *out); /* host [n_chains] */
int mmcmc_nuts_group_split_rhat_mean_ess(mmcmc_nuts_group *g, float *rhat, float *ess, int *used_rccl);
int mmcmc_nuts_group_sync(mmcmc_nuts_group *g);
int mmcmc_nuts_group_stream_timer(mmcmc_nuts_group *g, int stop, float *ms_per_device);
int mmcmc_nuts_group_exchange(mmcmc_nuts_group *g, int *status, int *n_rccl_ranks);
int mmcmc_nuts_group_destroy(mmcmc_nuts_group *g);

/* ---- diagnostics: stats.rs ------------------------------------------------------------------------------
 * split_rhat_mean_ess(sample[chains, n, params]) -> (rhat[params], ess[params])   stats.rs:416-423
 * (splitcat :396-402, withinvar :429-477, rhat :425-427 = sqrt(W/var+) as the reference defines it, ess :496-546).
 * sample: [n_chains, n, dim] of dtype, device or host memory; rhat, ess: host [dim] (f32 like the reference). */
int mmcmc_split_rhat_mean_ess(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n,
                              size_t dim, float *rhat, float *ess, int device, void *stream);
/* The two halves of the above, for multi-GPU runs (chains sharded over ranks):
 * partials: device-side sufficient statistics of the LOCAL chains -- means, ssq: device [2*n_chains, dim] in splitcat
 * order (first halves of all local chains, then second halves), acov_sum: device [n/2, dim] un-normalised lag sums
 * over the local half-chains.  All-gather the first two, all-reduce(sum) the third, then
 * finish (host): the arithmetic of stats.rs:449-465, :425-427, :509-545 on the global statistics. */
int mmcmc_stats_partials(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, float *means,
                         float *ssq, float *acov_sum, int device, void *stream);
int mmcmc_stats_finish(const float *means, const float *ssq, const float *acov_sum, size_t n_half_chains, size_t m,
                       size_t dim, float *rhat, float *ess);
/* finish from the cross-chain SUMS instead of the per-half-chain statistics, for runs where gathering 2 * chains * dim
 * numbers on every rank is the expensive part: dsum[d] = sum over ALL half-chains of (mean - overall mean)^2,
 * wsum[d] = sum of the biased half-chain variances ssq / m (two all-reduces: the overall mean first), acov_sum as
 * above.  Same arithmetic from there on (stats.rs:459-465, :425-427, :509-545). */
int mmcmc_stats_finish_sums(const double *dsum, const double *wsum, const float *acov_sum, size_t n_half_chains, size_t m,
                            size_t dim, float *rhat, float *ess);
/* Which kernel computes the lag sums (process-wide; every one of them implements the same biased, mean-removed
 * estimator of stats.rs:548-654 and they agree to f32 rounding -- tests/test_gpu_parity.py::test_stats_kernels_agree):
 *   AUTO    what the reference does (stats.rs:549): direct sums for half-chains up to 100 draws (register tiles on the
 *           vector ALU), the power spectrum above (one wave-level FFT per chain and parameter, one inverse for all);
 *   FFT / TILE1 / TILE / MFMA / DIRECT   that kernel wherever its shape limits allow, AUTO's choice elsewhere
 *           (FFT: 2 <= n/2 <= 131072 -- one wave-level transform up to 1024, N1 = ceil(n/2 / 1024) = 2 .. 128 residues of
 *           2048-point transforms beyond (any count, not a power of two), one residue and parameter per wave with its spectrum in registers (version 101; 100 ended
 *           at 16384); TILE1: n/2 <= 512; TILE: up to 8 tiles per lane; MFMA: LDS layout up to 64 KB).  Half-chains longer
 *           than 131072 draws -- and, under DIRECT, any the staging kernels cannot hold in a workgroup's LDS -- are reduced
 *           straight from global memory (O(chains x dim x n^2), meant for a few very long chains): any n < 2^31 with
 *           dim x n/2 < 2^32 and 2 x chains x dim < 2^31 (MMCMC_ERR_SHAPE beyond), and chains x dim x (n/2)^2 within the
 *           direct-work limit (default 2^46, about ten seconds of device time; MMCMC_ERR_UNSUPPORTED beyond;
 *           mmcmc_stats_set_direct_work_limit(0) removes the limit).
 * For measurements and for the agreement test; results never depend on it beyond rounding.  Process-wide and meant to
 * be set once: a call that races with it picks one kernel or the other (its work buffer fits both).  Under one selection
 * R-hat / ESS are bit-reproducible across devices too: no kernel's summation grouping depends on the device. */
#define MMCMC_STATS_KERNEL_AUTO 0
#define MMCMC_STATS_KERNEL_FFT 1
#define MMCMC_STATS_KERNEL_TILE1 2
#define MMCMC_STATS_KERNEL_TILE 3
#define MMCMC_STATS_KERNEL_MFMA 4
#define MMCMC_STATS_KERNEL_DIRECT 5
int mmcmc_stats_set_kernel(int kind);
/* lag products the O(n^2) from-global-memory path may be asked for before it returns MMCMC_ERR_UNSUPPORTED (process-wide;
 * default 2^46; 0 = no limit: the call then runs however long it takes) */
int mmcmc_stats_set_direct_work_limit(uint64_t max_lag_products);
/* basic_stats stats.rs:310-336 (host) and RunStats::from stats.rs:360-371 */
int mmcmc_basic_stats_from(const float *data, size_t len, mmcmc_basic_stats *out);
int mmcmc_run_stats_from(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n, size_t dim,
                    mmcmc_run_stats *out, int device, void *stream);

/* ---- running diagnostics: MultiChainTracker (stats.rs:189-306) -------------------------------------------------
 * What run_progress shows while sampling (hmc.rs:242-281: `p(accept)` and `max(rhat)` after every step).  The tracker
 * keeps per-chain running mean / mean of squares (the reference's f32 recurrences, exactly), the exponentially
 * averaged acceptance indicator and the running R-hat sqrt(var+ / W) (the conventional one, unlike split_rhat's,
 * quirk Q7) on the device; it is fed blocks of consecutive states and polled by the host between launches.
 *   mmcmc_tracker_steps: MultiChainTracker::step (stats.rs:236-262) for rows t0 .. t0+k-1 of every chain of
 *       states [n_chains, n_rows, dim] of dtype (device or host pointer) -- e.g. a slice of a sampler's output, or the
 *       current positions with n_rows = 1.
 *   mmcmc_tracker_stats: rhat [dim] (MultiChainTracker::rhat :280-286), its maximum (max_rhat :270-274) and p_accept;
 *       needs at least two steps.  Any of the outputs may be NULL. */
typedef struct mmcmc_tracker mmcmc_tracker;
int mmcmc_tracker_create(mmcmc_tracker **out, size_t n_chains, size_t dim, int device);
int mmcmc_tracker_steps(mmcmc_tracker *h, const void *states, int states_is_device, int dtype, size_t n_rows, size_t t0,
                        size_t k, void *stream);
int mmcmc_tracker_stats(mmcmc_tracker *h, float *rhat, float *max_rhat, float *p_accept, void *stream);
/* The per-chain flavour the generic run_progress uses (core.rs:90-140, 255-300): ChainTracker::new(initial_state) =
 * mmcmc_tracker_init_last (before any step; states [n_chains, dim]), then mmcmc_tracker_steps as above;
 * mmcmc_tracker_chain_stats = collect_rhat (stats.rs:150-178, with its chains*params - 1 divisor, quirk Q9), its
 * NaN-skipping maximum, and the average of the per-chain acceptance EMAs (stats.rs:109-123, started from the first
 * comparison of coordinate 0: quirk Q12). */
int mmcmc_tracker_init_last(mmcmc_tracker *h, const void *states, int states_is_device, int dtype, void *stream);
int mmcmc_tracker_chain_stats(mmcmc_tracker *h, float *rhat, float *max_rhat, float *avg_p_accept, void *stream);
int mmcmc_tracker_n(mmcmc_tracker *h, uint64_t *n);
int mmcmc_tracker_shape(mmcmc_tracker *h, size_t *n_chains, size_t *dim, int *device);
/* withinvar_from_cs (stats.rs:155-178) over the per-chain ChainStats (mean, sm2 = (mean_sq - mean^2) n / (n - 1), n):
 * within [dim], var [dim] (host) -- the pair collect_rhat and ess_from_chainstats share.  Needs at least two steps. */
int mmcmc_tracker_within_var(mmcmc_tracker *h, float *within, float *var, void *stream);
/* stats::ess_from_chainstats(sample, chain_stats) (stats.rs:668-671): the UN-split ESS of sample [n_chains, n, dim]
 * (device or host memory, dtype) with within / var from the per-chain trackers that were fed the run
 * (mmcmc_tracker_init_last + mmcmc_tracker_steps, or the tracker mmcmc_*_run_progress hands back); ess: host [dim].
 * The autocovariance follows the reference's switch on n (stats.rs:549), Geyer's truncation as stats.rs:523-545. */
int mmcmc_ess_from_chainstats(const void *sample, int sample_is_device, int dtype, size_t n_chains, size_t n, size_t dim,
                              mmcmc_tracker *tracker, float *ess, int device, void *stream);

/* ---- run_progress ------------------------------------------------------------------------------------------------
 * ChainRunner::run_progress (core.rs:208-360, the MH flavour), HMC::run_progress (hmc.rs:222-294), NUTS::run_progress
 * (nuts.rs:172-345 over NUTSChain::run_progress :473-526): burn-in + sampling with the reference's running diagnostics,
 * returning the sample and RunStats.  The tracker feeding orders live here once (csrc/mm_progress.hip):
 *   HMC  burn-in unobserved; one MultiChainTracker fed the positions sampling starts from, then every collected state;
 *   MH   per-chain ChainTrackers built on the state before the first step, fed EVERY state, burn-in included;
 *   NUTS as MH (all n_discard + n_collect states; the run is one launch, the trackers are fed after it).
 * cb (may be NULL) is called after every `every` transitions (0: ten times per run) with what the reference's bar shows:
 * p(accept) and max(rhat) -- the MultiChainTracker's for HMC (stats.rs:270-286), the average per-chain EMA and
 * collect_rhat's NaN-skipping maximum for MH and NUTS (core.rs:268-293).
 * out: [n_chains, n_collect, dim] of the sampler's dtype (NUTS: f32 for modes 0 / 1, f64 for mode 2), device or host, may
 * be NULL; stats (may be NULL; needs n_collect >= 2): RunStats::from(sample); tracker_out (may be NULL): receives the
 * tracker that was fed (the caller destroys it with mmcmc_tracker_destroy), e.g. for mmcmc_ess_from_chainstats.
 * stream NULL: a stream of the call's own; the call returns when everything has finished.  Device memory beside the sample:
 * MH a block of `every` burn-in states, NUTS all n_discard + n_collect states of every chain (its adaptation makes the
 * run one launch): n_chains x (n_discard + n_collect) x dim elements -- hipErrorOutOfMemory if that does not fit. */
typedef void (*mmcmc_progress_fn)(void *user, uint64_t transitions_done, uint64_t transitions_total, float p_accept,
                                  float max_rhat);
int mmcmc_mh_run_progress(mmcmc_mh *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                          void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream);
int mmcmc_hmc_run_progress(mmcmc_hmc *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                           void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream);
int mmcmc_nuts_run_progress(mmcmc_nuts *h, size_t n_collect, size_t n_discard, size_t every, mmcmc_progress_fn cb, void *user,
                            void *out, int out_is_device, mmcmc_run_stats *stats, mmcmc_tracker **tracker_out, void *stream);
int mmcmc_tracker_destroy(mmcmc_tracker *h);

/* ---- Metropolis-Hastings on integer states ---------------------------------------------------------------------
 * MHMarkovChain / MetropolisHastings are generic over the state type (metropolis_hastings.rs:28-48, 149-160); the
 * reference's own discrete users are the targets + proposals below.  One chain per lane, state i32, log-densities
 * f64 (a table per target, built with the reference's operation order), step = metropolis_hastings.rs:303-315.
 *   MMCMC_POISSON_REFLECT  params {lambda}: PoissonDist + PoissonRandomWalk      tests/metrohast_poisson_test.rs:18-85
 *   MMCMC_BINOMIAL_CLAMP   params {n, p}:   BinomialDist + BinomialRandomWalk    tests/metrohast_poisson_test.rs:150-212
 *                          (n < 1024: the log-density table lives in LDS; larger n: MMCMC_ERR_UNSUPPORTED)
 *   MMCMC_POISSON_NONNEG   params {lambda}: PoissonTarget + NonnegativeProposal  examples/poisson_mh.rs:8-76
 * (lambda <= 256: states above 1023 have probability zero here.)  init: host [n_chains] i32;
 * out: [n_chains, n_collect] i32 (the reference's Array3<i32>[C, n, 1]).  Integer states and accept counts are
 * bit-exact against the host build of the same header (oracle/engine_host.cpp) and against the oracle's independent
 * restatement on the same stream. */
#define MMCMC_POISSON_REFLECT 16
#define MMCMC_BINOMIAL_CLAMP 17
#define MMCMC_POISSON_NONNEG 18
/* An integer-state model of the caller's own (csrc/mm_rtc.hip, csrc/mm_discrete_kernels.h): `impl Target<i32, f64>` +
 * `impl Proposal<i32, f64>` as the reference's own tests write them (tests/metrohast_poisson_test.rs:18-85), as HIP source
 *     struct mmcmc_user_discrete {
 *         MM_HD static double logp(const double *params, int k);                  // Target::unnorm_logp
 *         MM_HD static int sample(const double *params, int current, int step);   // Proposal::sample; step = +1 / -1, p = 1/2 each
 *         MM_HD static double logq(const double *params, int from, int to);       // Proposal::logp = log q(to | from)
 *     };
 * (mm_log, mm_ln_factorial, MM_NEG_INF, ... of the engine's headers in scope).  *kind_out (>= MMCMC_USER_KIND_BASE) is a `kind` for
 * mmcmc_mh_discrete_create, whose `params` (eight doubles) reach the functor unchanged; the transition is
 * MHMarkovChain::step with both q-terms on the built-in models' stream, so a restatement of a built-in model reproduces
 * it bit for bit (tests/test_user_proposal.py). */
int mmcmc_discrete_register_source(const char *name, const char *hip_source, int *kind_out, char *log, size_t log_len);
typedef struct mmcmc_mh_discrete mmcmc_mh_discrete;
int mmcmc_mh_discrete_create(mmcmc_mh_discrete **out, int kind, const double *params, const int32_t *init,
                             size_t n_chains, int device);
/* kernel mapping: 1 = noise waves + transition waves, four waves per SIMD (default; used from 4096 chains on), 0 = one wave per
 * SIMD; bit-identical */
int mmcmc_mh_discrete_set_kernel_variant(mmcmc_mh_discrete *h, int variant);
int mmcmc_mh_discrete_seed(mmcmc_mh_discrete *h, uint64_t seed);                 /* MetropolisHastings::seed :187-193 */
int mmcmc_mh_discrete_set_chain_offset(mmcmc_mh_discrete *h, uint64_t chain_offset);
/* ChainRunner::run core.rs:176-186: n_discard + n_collect transitions, the last n_collect states kept */
int mmcmc_mh_discrete_run(mmcmc_mh_discrete *h, size_t n_collect, size_t n_discard, int32_t *out, int out_is_device,
                          void *stream);
int mmcmc_mh_discrete_state(mmcmc_mh_discrete *h, int32_t *out);                 /* host [n_chains] */
int mmcmc_mh_discrete_accept_counts(mmcmc_mh_discrete *h, uint64_t *out);        /* host [n_chains], since creation */
int mmcmc_mh_discrete_sync(mmcmc_mh_discrete *h);
int mmcmc_mh_discrete_destroy(mmcmc_mh_discrete *h);

/* ---- Gibbs sampling with a built-in full conditional --------------------------------------------------------
 * GibbsSampler / GibbsMarkovChain (gibbs.rs:31-203) take a user-written `Conditional`; the one the reference ships --
 * the two-component Gaussian mixture over the state [x, z] of gibbs.rs:231-285 and examples/mixture_gibbs.rs:24-76 --
 * is built in.  params = {mu0, sigma0, mu1, sigma1, pi0}; init: host [n_chains, 2] f64 (the reference starts from
 * init_det(n, 2)); out: [n_chains, n_collect, 2] f64.  One sweep = x | z then z | x (gibbs.rs:96-101). */
typedef struct mmcmc_gibbs_mixture mmcmc_gibbs_mixture;
int mmcmc_gibbs_mixture_create(mmcmc_gibbs_mixture **out, const double *params, const double *init, size_t n_chains,
                               int device);
int mmcmc_gibbs_mixture_set_kernel_variant(mmcmc_gibbs_mixture *h, int variant); /* as mmcmc_mh_discrete_set_kernel_variant */
int mmcmc_gibbs_mixture_seed(mmcmc_gibbs_mixture *h, uint64_t seed);            /* GibbsSampler::set_seed gibbs.rs:179-187 */
int mmcmc_gibbs_mixture_set_chain_offset(mmcmc_gibbs_mixture *h, uint64_t chain_offset);
int mmcmc_gibbs_mixture_run(mmcmc_gibbs_mixture *h, size_t n_collect, size_t n_discard, double *out, int out_is_device,
                            void *stream);                                      /* ChainRunner::run core.rs:176-186 */
int mmcmc_gibbs_mixture_state(mmcmc_gibbs_mixture *h, double *out);             /* host [n_chains, 2] */
int mmcmc_gibbs_mixture_sync(mmcmc_gibbs_mixture *h);
int mmcmc_gibbs_mixture_destroy(mmcmc_gibbs_mixture *h);

/* ---- sample sink: io/csv.rs:47-69 save_csv -- header `chain,observation,dim_0,...`, one record per (chain,
 *      observation), values in Rust `Display` formatting (shortest round-trip, no exponent).  sample: HOST
 *      [n_chains, n, dim] of dtype. */
int mmcmc_save_csv(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, const char *filename);

/* ---- densities, for parity tests ---------------------------------------------------------------------
 * BatchedGradientTarget::unnorm_logp_batch (distributions.rs:65-76) / unnorm_logp_and_grad (:81-87)
 * x: host [n, dim]; logp: host [n]; grad: host [n, dim] or NULL; all of `dtype` */
int mmcmc_logp_grad_batch(const mmcmc_target_desc *target, int dtype, const void *x, size_t n, void *logp,
                          void *grad, int device);

/* ---- the engine's random stream, for parity tests: noise of (seed, chain, iteration) exactly as the kernels
 *      draw it.  z: host [n_chains, dim] of dtype, u: host [n_chains] of dtype; computed on the device. */
int mmcmc_draw_noise(uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n_chains, int dim, int dtype,
                     void *z, void *u, int device);
/* The noise of the METROPOLIS-HASTINGS sampler: in f32 at dim <= 2 it draws from a stream of its own -- one Philox block per
 * TWO transitions, the accept uniform's low byte from a second block (csrc/mm_rng.h, "paired stream": half the generator
 * work of the step) --; everywhere else this is mmcmc_draw_noise. */
int mmcmc_draw_noise_mh(uint64_t seed, uint64_t chain_offset, uint32_t iteration, size_t n_chains, int dim, int dtype, void *z,
                        void *u, int device);

#ifdef __cplusplus
}
#endif
#endif /* MMCMC_H */
