//! Raw declarations of the C ABI in `include/mmcmc.h` (one `extern "C"` item per header entry the safe wrappers use;
//! the comment on each names the reference item it stands in for).  Status convention: 0 ok, < 0 `MMCMC_ERR_*`,
//! > 0 a HIP error code; no panics cross the boundary.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mmcmc_target_desc {
    pub kind: i32,
    pub dim: i32,
    pub params: [c_double; 8],
    pub matrix: *const c_double,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mmcmc_proposal_desc {
    pub kind: i32,
    pub reserved: i32,
    pub std: c_double,
}
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct mmcmc_basic_stats {
    pub min: f32,
    pub median: f32,
    pub max: f32,
    pub mean: f32,
    pub std: f32,
}
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct mmcmc_run_stats {
    pub ess: mmcmc_basic_stats,
    pub rhat: mmcmc_basic_stats,
}
#[repr(C)]
pub struct mmcmc_mh {
    _p: [u8; 0],
}
#[repr(C)]
pub struct mmcmc_hmc {
    _p: [u8; 0],
}
#[repr(C)]
pub struct mmcmc_nuts {
    _p: [u8; 0],
}
#[repr(C)]
pub struct mmcmc_hmc_group {
    _p: [u8; 0],
}
#[repr(C)]
pub struct mmcmc_nuts_group {
    _p: [u8; 0],
}

pub const MMCMC_OK: c_int = 0;
pub const MMCMC_ERR_INVALID_ARG: c_int = -1;
pub const MMCMC_ERR_UNSUPPORTED: c_int = -2;
pub const MMCMC_ERR_SHAPE: c_int = -3;
pub const MMCMC_ERR_NO_DEVICE: c_int = -4;
pub const MMCMC_ERR_STATE: c_int = -5;
pub const MMCMC_F32: c_int = 0;
pub const MMCMC_F64: c_int = 1;
pub const MMCMC_GAUSSIAN2D: i32 = 0;
pub const MMCMC_DIFFABLE_GAUSSIAN2D: i32 = 1;
pub const MMCMC_ISOTROPIC_GAUSSIAN: i32 = 2;
pub const MMCMC_ROSENBROCK2D: i32 = 3;
pub const MMCMC_ROSENBROCK_ND: i32 = 4;
pub const MMCMC_STANDARD_NORMAL: i32 = 5;
pub const MMCMC_GAUSSIAN_ND: i32 = 6;
pub const MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN: i32 = 0;

extern "C" {
    pub fn mmcmc_version() -> c_int;
    pub fn mmcmc_status_string(status: c_int) -> *const c_char;
    pub fn mmcmc_device_count(count: *mut c_int) -> c_int;
    /// core::init_with_seed / init_det (core.rs:404-435): the reference's own stream.
    pub fn mmcmc_init_with_seed(n: usize, d: usize, seed: u64, out: *mut c_double) -> c_int;

    /// MetropolisHastings::new (metropolis_hastings.rs:149-159)
    pub fn mmcmc_mh_create(
        out: *mut *mut mmcmc_mh,
        target: *const mmcmc_target_desc,
        proposal: *const mmcmc_proposal_desc,
        init: *const c_void,
        n_chains: usize,
        dtype: c_int,
        device: c_int,
    ) -> c_int;
    /// MetropolisHastings::seed (metropolis_hastings.rs:187-193)
    pub fn mmcmc_mh_seed(h: *mut mmcmc_mh, seed: u64) -> c_int;
    pub fn mmcmc_mh_set_chain_offset(h: *mut mmcmc_mh, first_global_chain: u64) -> c_int;
    /// ChainRunner::run (core.rs:176-186)
    pub fn mmcmc_mh_run(
        h: *mut mmcmc_mh,
        n_collect: usize,
        n_discard: usize,
        out: *mut c_void,
        out_is_device: c_int,
        accept_counts: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn mmcmc_mh_state(h: *mut mmcmc_mh, out: *mut c_void) -> c_int;
    pub fn mmcmc_mh_destroy(h: *mut mmcmc_mh) -> c_int;

    /// HMC::new (hmc.rs:87-109)
    pub fn mmcmc_hmc_create(
        out: *mut *mut mmcmc_hmc,
        target: *const mmcmc_target_desc,
        init: *const c_void,
        n_chains: usize,
        step_size: c_double,
        n_leapfrog: c_int,
        dtype: c_int,
        device: c_int,
    ) -> c_int;
    /// HMC::set_seed (hmc.rs:118-121)
    pub fn mmcmc_hmc_seed(h: *mut mmcmc_hmc, seed: u64) -> c_int;
    pub fn mmcmc_hmc_set_chain_offset(h: *mut mmcmc_hmc, first_global_chain: u64) -> c_int;
    /// HMC::run (hmc.rs:137-158)
    pub fn mmcmc_hmc_run(
        h: *mut mmcmc_hmc,
        n_collect: usize,
        n_discard: usize,
        out: *mut c_void,
        out_is_device: c_int,
        accept_counts: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    /// HMC::step (hmc.rs:304-377)
    pub fn mmcmc_hmc_step(h: *mut mmcmc_hmc, stream: *mut c_void) -> c_int;
    pub fn mmcmc_hmc_sync(h: *mut mmcmc_hmc) -> c_int;
    /// HMC::positions (hmc.rs:49)
    pub fn mmcmc_hmc_state(h: *mut mmcmc_hmc, out: *mut c_void) -> c_int;
    pub fn mmcmc_hmc_destroy(h: *mut mmcmc_hmc) -> c_int;

    /// NUTS::new (nuts.rs:123-129); mode 0 = f32 tensors + f64 scalars, 1 = f32 / f32, 2 = f64 / f64
    pub fn mmcmc_nuts_create(
        out: *mut *mut mmcmc_nuts,
        target: *const mmcmc_target_desc,
        init: *const c_double,
        n_chains: usize,
        target_accept_p: c_double,
        mode: c_int,
        device: c_int,
    ) -> c_int;
    /// NUTS::set_seed (nuts.rs:347-353)
    pub fn mmcmc_nuts_seed(h: *mut mmcmc_nuts, seed: u64) -> c_int;
    /// not in the reference (its `while s` is unbounded): doublings per transition, default 10, at most 12
    pub fn mmcmc_nuts_set_max_depth(h: *mut mmcmc_nuts, max_depth: c_int) -> c_int;
    /// NUTS::run (nuts.rs:163-170) with progress = 0, NUTS::run_progress (nuts.rs:194-338) with progress = 1
    pub fn mmcmc_nuts_run(
        h: *mut mmcmc_nuts,
        n_collect: usize,
        n_discard: usize,
        out: *mut c_void,
        out_is_device: c_int,
        progress: c_int,
        stream: *mut c_void,
    ) -> c_int;
    pub fn mmcmc_nuts_sync(h: *mut mmcmc_nuts) -> c_int;
    pub fn mmcmc_nuts_destroy(h: *mut mmcmc_nuts) -> c_int;

    /// one call for every chain on several GPUs (core.rs:176-186 "run executes all chains")
    pub fn mmcmc_hmc_group_create(
        out: *mut *mut mmcmc_hmc_group,
        target: *const mmcmc_target_desc,
        init: *const c_void,
        n_chains: usize,
        step_size: c_double,
        n_leapfrog: c_int,
        dtype: c_int,
        devices: *const c_int,
        n_devices: c_int,
    ) -> c_int;
    pub fn mmcmc_hmc_group_seed(g: *mut mmcmc_hmc_group, seed: u64) -> c_int;
    pub fn mmcmc_hmc_group_run(
        g: *mut mmcmc_hmc_group,
        n_collect: usize,
        n_discard: usize,
        out_host: *mut c_void,
        accept_counts: *mut u64,
    ) -> c_int;
    pub fn mmcmc_hmc_group_split_rhat_mean_ess(g: *mut mmcmc_hmc_group, rhat: *mut f32, ess: *mut f32, used_rccl: *mut c_int) -> c_int;
    pub fn mmcmc_hmc_group_destroy(g: *mut mmcmc_hmc_group) -> c_int;
    /// NUTS::run / run_progress (nuts.rs:163-170, 194-338) of every chain on several GPUs; init: [n_chains, dim] f64
    pub fn mmcmc_nuts_group_create(
        out: *mut *mut mmcmc_nuts_group,
        target: *const mmcmc_target_desc,
        init: *const c_double,
        n_chains: usize,
        target_accept_p: c_double,
        mode: c_int,
        devices: *const c_int,
        n_devices: c_int,
    ) -> c_int;
    pub fn mmcmc_nuts_group_seed(g: *mut mmcmc_nuts_group, seed: u64) -> c_int;
    pub fn mmcmc_nuts_group_run(g: *mut mmcmc_nuts_group, n_collect: usize, n_discard: usize, out_host: *mut c_void, progress: c_int) -> c_int;
    pub fn mmcmc_nuts_group_split_rhat_mean_ess(g: *mut mmcmc_nuts_group, rhat: *mut f32, ess: *mut f32, used_rccl: *mut c_int) -> c_int;
    pub fn mmcmc_nuts_group_destroy(g: *mut mmcmc_nuts_group) -> c_int;

    /// stats::split_rhat_mean_ess (stats.rs:416-423); rhat = sqrt(W / var+), the crate's definition
    pub fn mmcmc_split_rhat_mean_ess(
        sample: *const c_void,
        sample_is_device: c_int,
        dtype: c_int,
        n_chains: usize,
        n: usize,
        dim: usize,
        rhat: *mut f32,
        ess: *mut f32,
        device: c_int,
        stream: *mut c_void,
    ) -> c_int;
    /// RunStats::from (stats.rs:360-371)
    pub fn mmcmc_run_stats_from(
        sample: *const c_void,
        sample_is_device: c_int,
        dtype: c_int,
        n_chains: usize,
        n: usize,
        dim: usize,
        out: *mut mmcmc_run_stats,
        device: c_int,
        stream: *mut c_void,
    ) -> c_int;
    /// GradientTarget::unnorm_logp_and_grad / BatchedGradientTarget::unnorm_logp_batch (distributions.rs:65-88)
    pub fn mmcmc_logp_grad_batch(
        target: *const mmcmc_target_desc,
        dtype: c_int,
        x: *const c_void,
        n: usize,
        logp: *mut c_void,
        grad: *mut c_void,
        device: c_int,
    ) -> c_int;
    /// io::csv::save_csv (io/csv.rs:47-69)
    pub fn mmcmc_save_csv(sample: *const c_void, dtype: c_int, n_chains: usize, n: usize, dim: usize, filename: *const c_char) -> c_int;
}
