//! Safe wrappers over the C ABI (`mini-mcmc-hip-sys`) with the call shapes of mini-mcmc 0.8.3:
//!
//! | mini-mcmc | here |
//! |---|---|
//! | `MetropolisHastings::new(target, proposal, init).seed(s).run(n, d)` (metropolis_hastings.rs:149-193, core.rs:176-186) | `GpuMetropolisHastings::new(..).seed(s).run(n, d)` |
//! | `HMC::new(target, init, eps, L).set_seed(s).run(n, d)` / `.step()` (hmc.rs:87-158, 304-377) | `GpuHmc::new(..).set_seed(s).run(n, d)` / `.step()` |
//! | `NUTS::new(target, init, p).set_seed(s).run(n, d)` (nuts.rs:123-170, 347-353) | `GpuNuts::new(..).set_seed(s).run(n, d)` |
//! | `stats::split_rhat_mean_ess(view)` (stats.rs:416-423) | `split_rhat_mean_ess(view)` |
//! | `core::init_det(n, d)` (core.rs:404-409) | `init_det(n, d)` |
//!
//! Targets: user closures cannot run on the GPU, so the samplers take a [`GpuTarget`] -- a description of one of the
//! built-in densities (the reference's `Gaussian2D`, `DiffableGaussian2D`, `IsotropicGaussian`, `Rosenbrock2D`,
//! `RosenbrockND` plus a dense `GaussianND`).  `run` returns `Result<Array3<T>, ShapeError>` like `ChainRunner::run`;
//! every other failure of the engine (no device, unsupported kind) is an [`MmcmcError`].
//!
//! NOT compiled in the build image (no Rust toolchain there); `tests/c/abi_call_sequence.c` performs the same calls.

use mini_mcmc_hip_sys as sys;
use ndarray::{Array1, Array2, Array3, ArrayView3, ErrorKind, ShapeError};
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};
use std::ptr::{null, null_mut};

/// A status of the engine that is not a shape error.
#[derive(Debug, Clone)]
pub struct MmcmcError {
    pub status: i32,
    pub message: String,
}
impl std::fmt::Display for MmcmcError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "mmcmc status {}: {}", self.status, self.message)
    }
}
impl std::error::Error for MmcmcError {}

fn check(status: c_int) -> Result<(), MmcmcError> {
    if status == sys::MMCMC_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::mmcmc_status_string(status)) }.to_string_lossy().into_owned();
    Err(MmcmcError { status, message })
}

/// The loaded `libmmcmc.so` must be at least the version the `-sys` crate was generated from (`MMCMC_VERSION` of
/// include/mmcmc.h: entry points are only ever added between versions, removals bump it).  Every constructor calls this.
pub fn assert_abi() -> Result<(), MmcmcError> {
    let have = unsafe { sys::mmcmc_version() };
    if have >= sys::MMCMC_VERSION {
        return Ok(());
    }
    Err(MmcmcError { status: sys::MMCMC_ERR_UNSUPPORTED, message: format!("libmmcmc.so is version {have}, this crate was generated for {}", sys::MMCMC_VERSION) })
}

/// `ChainRunner::run` can only fail with a `ShapeError` (core.rs:184); the engine's MMCMC_ERR_SHAPE maps onto it, and so
/// does anything else `run` reports (the trait's signature has no other channel).
fn shape_error(status: c_int) -> ShapeError {
    ShapeError::from_kind(if status == sys::MMCMC_ERR_SHAPE { ErrorKind::IncompatibleShape } else { ErrorKind::Unsupported })
}

/// f32 / f64: the element types the kernels exist for.
pub trait GpuFloat: Copy + Default + 'static {
    const DTYPE: c_int;
}
impl GpuFloat for f32 {
    const DTYPE: c_int = sys::MMCMC_F32;
}
impl GpuFloat for f64 {
    const DTYPE: c_int = sys::MMCMC_F64;
}

/// One of the built-in targets (`Target` / `GradientTarget` of distributions.rs:65-108 for the GPU path).
#[derive(Clone, Debug)]
pub enum GpuTarget {
    /// `Gaussian2D { mean, cov }` (distributions.rs:158-206)
    Gaussian2D { mean: [f64; 2], cov: [[f64; 2]; 2] },
    /// `DiffableGaussian2D::new(mean, cov)` (distributions.rs:212-316)
    DiffableGaussian2D { mean: [f64; 2], cov: [[f64; 2]; 2] },
    /// `IsotropicGaussian::new(std)` as a target (distributions.rs:394-402)
    IsotropicGaussian { std: f64 },
    /// `Rosenbrock2D { a, b }` (distributions.rs:490-524)
    Rosenbrock2D { a: f64, b: f64 },
    /// `RosenbrockND {}` (distributions.rs:528-547); the dimension comes from the initial positions
    RosenbrockND,
    /// zero-mean Gaussian with a dense precision matrix, row-major `[dim, dim]` (not in the reference)
    GaussianND { precision: Vec<f64> },
}
impl GpuTarget {
    fn desc(&self, dim: usize) -> sys::mmcmc_target_desc {
        let mut d = sys::mmcmc_target_desc { kind: 0, dim: dim as i32, params: [0.0; 8], matrix: null() };
        match self {
            GpuTarget::Gaussian2D { mean, cov } | GpuTarget::DiffableGaussian2D { mean, cov } => {
                d.kind = if matches!(self, GpuTarget::Gaussian2D { .. }) { sys::MMCMC_GAUSSIAN2D } else { sys::MMCMC_DIFFABLE_GAUSSIAN2D };
                d.params[..6].copy_from_slice(&[mean[0], mean[1], cov[0][0], cov[0][1], cov[1][0], cov[1][1]]);
            }
            GpuTarget::IsotropicGaussian { std } => {
                d.kind = sys::MMCMC_ISOTROPIC_GAUSSIAN;
                d.params[0] = *std;
            }
            GpuTarget::Rosenbrock2D { a, b } => {
                d.kind = sys::MMCMC_ROSENBROCK2D;
                d.params[0] = *a;
                d.params[1] = *b;
            }
            GpuTarget::RosenbrockND => d.kind = sys::MMCMC_ROSENBROCK_ND,
            GpuTarget::GaussianND { precision } => {
                d.kind = sys::MMCMC_GAUSSIAN_ND;
                d.matrix = precision.as_ptr(); // copied to the device by *_create
            }
        }
        d
    }
}

fn flatten<T: Copy>(init: &[Vec<T>]) -> (Vec<T>, usize, usize) {
    let n = init.len();
    let d = init.first().map(|r| r.len()).unwrap_or(0);
    let mut flat = Vec::with_capacity(n * d);
    for row in init {
        assert_eq!(row.len(), d, "every chain needs a starting point of the same dimension");
        flat.extend_from_slice(row);
    }
    (flat, n, d)
}

/// `core::init_det(n, d)` (core.rs:404-409): the reference's own seeded stream, bit for bit.
pub fn init_det<T: GpuFloat + From<f32>>(n: usize, d: usize) -> Vec<Vec<f64>> {
    init_with_seed(n, d, 42)
}
/// `core::init_with_seed(n, d, seed)` (core.rs:413-419).
pub fn init_with_seed(n: usize, d: usize, seed: u64) -> Vec<Vec<f64>> {
    let mut flat = vec![0f64; n * d];
    unsafe { sys::mmcmc_init_with_seed(n, d, seed, flat.as_mut_ptr()) };
    flat.chunks(d.max(1)).map(|c| c.to_vec()).collect()
}

/// `MetropolisHastings<T, _, _, IsotropicGaussian<T>>` on the GPU: one chain per lane.
pub struct GpuMetropolisHastings<T: GpuFloat> {
    h: *mut sys::mmcmc_mh,
    n_chains: usize,
    dim: usize,
    /// accept counts of the last `run`, one per chain
    pub accept_counts: Vec<u64>,
    _t: PhantomData<T>,
}
unsafe impl<T: GpuFloat> Send for GpuMetropolisHastings<T> {}
impl<T: GpuFloat> GpuMetropolisHastings<T> {
    /// `MetropolisHastings::new(target, proposal, initial_states)` (metropolis_hastings.rs:149-159);
    /// `proposal_std` = `IsotropicGaussian::new(std)`.
    pub fn new(target: GpuTarget, proposal_std: f64, initial_states: Vec<Vec<T>>) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_states);
        let desc = target.desc(d);
        let prop = sys::mmcmc_proposal_desc { kind: sys::MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN, reserved: 0, std: proposal_std };
        let mut h = null_mut();
        check(unsafe { sys::mmcmc_mh_create(&mut h, &desc, &prop, flat.as_ptr() as *const c_void, n, T::DTYPE, 0) })?;
        Ok(Self { h, n_chains: n, dim: d, accept_counts: vec![0; n], _t: PhantomData })
    }
    /// `.seed(seed)` (metropolis_hastings.rs:187-193)
    pub fn seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_mh_seed(self.h, seed) };
        self
    }
    /// `ChainRunner::run` (core.rs:176-186): `[n_chains, n_collect, dim]`.
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<T>, ShapeError> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let st = unsafe {
            sys::mmcmc_mh_run(self.h, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, 0, self.accept_counts.as_mut_ptr(), null_mut())
        };
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
    /// current states `[n_chains, dim]` (`MarkovChain::current_state` of every chain, core.rs:39-45)
    pub fn current_states(&self) -> Result<Array2<T>, MmcmcError> {
        let mut out = Array2::<T>::default((self.n_chains, self.dim));
        check(unsafe { sys::mmcmc_mh_state(self.h, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
}
impl<T: GpuFloat> Drop for GpuMetropolisHastings<T> {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_mh_destroy(self.h) };
    }
}

/// `HMC<T, B, GTarget>` on the GPU (hmc.rs:44-158): the leapfrog runs in registers, one chain per lane.
pub struct GpuHmc<T: GpuFloat> {
    h: *mut sys::mmcmc_hmc,
    n_chains: usize,
    dim: usize,
    pub accept_counts: Vec<u64>,
    _t: PhantomData<T>,
}
unsafe impl<T: GpuFloat> Send for GpuHmc<T> {}
impl<T: GpuFloat> GpuHmc<T> {
    /// `HMC::new(target, initial_positions, step_size, n_leapfrog)` (hmc.rs:87-109)
    pub fn new(target: GpuTarget, initial_positions: Vec<Vec<T>>, step_size: f64, n_leapfrog: usize) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_positions);
        let desc = target.desc(d);
        let mut h = null_mut();
        check(unsafe { sys::mmcmc_hmc_create(&mut h, &desc, flat.as_ptr() as *const c_void, n, step_size, n_leapfrog as c_int, T::DTYPE, 0) })?;
        Ok(Self { h, n_chains: n, dim: d, accept_counts: vec![0; n], _t: PhantomData })
    }
    /// `.set_seed(seed)` (hmc.rs:118-121; inert in the reference -- quirk Q6 -- effective here)
    pub fn set_seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_hmc_seed(self.h, seed) };
        self
    }
    /// `HMC::run(n_collect, n_discard)` (hmc.rs:137-158): `[n_chains, n_collect, dim]`
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<T>, ShapeError> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let st = unsafe {
            sys::mmcmc_hmc_run(self.h, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, 0, self.accept_counts.as_mut_ptr(), null_mut())
        };
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
    /// `HMC::step()` (hmc.rs:304-377): one transition of every chain
    pub fn step(&mut self) -> Result<(), MmcmcError> {
        check(unsafe { sys::mmcmc_hmc_step(self.h, null_mut()) })?;
        check(unsafe { sys::mmcmc_hmc_sync(self.h) })
    }
    /// `positions` (hmc.rs:49): `[n_chains, dim]`
    pub fn positions(&self) -> Result<Array2<T>, MmcmcError> {
        let mut out = Array2::<T>::default((self.n_chains, self.dim));
        check(unsafe { sys::mmcmc_hmc_state(self.h, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
}
impl<T: GpuFloat> Drop for GpuHmc<T> {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_hmc_destroy(self.h) };
    }
}

/// `NUTS<T, B, GTarget>` on the GPU (nuts.rs:100-353).  `T = f64`: f32 tensors + f64 scalars like the reference's
/// `NUTS<f64, Autodiff<NdArray>, _>` (samples come back as f32); `T = f32`: f32 / f32.
pub struct GpuNuts {
    h: *mut sys::mmcmc_nuts,
    n_chains: usize,
    dim: usize,
}
unsafe impl Send for GpuNuts {}
impl GpuNuts {
    /// `NUTS::new(target, initial_positions, target_accept_p)` (nuts.rs:123-129); `scalars_f64` picks the reference's `T`.
    pub fn new(target: GpuTarget, initial_positions: Vec<Vec<f64>>, target_accept_p: f64, scalars_f64: bool) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_positions);
        let desc = target.desc(d);
        let mut h = null_mut();
        check(unsafe { sys::mmcmc_nuts_create(&mut h, &desc, flat.as_ptr(), n, target_accept_p, if scalars_f64 { 0 } else { 1 }, 0) })?;
        Ok(Self { h, n_chains: n, dim: d })
    }
    /// `.set_seed(seed)` (nuts.rs:347-353)
    pub fn set_seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_nuts_seed(self.h, seed) };
        self
    }
    /// `NUTS::run(n_collect, n_discard)` (nuts.rs:163-170): N - 1 transitions, row 0 may be the initial point (quirk Q11)
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<f32>, ShapeError> {
        let mut out = Array3::<f32>::default((self.n_chains, n_collect, self.dim));
        let mut st = unsafe { sys::mmcmc_nuts_run(self.h, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, 0, 0, null_mut()) };
        if st == sys::MMCMC_OK {
            st = unsafe { sys::mmcmc_nuts_sync(self.h) };
        }
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
}
impl Drop for GpuNuts {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_nuts_destroy(self.h) };
    }
}

/// The same sampler over several GPUs from one call (`mmcmc_nuts_group_*`): `run` executes every chain, chain i's
/// results do not depend on the number of devices (the stream is keyed by the global chain index).
pub struct GpuNutsGroup {
    g: *mut sys::mmcmc_nuts_group,
    n_chains: usize,
    dim: usize,
}
unsafe impl Send for GpuNutsGroup {}
impl GpuNutsGroup {
    pub fn new_on(devices: &[i32], target: GpuTarget, initial_positions: Vec<Vec<f64>>, target_accept_p: f64, scalars_f64: bool) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_positions);
        let desc = target.desc(d);
        let mut g = null_mut();
        check(unsafe {
            sys::mmcmc_nuts_group_create(&mut g, &desc, flat.as_ptr(), n, target_accept_p, if scalars_f64 { 0 } else { 1 }, devices.as_ptr(), devices.len() as i32)
        })?;
        Ok(Self { g, n_chains: n, dim: d })
    }
    pub fn set_seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_nuts_group_seed(self.g, seed) };
        self
    }
    /// `NUTS::run(n_collect, n_discard)` of every chain on every device of the group
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<f32>, ShapeError> {
        let mut out = Array3::<f32>::default((self.n_chains, n_collect, self.dim));
        let st = unsafe { sys::mmcmc_nuts_group_run(self.g, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, 0) };
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
    /// `(rhat, ess)` of the last run over the chains of all devices (RCCL inside the library)
    pub fn split_rhat_mean_ess(&mut self) -> Result<(Array1<f32>, Array1<f32>), MmcmcError> {
        let (mut rhat, mut ess) = (Array1::<f32>::zeros(self.dim), Array1::<f32>::zeros(self.dim));
        check(unsafe { sys::mmcmc_nuts_group_split_rhat_mean_ess(self.g, rhat.as_mut_ptr(), ess.as_mut_ptr(), null_mut()) })?;
        Ok((rhat, ess))
    }
}
impl Drop for GpuNutsGroup {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_nuts_group_destroy(self.g) };
    }
}

/// `stats::split_rhat_mean_ess(sample)` (stats.rs:416-423): `(rhat, ess)`, rhat = sqrt(W / var+) as the crate defines it.
pub fn split_rhat_mean_ess(sample: ArrayView3<f32>) -> Result<(Array1<f32>, Array1<f32>), MmcmcError> {
    let (c, n, p) = sample.dim();
    let owned = sample.as_standard_layout();
    let (mut rhat, mut ess) = (Array1::<f32>::zeros(p), Array1::<f32>::zeros(p));
    check(unsafe {
        sys::mmcmc_split_rhat_mean_ess(owned.as_ptr() as *const c_void, 0, sys::MMCMC_F32, c, n, p, rhat.as_mut_ptr(), ess.as_mut_ptr(), 0, null_mut())
    })?;
    Ok((rhat, ess))
}

/// `RunStats` (stats.rs:338-342): summaries of ESS and split R-hat over the parameters.
pub type RunStats = sys::mmcmc_run_stats;

/// `RunStats::from(sample.view())` (stats.rs:360-371)
pub fn run_stats<T: GpuFloat>(sample: &Array3<T>) -> Result<RunStats, MmcmcError> {
    let (c, n, p) = sample.dim();
    let owned = sample.as_standard_layout();
    let mut out = RunStats::default();
    check(unsafe { sys::mmcmc_run_stats_from(owned.as_ptr() as *const c_void, 0, T::DTYPE, c, n, p, &mut out, 0, null_mut()) })?;
    Ok(out)
}

impl<T: GpuFloat> GpuMetropolisHastings<T> {
    /// `ChainRunner::run_progress(n_collect, n_discard)` (core.rs:208-360): the sample and its `RunStats`; the per-chain
    /// `ChainTracker`s (built on the initial states, fed every state, burn-in included) come back as a `GpuTracker`.
    /// The feeding order is the library's (`mmcmc_mh_run_progress`).  The reference's progress bars (one OS thread per
    /// chain feeding indicatif) have no analogue at 65 536 chains.
    pub fn run_progress(&mut self, n_collect: usize, n_discard: usize) -> Result<(Array3<T>, RunStats, GpuTracker), Box<dyn std::error::Error>> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let (mut stats, mut tr) = (RunStats::default(), null_mut());
        check(unsafe { sys::mmcmc_mh_run_progress(self.h, n_collect, n_discard, 0, None, null_mut(), out.as_mut_ptr() as *mut c_void, 0, &mut stats, &mut tr, null_mut()) })?;
        Ok((out, stats, GpuTracker { h: tr, dim: self.dim, n_chains: self.n_chains }))
    }
}
impl<T: GpuFloat> GpuHmc<T> {
    /// `HMC::run_progress(n_collect, n_discard)` (hmc.rs:222-294): `(sample, RunStats)` and the `MultiChainTracker` that
    /// was fed the positions after the burn-in and every collected state (`mmcmc_hmc_run_progress`)
    pub fn run_progress(&mut self, n_collect: usize, n_discard: usize) -> Result<(Array3<T>, RunStats, GpuTracker), Box<dyn std::error::Error>> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let (mut stats, mut tr) = (RunStats::default(), null_mut());
        check(unsafe { sys::mmcmc_hmc_run_progress(self.h, n_collect, n_discard, 0, None, null_mut(), out.as_mut_ptr() as *mut c_void, 0, &mut stats, &mut tr, null_mut()) })?;
        Ok((out, stats, GpuTracker { h: tr, dim: self.dim, n_chains: self.n_chains }))
    }
}
impl GpuNuts {
    /// `NUTS::run_progress(n_collect, n_discard)` (nuts.rs:172-345): all N transitions (`run` performs N - 1),
    /// `(sample, RunStats)` and the per-chain trackers fed the initial position and every state (`mmcmc_nuts_run_progress`)
    pub fn run_progress(&mut self, n_collect: usize, n_discard: usize) -> Result<(Array3<f32>, RunStats, GpuTracker), Box<dyn std::error::Error>> {
        let mut out = Array3::<f32>::default((self.n_chains, n_collect, self.dim));
        let (mut stats, mut tr) = (RunStats::default(), null_mut());
        check(unsafe { sys::mmcmc_nuts_run_progress(self.h, n_collect, n_discard, 0, None, null_mut(), out.as_mut_ptr() as *mut c_void, 0, &mut stats, &mut tr, null_mut()) })?;
        Ok((out, stats, GpuTracker { h: tr, dim: self.dim, n_chains: self.n_chains }))
    }
}

/// `stats::ess_from_chainstats(sample, chain_stats)` (stats.rs:668-671): the un-split ESS with within / var from the
/// per-chain trackers `tracker` (as returned by a `run_progress`)
pub fn ess_from_chainstats(sample: ArrayView3<f32>, tracker: &GpuTracker) -> Result<Array1<f32>, MmcmcError> {
    let (c, n, p) = sample.dim();
    let owned = sample.as_standard_layout();
    let mut ess = Array1::<f32>::zeros(p);
    check(unsafe { sys::mmcmc_ess_from_chainstats(owned.as_ptr() as *const c_void, 0, sys::MMCMC_F32, c, n, p, tracker.h, ess.as_mut_ptr(), 0, null_mut()) })?;
    Ok(ess)
}

/// `MultiChainTracker` (stats.rs:189-306) on the GPU: running R-hat and the acceptance EMA of all chains.
pub struct GpuTracker {
    h: *mut sys::mmcmc_tracker,
    dim: usize,
    n_chains: usize,
}
unsafe impl Send for GpuTracker {}
impl GpuTracker {
    /// `MultiChainTracker::new(n_chains, n_params)` (stats.rs:207-226)
    pub fn new(n_chains: usize, n_params: usize) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let mut h = null_mut();
        check(unsafe { sys::mmcmc_tracker_create(&mut h, n_chains, n_params, 0) })?;
        Ok(Self { h, dim: n_params, n_chains })
    }
    /// `MultiChainTracker::step` (stats.rs:236-262) for every row of `states [n_chains, k, dim]`
    pub fn steps<T: GpuFloat>(&mut self, states: &Array3<T>) -> Result<(), MmcmcError> {
        let (c, k, d) = states.dim();
        assert!(c == self.n_chains && d == self.dim);
        let owned = states.as_standard_layout();
        check(unsafe { sys::mmcmc_tracker_steps(self.h, owned.as_ptr() as *const c_void, 0, T::DTYPE, k, 0, k, null_mut()) })
    }
    /// `(rhat per parameter, max_rhat, p_accept)` (stats.rs:264-286)
    pub fn stats(&mut self) -> Result<(Array1<f32>, f32, f32), MmcmcError> {
        let mut rhat = Array1::<f32>::zeros(self.dim);
        let (mut mx, mut pa) = (0f32, 0f32);
        check(unsafe { sys::mmcmc_tracker_stats(self.h, rhat.as_mut_ptr(), &mut mx, &mut pa, null_mut()) })?;
        Ok((rhat, mx, pa))
    }
}
impl Drop for GpuTracker {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_tracker_destroy(self.h) };
    }
}

/// How a device group exchanged its diagnostics (`*used_rccl` of `mmcmc_*_group_split_rhat_mean_ess`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum GroupExchange {
    Rccl,
    HostByDesign,
    HostFallbackNoLibrary,
    HostFallbackInitFailed,
}
fn exchange_of(code: c_int) -> GroupExchange {
    match code {
        1 => GroupExchange::Rccl,
        0 => GroupExchange::HostByDesign,
        -1 => GroupExchange::HostFallbackNoLibrary,
        _ => GroupExchange::HostFallbackInitFailed,
    }
}

/// `HMC` over several GPUs from one call (`mmcmc_hmc_group_*`): contiguous shards, the stream keyed by the global chain index.
pub struct GpuHmcGroup<T: GpuFloat> {
    g: *mut sys::mmcmc_hmc_group,
    n_chains: usize,
    dim: usize,
    pub accept_counts: Vec<u64>,
    _t: PhantomData<T>,
}
unsafe impl<T: GpuFloat> Send for GpuHmcGroup<T> {}
impl<T: GpuFloat> GpuHmcGroup<T> {
    pub fn new_on(devices: &[i32], target: GpuTarget, initial_positions: Vec<Vec<T>>, step_size: f64, n_leapfrog: usize) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_positions);
        let desc = target.desc(d);
        let mut g = null_mut();
        check(unsafe {
            sys::mmcmc_hmc_group_create(&mut g, &desc, flat.as_ptr() as *const c_void, n, step_size, n_leapfrog as c_int, T::DTYPE, devices.as_ptr(), devices.len() as c_int)
        })?;
        Ok(Self { g, n_chains: n, dim: d, accept_counts: vec![0; n], _t: PhantomData })
    }
    pub fn set_seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_hmc_group_seed(self.g, seed) };
        self
    }
    /// `HMC::run(n_collect, n_discard)` of every chain on every device
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<T>, ShapeError> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let st = unsafe { sys::mmcmc_hmc_group_run(self.g, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, self.accept_counts.as_mut_ptr()) };
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
    /// The same run with the sample left on the devices: only enqueued on the shards' streams (returns at once; several in
    /// a row keep every device busy).  `sync` waits; `split_rhat_mean_ess` orders itself behind the queued runs.
    pub fn run_on_devices(&mut self, n_collect: usize, n_discard: usize) -> Result<(), MmcmcError> {
        check(unsafe { sys::mmcmc_hmc_group_run_async(self.g, n_collect, n_discard) })
    }
    pub fn sync(&mut self) -> Result<(), MmcmcError> {
        check(unsafe { sys::mmcmc_hmc_group_sync(self.g) })
    }
    /// How the diagnostics will travel, known from construction on (the RCCL communicators are made there)
    pub fn exchange(&self) -> Result<GroupExchange, MmcmcError> {
        let mut how: c_int = 0;
        check(unsafe { sys::mmcmc_hmc_group_exchange(self.g, &mut how, null_mut()) })?;
        Ok(exchange_of(how))
    }
    /// `(rhat, ess)` of the last run over the chains of all devices, and how the statistics travelled
    pub fn split_rhat_mean_ess(&mut self) -> Result<(Array1<f32>, Array1<f32>, GroupExchange), MmcmcError> {
        let (mut rhat, mut ess) = (Array1::<f32>::zeros(self.dim), Array1::<f32>::zeros(self.dim));
        let mut how: c_int = 0;
        check(unsafe { sys::mmcmc_hmc_group_split_rhat_mean_ess(self.g, rhat.as_mut_ptr(), ess.as_mut_ptr(), &mut how) })?;
        Ok((rhat, ess, exchange_of(how)))
    }
}
impl<T: GpuFloat> Drop for GpuHmcGroup<T> {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_hmc_group_destroy(self.g) };
    }
}

/// `MetropolisHastings` over several GPUs from one call (`mmcmc_mh_group_*`).
pub struct GpuMhGroup<T: GpuFloat> {
    g: *mut sys::mmcmc_mh_group,
    n_chains: usize,
    dim: usize,
    pub accept_counts: Vec<u64>,
    _t: PhantomData<T>,
}
unsafe impl<T: GpuFloat> Send for GpuMhGroup<T> {}
impl<T: GpuFloat> GpuMhGroup<T> {
    pub fn new_on(devices: &[i32], target: GpuTarget, proposal_std: f64, initial_states: Vec<Vec<T>>) -> Result<Self, MmcmcError> {
        assert_abi()?;
        let (flat, n, d) = flatten(&initial_states);
        let desc = target.desc(d);
        let prop = sys::mmcmc_proposal_desc { kind: sys::MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN, reserved: 0, std: proposal_std };
        let mut g = null_mut();
        check(unsafe {
            sys::mmcmc_mh_group_create(&mut g, &desc, &prop, flat.as_ptr() as *const c_void, n, T::DTYPE, devices.as_ptr(), devices.len() as c_int)
        })?;
        Ok(Self { g, n_chains: n, dim: d, accept_counts: vec![0; n], _t: PhantomData })
    }
    pub fn seed(self, seed: u64) -> Self {
        unsafe { sys::mmcmc_mh_group_seed(self.g, seed) };
        self
    }
    /// `ChainRunner::run` of every chain on every device
    pub fn run(&mut self, n_collect: usize, n_discard: usize) -> Result<Array3<T>, ShapeError> {
        let mut out = Array3::<T>::default((self.n_chains, n_collect, self.dim));
        let st = unsafe { sys::mmcmc_mh_group_run(self.g, n_collect, n_discard, out.as_mut_ptr() as *mut c_void, self.accept_counts.as_mut_ptr()) };
        if st == sys::MMCMC_OK { Ok(out) } else { Err(shape_error(st)) }
    }
    pub fn split_rhat_mean_ess(&mut self) -> Result<(Array1<f32>, Array1<f32>, GroupExchange), MmcmcError> {
        let (mut rhat, mut ess) = (Array1::<f32>::zeros(self.dim), Array1::<f32>::zeros(self.dim));
        let mut how: c_int = 0;
        check(unsafe { sys::mmcmc_mh_group_split_rhat_mean_ess(self.g, rhat.as_mut_ptr(), ess.as_mut_ptr(), &mut how) })?;
        Ok((rhat, ess, exchange_of(how)))
    }
}
impl<T: GpuFloat> Drop for GpuMhGroup<T> {
    fn drop(&mut self) {
        unsafe { sys::mmcmc_mh_group_destroy(self.g) };
    }
}

/// A target or proposal written as HIP source (`mmcmc_target_register_source` / `mmcmc_proposal_register_source`): the
/// GPU analogue of implementing `Target` / `GradientTarget` / `Proposal` (distributions.rs:65-108) oneself.  Returns the
/// kind to put into a target / proposal description, or the compiler's log.
pub fn register_target_source(name: &str, dim: usize, hip_source: &str) -> Result<i32, String> {
    let (n, s) = (std::ffi::CString::new(name).unwrap(), std::ffi::CString::new(hip_source).unwrap());
    let mut kind: c_int = 0;
    let mut log = vec![0 as std::os::raw::c_char; 1 << 16];
    let st = unsafe { sys::mmcmc_target_register_source(n.as_ptr(), dim as c_int, s.as_ptr(), &mut kind, log.as_mut_ptr(), log.len()) };
    if st == sys::MMCMC_OK { Ok(kind) } else { Err(unsafe { CStr::from_ptr(log.as_ptr()) }.to_string_lossy().into_owned()) }
}
pub fn register_proposal_source(name: &str, target_kind: i32, dim: usize, hip_source: &str) -> Result<i32, String> {
    let (n, s) = (std::ffi::CString::new(name).unwrap(), std::ffi::CString::new(hip_source).unwrap());
    let mut kind: c_int = 0;
    let mut log = vec![0 as std::os::raw::c_char; 1 << 16];
    let st = unsafe { sys::mmcmc_proposal_register_source(n.as_ptr(), target_kind, dim as c_int, s.as_ptr(), &mut kind, log.as_mut_ptr(), log.len()) };
    if st == sys::MMCMC_OK { Ok(kind) } else { Err(unsafe { CStr::from_ptr(log.as_ptr()) }.to_string_lossy().into_owned()) }
}
