//! examples/minimal_hmc.rs of the reference (3-D Rosenbrock, step size 0.032, 10 leapfrog steps, 400 + 50) on the GPU
//! with 65 536 chains: BASELINE.json config 3.
use mini_mcmc_hip::{init_with_seed, split_rhat_mean_ess, GpuHmc, GpuTarget};

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let init: Vec<Vec<f32>> = init_with_seed(65_536, 3, 42).into_iter().map(|r| r.into_iter().map(|v| v as f32).collect()).collect();
    let mut sampler = GpuHmc::<f32>::new(GpuTarget::RosenbrockND, init, 0.032, 10)?.set_seed(42);
    let sample = sampler.run(400, 50)?; // [65536, 400, 3]
    let (rhat, ess) = split_rhat_mean_ess(sample.view())?;
    println!("sample {:?}  split-rhat {:?}  ess {:?}", sample.dim(), rhat, ess);
    Ok(())
}
