"""examples/mixture_gibbs.rs: a two-component Gaussian mixture by Gibbs sampling (x | z, z | x), four chains."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.gibbs import GibbsSampler, MixtureConditional

N_CHAINS, BURNIN, TOTAL_STEPS = 4, 1000, 1100


def main(seed=None):
    seed = int(np.random.SeedSequence().generate_state(1)[0]) if seed is None else seed  # `rng().random()` in the reference
    conditional = MixtureConditional(mu0=-2.0, sigma0=1.0, mu1=3.0, sigma1=1.5, pi0=0.25)
    sampler = GibbsSampler(conditional, init_det(N_CHAINS, 2)).set_seed(seed)
    sample = sampler.run(TOTAL_STEPS, BURNIN)  # as the reference calls it: 1100 collected after 1000 discarded
    pooled = sample.reshape(TOTAL_STEPS * N_CHAINS, 2)
    print(f"Generated {pooled.size} sample")
    row_mean = pooled.mean(axis=0)
    print(f"Mean after burn-in: ({row_mean[0]:.2f}, {row_mean[1]:.2f})")
    return sample


if __name__ == "__main__":
    main()
