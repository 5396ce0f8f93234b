"""examples/minimal_mh.rs: Gaussian2D by Metropolis-Hastings, four chains (BASELINE.json config 1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings


def main():
    target = Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]])
    proposal = IsotropicGaussian(1.0)
    mh = MetropolisHastings(target, proposal, init_det(4, 2))  # four chains, one GPU lane each
    sample = mh.run(1000, 100)  # 1100 steps, the first 100 discarded
    assert sample.shape[0] == 4 and sample.shape[1] == 1000
    return sample


if __name__ == "__main__":
    print("Collected sample with shape:", main().shape)
