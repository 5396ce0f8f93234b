"""examples/minimal_hmc.rs: the 3-D Rosenbrock density by HMC, four chains, eps 0.032, 10 leapfrog steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC


def main():
    sampler = HMC(RosenbrockND(3), init_det(4, 3, np.float32), 0.032, 10)
    sample = sampler.run(400, 50)
    assert sample.shape == (4, 400, 3)
    return sample


if __name__ == "__main__":
    print("Collected sample with shape:", main().shape)
