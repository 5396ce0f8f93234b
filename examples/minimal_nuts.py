"""examples/minimal_nuts.rs: Rosenbrock2D by NUTS, four chains, target acceptance 0.95, run_progress(400, 400)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init
from mini_mcmc_amd.distributions import Rosenbrock2D
from mini_mcmc_amd.nuts import NUTS


def main():
    target = Rosenbrock2D(1.0, 100.0)
    sampler = NUTS(target, init(4, 2, np.float32), 0.95).set_seed(42)
    sample, stats = sampler.run_progress(400, 400)
    print("Sample sample:", list(sample.shape))
    print(stats)
    assert sample.shape == (4, 400, 2)
    return sample, stats


if __name__ == "__main__":
    main()
