"""examples/rosenbrock_mh.rs: the 2-D Rosenbrock density (a = 1, b = 100) by MH, seed 42, four chains, run_progress."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.distributions import IsotropicGaussian, Rosenbrock2D
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

SAMPLE_SIZE, BURNIN, N_CHAINS = 5000, 1000, 4


def main():
    mh = MetropolisHastings(Rosenbrock2D(1.0, 100.0), IsotropicGaussian(1.0), init_det(N_CHAINS, 2)).seed(42)
    sample, stats = mh.run_progress(SAMPLE_SIZE // N_CHAINS, BURNIN)
    print(stats)
    pooled = sample.reshape(SAMPLE_SIZE, 2)
    print(f"Generated {pooled.shape[0]} sample")
    row_mean = pooled.mean(axis=0)
    print(f"Mean after burn-in: ({row_mean[0]:.2f}, {row_mean[1]:.2f})")
    return sample, stats


if __name__ == "__main__":
    main()
