"""examples/gauss_mh.rs: a correlated 2-D Gaussian by MH with run_progress, the pooled mean, and the sample saved as CSV
(the reference draws a plotly scatter plot and, with its `parquet` feature, saves a parquet file)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian
from mini_mcmc_amd.io.csv import save_csv
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

SAMPLE_SIZE, BURNIN, N_CHAINS = 5000, 1000, 4


def main(seed=None, out_dir=None):
    seed = int(np.random.SeedSequence().generate_state(1)[0]) if seed is None else seed  # `rng().random()` in the reference
    target = Gaussian2D([0.0, 0.0], [[2.0, 1.0], [1.0, 2.0]])
    mh = MetropolisHastings(target, IsotropicGaussian(2.0), init_det(N_CHAINS, 2)).seed(seed)
    sample, stats = mh.run_progress(SAMPLE_SIZE // N_CHAINS, BURNIN)
    pooled = sample.reshape(SAMPLE_SIZE, 2)
    print(f"Generated {pooled.shape[0]} sample\n{stats}")
    row_mean = pooled.mean(axis=0)
    print(f"Mean after burn-in: ({row_mean[0]:.2f}, {row_mean[1]:.2f})")
    path = os.path.join(out_dir or tempfile.gettempdir(), "sample.csv")
    save_csv(sample, path)
    print(f"Saved sample in file {path}.")
    return sample, stats, path


if __name__ == "__main__":
    main()
