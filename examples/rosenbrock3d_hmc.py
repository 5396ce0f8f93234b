"""examples/rosenbrock3d_hmc.rs: the 3-D Rosenbrock density by HMC with run_progress (4 chains, eps 0.01, 10 leapfrog steps,
400 + 50), timed as the reference times it -- and then the same call at the chain count the GPU is built for."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.core import init_det, init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC


def run_demo(initial, label):
    sampler = HMC(RosenbrockND(3), initial, 0.01, 10)
    start = time.perf_counter()
    sample, stats = sampler.run_progress(400, 50)
    duration = time.perf_counter() - start
    print("Shape:", list(sample.shape))
    print(stats)
    print(f"HMC sampler ({label}): generating {sample.shape[0] * sample.shape[1]} observations took {duration * 1e3:.2f} ms")
    return sample, stats


def main():
    sample, stats = run_demo(init_det(4, 3, np.float32), "4 chains, as the reference's example")
    assert sample.shape == (4, 400, 3)
    big, _ = run_demo(init_with_seed(65536, 3, 42, np.float32), "65 536 chains")
    assert big.shape == (65536, 400, 3)
    return sample, stats


if __name__ == "__main__":
    main()
