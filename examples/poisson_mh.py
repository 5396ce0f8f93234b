"""examples/poisson_mh.rs: Poisson(4) on the non-negative integers by MH with a +-1 random walk that stays at 0 from 0,
ONE chain from 0, run(10 000, 1000); frequencies for k = 0..20 next to the pmf."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd.discrete import DiscreteMetropolisHastings, PoissonNonneg


def poisson_pmf(k, lam):
    return math.exp(k * math.log(lam) - lam - math.lgamma(k + 1.0))


def main():
    mh = DiscreteMetropolisHastings(PoissonNonneg(4.0), [0])
    sample = mh.run(10_000, 1_000)
    chain0 = sample.reshape(10_000)
    print("Elements in chain:", chain0.size)
    cutoff = 20
    counts = np.bincount(chain0[chain0 <= cutoff], minlength=cutoff + 1)
    print(f"Frequencies for k=0..{cutoff}, from chain after burn-in:")
    freqs = counts / chain0.size
    for k, f in enumerate(freqs):
        print(f"k={k:2}: freq ~ {f:.3f}   pmf {poisson_pmf(k, 4.0):.3f}")
    return chain0, freqs, np.array([poisson_pmf(k, 4.0) for k in range(cutoff + 1)])


if __name__ == "__main__":
    main()
