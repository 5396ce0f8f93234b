"""Configs 2 and 3 in f64 on the library in place (ms per run, back to back): A/B of the f64 split-kernel settings."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
C = 65536
for name in ("hmc", "mh"):
    if name == "hmc":
        s = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float64), 0.032, 10).set_seed(42); nc, nd = 400, 50
    else:
        s = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float64)).seed(42); nc, nd = 1000, 100
    if os.environ.get("V"):
        s.set_kernel_variant(int(os.environ["V"]))
    s.enable_timing(False)
    for _ in range(10):
        s.run(nc, nd, to="torch", accept_counts=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        s.run(nc, nd, to="torch", accept_counts=False)
    torch.cuda.synchronize()
    print(name, "f64", round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms", "variant", s.kernel_variant if hasattr(s, "kernel_variant") else "")
