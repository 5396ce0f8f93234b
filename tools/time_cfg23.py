"""Back-to-back launches of config 2 (MH) and config 3 (HMC) on the library in place: ms per run, best of 3 x 200
(used to compare builds of the split kernel: python tools/time_cfg23.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

C = 65536
res = {}
for name in ("hmc", "mh"):
    if name == "hmc":
        s = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42)
        nc, nd = 400, 50
    else:
        s = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float32)).seed(42)
        nc, nd = 1000, 100
    s.enable_timing(False)
    for _ in range(20):
        s.run(nc, nd, to="torch", accept_counts=False)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            s.run(nc, nd, to="torch", accept_counts=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    res[name] = round(best, 4)
print(res)
