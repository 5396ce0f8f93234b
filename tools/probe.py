import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
C = 65536
def t(s, nc, nd, collect, reps=8):
    s.run(nc, nd, to="torch", accept_counts=False, collect=collect); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s.run(nc, nd, to="torch", accept_counts=False, collect=collect); ms.append(s.timing()["kernel_ms"])
    return float(np.median(ms))
for v in (0, 2):
    mh = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float32)).seed(42).set_kernel_variant(v)
    print("MH v%d collect=1 run(1000,100): %.3f ms   collect=0 run(0,1100): %.3f ms   run(1100,0) collect: %.3f" % (v, t(mh, 1000, 100, True), t(mh, 0, 1100, False), t(mh, 1100, 0, True)))
    h = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42).set_kernel_variant(v)
    print("HMC v%d collect=1 run(400,50): %.3f ms   collect=0 run(0,450): %.3f ms" % (v, t(h, 400, 50, True), t(h, 0, 450, False)))
    for L in (1, 2, 5, 10, 20):
        h = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, L).set_seed(42).set_kernel_variant(v)
        print("   HMC v%d L=%d run(0,450) no output: %.3f ms" % (v, L, t(h, 0, 450, False)))
