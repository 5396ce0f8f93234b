import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
def t(s, nc, nd, collect, reps=8):
    s.run(nc, nd, to="torch", accept_counts=False, collect=collect); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s.run(nc, nd, to="torch", accept_counts=False, collect=collect); ms.append(s.timing()["kernel_ms"])
    return float(np.median(ms))
for C in (4096, 16384, 65536, 131072, 262144):
    h = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42)
    a, b = t(h, 400, 50, True), t(h, 0, 450, False)
    print("C=%7d  with output %.3f ms   without %.3f ms   overhead %.3f ms   out bytes/overhead = %.2f TB/s" % (C, a, b, a - b, C * 400 * 12 / ((a - b) * 1e-3) / 1e12))
