"""A user-defined target (the 3-D Rosenbrock density restated as a HIP functor, compiled at run time) against the built-in
kernels on config 3's shape: ms per HMC run(400, 50) of 65 536 chains, back to back."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_user_target import ROSENBROCK3
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import IsotropicGaussian, RosenbrockND, UserTarget
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings


def timed(s, nc, nd):
    s.enable_timing(False)
    for _ in range(20):
        s.run(nc, nd, to="torch", accept_counts=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        s.run(nc, nd, to="torch", accept_counts=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100 * 1e3


user = UserTarget("rosenbrock3_t", 3, ROSENBROCK3)
init = init_with_seed(65536, 3, 42, np.float32)
out = {}
out["hmc user target"] = timed(HMC(user, init, 0.032, 10).set_seed(42), 400, 50)
out["hmc built-in, split kernel (variant 5, L unrolled)"] = timed(HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42), 400, 50)
out["hmc built-in, one wave per SIMD (variant 2)"] = timed(HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42).set_kernel_variant(2), 400, 50)
out["mh user target"] = timed(MetropolisHastings(user, IsotropicGaussian(0.1), init).seed(7), 1000, 100)
out["mh built-in (variant 5)"] = timed(MetropolisHastings(RosenbrockND(3), IsotropicGaussian(0.1), init).seed(7), 1000, 100)
for k, v in out.items():
    print(f"{k}: {v:.4f} ms")
