"""The reference's large-dimension timing shape (hmc.rs:882-916: 6 chains, D = 10 000, RosenbrockND, eps = 0.01, L = 50,
run(100, 100)) on the one-chain-per-workgroup kernel (variant 8) and on the run-time-D kernel (variant 6); plus more
chains of the same dimension."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

dim = 10000
for chains, variants in ((6, (8, 6)), (64, (8,)), (256, (8,)), (1024, (8,))):
    one = init_with_seed(1, dim, 42, np.float32)
    init = np.repeat(one, chains, axis=0)
    for v in variants:
        s = HMC(RosenbrockND(dim), init, 0.01, 50).set_seed(42).set_kernel_variant(v)
        s.run(2, 0, to="torch")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(100, 100, to="torch", accept_counts=False)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        k = s.timing()["kernel_ms"]
        lf = chains * 200 * 50
        print(json.dumps({"chains": chains, "dim": dim, "variant": v, "kernel_ms": k, "wall_ms": wall, "leapfrog_steps_per_s": lf / (k * 1e-3),
                          "coordinate_updates_per_s": lf * dim / (k * 1e-3), "us_per_leapfrog_step_of_a_chain": k * 1e3 / (200 * 50)}), flush=True)
