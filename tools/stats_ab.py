"""A/B of the statistics kernels on one HMC sample in one process (MMCMC_STATS_KERNEL is read at every call):
all-parameters register-tile kernel ("tile") against the default (one parameter per wave where it applies)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
x = h.run(n, 50, to="torch")
res = {"tile": [], "default": []}
for rep in range(8):
    for k in ("tile", "default"):
        if k == "tile":
            os.environ["MMCMC_STATS_KERNEL"] = "tile"
        else:
            os.environ.pop("MMCMC_STATS_KERNEL", None)
        for _ in range(5):
            r, e = S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r, e = S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 50 * 1e3)
    last = (r, e)
for k, v in res.items():
    print(f"n={n} {k}: median {np.median(v):.4f} ms  min {min(v):.4f}  all {[round(a, 4) for a in v]}")
print("rhat", last[0], "ess", last[1])
