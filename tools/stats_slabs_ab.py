"""Slab-count sweep (MMCMC_STATS_WAVES) of the one-parameter-per-wave statistics kernel on one HMC sample [65536, 400, 3]:
median / minimum ms per split_rhat_mean_ess call for each setting."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
x = h.run(400, 50, to="torch")
ws = [1024, 1365, 1707, 2048, 2730, 4096]
res = {w: [] for w in ws}
for rep in range(5):
    for w in ws:
        os.environ["MMCMC_STATS_WAVES"] = str(w)
        for _ in range(5):
            S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        res[w].append((time.perf_counter() - t0) / 50 * 1e3)
for w, v in res.items():
    print(w, round(float(np.median(v)), 4), round(min(v), 4))
