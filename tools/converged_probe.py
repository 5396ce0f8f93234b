"""How long must config 3 (RosenbrockND D = 3 HMC, eps = 0.032, L = 10, 65 536 chains, f32) run before its conventional
split R-hat drops under 1.05, and what is ESS/s there?  python3 tools/converged_probe.py
One JSON line per (burn-in, n_collect): R-hat (conventional, max over parameters), ESS (min), kernel and diagnostics time."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import DiffableGaussian2D, RosenbrockND
from mini_mcmc_amd.hmc import HMC

C = 65536
cases = [("rosenbrock3", RosenbrockND(3), 3, 0.02, 150, [(6000, 2000)]),
         ("rosenbrock3", RosenbrockND(3), 3, 0.02, 300, [(4000, 2000)]),
         ("rosenbrock3", RosenbrockND(3), 3, 0.015, 250, [(4000, 2000)]),
         ("rosenbrock3", RosenbrockND(3), 3, 0.025, 120, [(6000, 2000)])]
for name, tgt, d, eps, L, runs in cases:
    h = HMC(tgt, init_with_seed(C, d, 42, np.float32), eps, L).set_seed(42)
    done = 0
    for burn, n in runs:
        extra = burn - done
        t0 = time.perf_counter()
        t = h.run(n, max(extra, 0), to="torch", accept_counts=False)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        done = burn + n
        k = float(h.timing()["kernel_ms"])
        S.split_rhat_mean_ess(t)
        t1 = time.perf_counter()
        rhat, ess = S.split_rhat_mean_ess(t)
        st = (time.perf_counter() - t1) * 1e3
        print(json.dumps({"case": name, "eps": eps, "L": L, "burn_in_total": burn, "n_collect": n, "kernel_ms": k, "wall_ms": wall * 1e3,
                          "stats_ms": st, "split_rhat_max_conventional": float((1.0 / rhat).max()), "ess_min": float(ess.min()),
                          "ess_per_s_collect_only": float(ess.min()) / ((k * n / max(n + max(extra, 0), 1) + st) * 1e-3),
                          "mean": [float(v) for v in t.mean(dim=(0, 1)).cpu()]}), flush=True)
        del t
