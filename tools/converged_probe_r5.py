"""Round 5: a configuration of config 3's TARGET (RosenbrockND D = 3, 65 536 chains) whose conventional split R-hat is <= 1.05,
so that an ESS/s on the metric's own target may be called one.  Candidates: NUTS (per-chain dual-averaging
step size, nuts.rs:550-691) in the three type modes, and HMC with trajectories long enough to cross the banana at longer
collection windows.  One JSON line per case.   python3 tools/converged_probe_r5.py [nuts|hmc|all]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.nuts import NUTS

C = 65536
what = sys.argv[1] if len(sys.argv) > 1 else "all"
# optional case lists as JSON: argv[2] = NUTS [[mode, n_discard, n_collect, target_accept_p], ...], argv[3] = HMC [[eps, L, n_discard, n_collect], ...]
NUTS_CASES = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [(0, 500, 1000, 0.8), (1, 500, 1000, 0.8), (0, 1000, 2000, 0.8), (0, 500, 1000, 0.9), (0, 300, 600, 0.8)]
HMC_CASES = json.loads(sys.argv[3]) if len(sys.argv) > 3 else [(0.02, 150, 6000, 6000), (0.02, 150, 6000, 12000), (0.025, 100, 4000, 8000)]
# exact moments of RosenbrockND(3) by quadrature over x0 (x1 | x0 and x2 | x1 are Gaussian): what a converged sample must show
TRUE_MEAN, TRUE_VAR = [0.78522, 0.84746, 1.06837], [0.22937, 0.35018, 1.63898]


def report(name, t, k_ms, extra):
    S.split_rhat_mean_ess(t)
    t1 = time.perf_counter()
    rhat, ess = S.split_rhat_mean_ess(t)
    st = (time.perf_counter() - t1) * 1e3
    r = float((1.0 / rhat).max())
    x = t.double()
    out = {"case": name, "kernel_ms": k_ms, "stats_ms": st, "split_rhat_max_conventional": r, "ess_min": float(ess.min()),
           "ess": [float(v) for v in ess], "converged": r <= 1.05, "ess_per_s": float(ess.min()) / ((k_ms + st) * 1e-3),
           "mean": [float(v) for v in x.mean(dim=(0, 1)).cpu()], "var": [float(v) for v in x.reshape(-1, x.shape[-1]).var(dim=0).cpu()]}
    out["mean_err"] = [out["mean"][i] - TRUE_MEAN[i] for i in range(3)]
    out["var_ratio"] = [out["var"][i] / TRUE_VAR[i] for i in range(3)]
    out.update(extra)
    print(json.dumps(out), flush=True)


if what in ("nuts", "all"):
    for mode, nd, nc, p in NUTS_CASES:
        s = NUTS(RosenbrockND(3), init_with_seed(C, 3, 42), p, mode=mode).set_seed(42)
        t0 = time.perf_counter()
        t = s._run(nc, nd, True, "torch")
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        tm = s.timing()
        lf = float(s.leapfrog_counts().sum())
        eb = s.adapt_state()["epsilon_bar"]
        report("nuts", t, float(tm["kernel_ms"]), {"mode": mode, "target_accept_p": p, "n_discard": nd, "n_collect": nc, "wall_ms": wall,
                                                   "leapfrogs": lf, "eps_bar_quantiles": [float(np.quantile(eb, q)) for q in (0.001, 0.01, 0.5, 0.99)],
                                                   "depth_histogram": [int(v) for v in s.depth_histogram()]})
        del t, s

if what in ("hmc", "all"):
    for eps, L, nd, nc in HMC_CASES:
        h = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), eps, L).set_seed(42)
        t = h.run(nc, nd, to="torch", accept_counts=False)
        torch.cuda.synchronize()
        report("hmc", t, float(h.timing()["kernel_ms"]), {"eps": eps, "L": L, "n_discard": nd, "n_collect": nc})
        del t, h
