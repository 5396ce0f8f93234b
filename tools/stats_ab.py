"""A/B of the statistics kernels on one HMC sample of config 3's shape in one process (mmcmc_stats_set_kernel):
the power-spectrum kernel ("fft", the default above 100 draws per half-chain) against the register-tile kernel with one
parameter per wave ("tile1").  End-to-end time per call (device sample -> R-hat / ESS on the host).

    python tools/stats_ab.py [n ...]        default: 400 1000
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

ns = [int(a) for a in sys.argv[1:]] or [400, 1000]
for n in ns:
    h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
    x = h.run(n, 50, to="torch")
    kinds = ("fft", "tile1")
    res = {k: [] for k in kinds}
    last = {}
    for rep in range(6):
        for k in kinds:
            S.set_kernel(k)
            for _ in range(5):
                r, e = S.split_rhat_mean_ess(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                r, e = S.split_rhat_mean_ess(x)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 50 * 1e3)
            last[k] = (r, e)
    S.set_kernel("auto")
    for k, v in res.items():
        print(f"n={n} {k}: median {np.median(v):.4f} ms  min {min(v):.4f}  all {[round(a, 4) for a in v]}", flush=True)
    for k in kinds:
        print(f"   {k}: rhat {last[k][0]} ess {last[k][1]}")
    print(f"   fft vs tile1: rhat rel {np.max(np.abs(last['fft'][0] / last['tile1'][0] - 1)):.2e} ess rel {np.max(np.abs(last['fft'][1] / last['tile1'][1] - 1)):.2e}", flush=True)
