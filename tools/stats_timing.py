"""split-R-hat / ESS of a [65536, n, 3] f32 sample in HBM: end-to-end ms per call (MMCMC_STATS_KERNEL=mfma selects the
matrix-core kernel, default: register tiles)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
for n in (400, 200, 100, 1000):
    x = torch.randn(65536, n, 3, device="cuda")
    for _ in range(3):
        r, e = S.split_rhat_mean_ess(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r, e = S.split_rhat_mean_ess(x)
    torch.cuda.synchronize()
    print(f"n={n} kernel={os.environ.get('MMCMC_STATS_KERNEL', 'tiles')}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  rhat {r} ess {e}")
