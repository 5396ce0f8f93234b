"""Sweep of the power-spectrum kernel's build variants (library built with `make TUNING=1`): waves per SIMD and twiddles
in registers / LDS, on an HMC sample of config 3's shape.  End-to-end ms per call.
    python tools/stats_fft_sweep.py n wpe:twl [wpe:twl ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

n = int(sys.argv[1])
variants = [tuple(int(v) for v in (a.split(":") + ["100"])[:3]) for a in sys.argv[2:]]  # wpe:twl[:workgroups %]
h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
x = h.run(n, 50, to="torch")
res = {v: [] for v in variants}
ref = None
for rep in range(5):
    for v in variants:
        os.environ["MMCMC_FFT_WPE"], os.environ["MMCMC_FFT_TWL"], os.environ["MMCMC_FFT_NWG_MULT"] = str(v[0]), str(v[1]), str(v[2])
        for _ in range(5):
            r, e = S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r, e = S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 50 * 1e3)
        if ref is None:
            ref = (r, e)
        assert np.allclose(r, ref[0], rtol=1e-5) and np.allclose(e, ref[1], rtol=1e-4), (v, r, e, ref)
for v, t in res.items():
    print(f"n={n} wpe={v[0]} twl={v[1]} workgroups {v[2]} %: median {np.median(t):.4f} ms  min {min(t):.4f}", flush=True)
