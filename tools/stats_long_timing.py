"""Diagnostics of long samples: split-R-hat / ESS of [65536, n, 3] f32 for n = 2048 (one wave-level transform), 4000, 8000
(mm_chain_fft_long_kernel) -- and the direct kernel at 4000 for comparison.  python3 tools/stats_long_timing.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
for n, kernels in ((2048, ("auto",)), (4000, ("auto", "direct")), (8000, ("auto",)), (16000, ("auto",))):
    c = 65536 if n <= 8000 else 16384
    x = (torch.randn(c, n, 3, device="cuda") * 0.1).cumsum(dim=1)
    for k in kernels:
        S.set_kernel(k)
        S.split_rhat_mean_ess(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3 if k == "auto" else 1
        for _ in range(reps):
            r, e = S.split_rhat_mean_ess(x)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"shape": [c, n, 3], "kernel": k, "ms": ms, "GBps": c * n * 3 * 4 / (ms * 1e-3) / 1e9,
                          "ess_min": float(e.min())}), flush=True)
    S.set_kernel("auto")
    del x
