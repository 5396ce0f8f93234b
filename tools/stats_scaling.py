"""Device time of mmcmc_stats_partials (power-spectrum kernel + tail + inverse) against the number of chains: fixed cost
and cost per chain.  python tools/stats_scaling.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for c in (4096, 16384, 32768, 65536, 131072, 262144):
    x = torch.randn(c, n, 3, device="cuda")
    for _ in range(5):
        S.stats_partials(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        S.stats_partials(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"n={n} chains={c}: {ms * 1e3:.1f} us per call, {c * n * 3 * 4 / ms / 1e9:.3f} TB/s", flush=True)
