"""Config 5 (NUTS, 32-D Gaussian of condition number 1e4, 65 536 chains, f64, max depth 10) kernel time:
    python tools/nuts_cfg5_timing.py [warmup draws [repeats]]      (default 200 100 2: bench.py's side configuration)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = GaussianND.ill_conditioned(32, 1e4, 7)
for _ in range(reps):
    s = NUTS(g, init_with_seed(65536, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10)
    t = s._run(nc, nd, True, "torch")
    torch.cuda.synchronize()
    tm = s.timing()
    lf = float(s.leapfrog_counts().sum())
    k = float(tm["kernel_ms"])
    print(json.dumps({"warmup": nd, "draws": nc, "kernel_ms": k, "n_launches": int(tm["n_launches"]), "leapfrogs": lf,
                      "leapfrog_steps_per_s": lf / (k * 1e-3), "f64_mfma_frac": lf * 2048 / (k * 1e-3) / 78.6e12,
                      "checksum": float(t.double().sum().item()), "variant": s.kernel_variant}), flush=True)
    del s, t
