"""Config 5's target at smaller chain counts (the persistent scheduler from 2048 chains on): kernel ms per count.
    python tools/experiments/nuts_cfg5_chains_sweep.py [chains ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS

counts = [int(v) for v in sys.argv[1:]] or [2048, 4096, 16384, 32768]
g = GaussianND.ill_conditioned(32, 1e4, 7)
for c in counts:
    s = NUTS(g, init_with_seed(c, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10)
    t = s._run(100, 200, True, "torch")
    torch.cuda.synchronize()
    tm = s.timing()
    lf = float(s.leapfrog_counts().sum())
    print(json.dumps({"chains": c, "kernel_ms": float(tm["kernel_ms"]), "leapfrog_steps_per_s": lf / (float(tm["kernel_ms"]) * 1e-3),
                      "variant": s.kernel_variant, "min_unit": os.environ.get("MMCMC_LGQ_MIN_UNIT", "default")}), flush=True)
    del s, t
