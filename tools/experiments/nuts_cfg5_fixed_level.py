"""Config 5 with the first compaction level fixed by the caller (no pilot launch) against the automatic choice:
    python tools/experiments/nuts_cfg5_fixed_level.py [level ...]      (default 6 7 8)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS

levels = [int(v) for v in sys.argv[1:]] or [6, 7, 8]
g = GaussianND.ill_conditioned(32, 1e4, 7)
for lv in [None] + levels:
    s = NUTS(g, init_with_seed(65536, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10)
    if lv is not None:
        s.set_compaction(lv)
    t = s._run(100, 200, True, "torch")
    torch.cuda.synchronize()
    tm = s.timing()
    print(json.dumps({"first_level": lv if lv is not None else "auto (pilot of 16 transitions at 5)", "kernel_ms": float(tm["kernel_ms"]),
                      "n_launches": int(tm["n_launches"]), "checksum": float(t.double().sum().item())}), flush=True)
    del s, t
