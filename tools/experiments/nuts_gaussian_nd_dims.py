"""NUTS on the dense Gaussian at the compiled dimensions 10 / 16 / 20 in the f32 modes: pair kernel (default) against the lanes in step."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS
for d in (10, 16, 20):
    tgt = GaussianND.ill_conditioned(d, 100.0, 7)
    for mode in (0, 1, 2):
        s = NUTS(tgt, init_with_seed(16384, d, 42) * 0.1, 0.8, mode=mode).set_seed(1)
        row = {"target": "GaussianND", "dim": d, "mode": mode, "default": s.kernel_variant}
        outs = {}
        for v in sorted({s.kernel_variant, 0}, reverse=True):
            s2 = NUTS(tgt, init_with_seed(16384, d, 42) * 0.1, 0.8, mode=mode).set_seed(1).set_kernel_variant(v)
            outs[v] = s2._run(50, 50, True, "torch"); torch.cuda.synchronize()
            row[f"variant_{v}_ms"] = round(s2.timing()["kernel_ms"], 2)
        if s.kernel_variant in (5, 0):
            row["equal"] = all(bool((o == outs[0]).all()) for o in outs.values())
        print(json.dumps(row), flush=True)
