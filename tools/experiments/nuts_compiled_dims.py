"""NUTS on the compiled instances at D = 10, 16, 20: the pair kernel (variant 5, now the default there) against the lanes in
step (variant 0); 16 384 chains, 50 + 50."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
for tgt, d in ((StandardNormal(10), 10), (RosenbrockND(10), 10), (RosenbrockND(16), 16), (StandardNormal(20), 20)):
    for mode in (0, 2):
        s = NUTS(tgt, init_with_seed(16384, d, 42) * 0.5, 0.8, mode=mode).set_seed(1)
        row = {"target": type(tgt).__name__, "dim": d, "mode": mode, "default": s.kernel_variant}
        outs = {}
        for v in sorted({s.kernel_variant, 0}, reverse=True):
            s2 = NUTS(tgt, init_with_seed(16384, d, 42) * 0.5, 0.8, mode=mode).set_seed(1).set_kernel_variant(v)
            outs[v] = s2._run(50, 50, True, "torch"); torch.cuda.synchronize()
            row[f"variant_{v}_ms"] = round(s2.timing()["kernel_ms"], 2)
        row["equal"] = all(bool((o == outs[0]).all()) for o in outs.values())
        print(json.dumps(row), flush=True)
