import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
for tgt, d in ((StandardNormal(12), 12), (RosenbrockND(11), 11), (StandardNormal(24), 24)):
    for mode in (0, 2):
        t0 = time.perf_counter()
        s = NUTS(tgt, init_with_seed(16384, d, 42) * 0.5, 0.8, mode=mode).set_seed(1)
        row = {"target": type(tgt).__name__, "dim": d, "mode": mode, "create_s": round(time.perf_counter() - t0, 2), "default": s.kernel_variant}
        for v in (s.kernel_variant, 6):  # 7: the run-time compiled pair kernel; 6: run-time-D kernel
            s2 = NUTS(tgt, init_with_seed(16384, d, 42) * 0.5, 0.8, mode=mode).set_seed(1).set_kernel_variant(v)
            s2._run(50, 50, True, "torch"); torch.cuda.synchronize()
            row[f"variant_{v}_ms"] = round(s2.timing()["kernel_ms"], 2)
        print(json.dumps(row), flush=True)
