"""HMC RosenbrockND at dimensions without a compiled register-resident instance (9..31): the functor compiled on first use
into the register skeleton (variant 7, hipRTC) against the run-time-D kernel (variant 6); 65 536 chains, L = 10, run(100, 20)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

for dtype in (np.float32, np.float64):
    for dim in (8, 10, 12, 24, 31):
        init = init_with_seed(65536, dim, 42, dtype) * dtype(0.3)
        t0 = time.perf_counter()
        s = HMC(RosenbrockND(dim), init, 0.01, 10).set_seed(1)
        create_s = time.perf_counter() - t0
        row = {"dim": dim, "dtype": dtype.__name__, "default_variant": s.kernel_variant, "create_s": round(create_s, 2)}
        for v in ([s.kernel_variant] + ([6] if s.kernel_variant == 7 else [])):
            s.set_kernel_variant(v)
            s.run(100, 20, to="torch", accept_counts=False)
            torch.cuda.synchronize()
            s.run(100, 20, to="torch", accept_counts=False)
            torch.cuda.synchronize()
            row[f"variant_{v}_ms"] = round(s.timing()["kernel_ms"], 3)
        print(json.dumps(row), flush=True)
