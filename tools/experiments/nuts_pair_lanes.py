"""Small-D NUTS (asynchronous lanes, leaves in pairs) with fewer than 64 chains per wave (MMCMC_NUTS_PAIR_LANES in a TUNING
build): half-filled waves, twice as many, two per SIMD.  Prints kernel_ms and a checksum of the sample per setting."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import DiffableGaussian2D, RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
for name, tgt, d in (("RosenbrockND(3)", RosenbrockND(3), 3), ("DiffableGaussian2D", DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2),
                     ("StandardNormal(8)", StandardNormal(8), 8)):
    for lanes in (64, 32, 48, 16):
        os.environ["MMCMC_NUTS_PAIR_LANES"] = str(lanes)
        for rep in range(2):
            s = NUTS(tgt, init_with_seed(65536, d, 42) * 0.5, 0.8, mode=0).set_seed(1).set_kernel_variant(5)
            out = s._run(100, 100, True, "torch")
            torch.cuda.synchronize()
            ms = s.timing()["kernel_ms"]
        print(json.dumps({"target": name, "lanes": lanes, "kernel_ms": ms, "checksum": float(out.double().sum()),
                          "leapfrogs": int(s.leapfrog_counts().sum())}), flush=True)
