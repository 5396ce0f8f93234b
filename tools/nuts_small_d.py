"""Side measurement: the one-chain-per-lane NUTS kernels on small targets at many chains (lane divergence by tree
depth): variant 5 (asynchronous lanes, leaves in pairs: default), variant 4 (asynchronous lanes) and variant 0 (lanes in step)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import DiffableGaussian2D, RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
for name, tgt, d in (("DiffableGaussian2D", DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2),
                     ("RosenbrockND(3)", RosenbrockND(3), 3), ("StandardNormal(8)", StandardNormal(8), 8)):
    for mode in (0, 2):
        for variant in (5, 4, 0):
            n = 65536
            s = NUTS(tgt, init_with_seed(n, d, 42) * 0.5, 0.8, mode=mode).set_seed(1).set_kernel_variant(variant)
            s._run(100, 100, True, "torch")
            torch.cuda.synchronize()
            ms = s.timing()["kernel_ms"]
            lf = int(s.leapfrog_counts().sum())
            print(json.dumps({"target": name, "mode": mode, "variant": variant, "chains": n, "kernel_ms": ms,
                              "leapfrog_steps_per_s": lf / (ms * 1e-3), "draws_per_s": n * 100 / (ms * 1e-3),
                              "depth_hist": s.depth_histogram().tolist()}), flush=True)
