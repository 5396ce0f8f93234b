import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, DiffableGaussian2D, StandardNormal
from mini_mcmc_amd.nuts import NUTS
for name, tgt, d in (("Rosenbrock3", RosenbrockND(3), 3), ("Gauss2D", DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2), ("StdNormal8", StandardNormal(8), 8)):
    s = NUTS(tgt, init_with_seed(65536, d, 42) * 0.5, 0.8, mode=0).set_seed(1)
    s._run(100, 100, True, "torch")
    torch.cuda.synchronize()
    lf = s.leapfrog_counts().astype(np.float64)
    w = lf.reshape(1024, 64)
    print(name, "ms", s.timing()["kernel_ms"], "mean", lf.mean(), "pcts 50/90/99/99.9/max", np.percentile(lf, [50, 90, 99, 99.9, 100]),
          "mean of per-wave max", w.max(axis=1).mean(), "max of per-wave max", w.max(axis=1).max(), "hist", s.depth_histogram().tolist())
