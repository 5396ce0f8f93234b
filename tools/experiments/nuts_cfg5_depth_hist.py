import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS
g = GaussianND.ill_conditioned(32, 1e4, 7)
s = NUTS(g, init_with_seed(65536, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10)
s._run(0, 200, True, "torch")
h0 = s.depth_histogram().copy()
s._run(100, 0, True, "torch")
h = s.depth_histogram() - h0
print("depth histogram of the 100 sampling transitions:", h.tolist(), "fractions", np.round(h / h.sum(), 4).tolist())
lf = s.leapfrog_counts()
print("mean leapfrogs per transition", lf.sum() / 65536 / 300)
