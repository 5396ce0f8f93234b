import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle as O
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.nuts import NUTS
g = GaussianND.ill_conditioned(16, 1e6, 3)
init = init_with_seed(40, 16, 5) * 0.3
ref, pos, ad, nlf = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, 16, [], init, 0.8, 4, 70, seed=7, matrix=g.precision, max_depth=12)
for v in (1, 2, 3):
    s = NUTS(g, init, 0.8, mode=2).set_seed(7).set_max_depth(12).set_kernel_variant(v)
    out = s.run(4, 70)
    print(v, np.array_equal(out, ref), np.array_equal(s.leapfrog_counts(), nlf), s.depth_histogram().tolist())
# and at scale through the scheduler
init = init_with_seed(2304, 16, 5) * 0.3
a = NUTS(g, init, 0.8, mode=2).set_seed(7).set_max_depth(12).set_kernel_variant(3).run(3, 60)
b = NUTS(g, init, 0.8, mode=2).set_seed(7).set_max_depth(12).set_kernel_variant(1).run(3, 60)
print("scheduler depth 12:", np.array_equal(a, b))
