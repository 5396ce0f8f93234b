"""How well do 65 536 chains x 100 recorded NUTS draws (mode 0, 200 warm-up) of RosenbrockND(3) pin the pooled covariance?
GPU run against the recursive oracle on the same stream: relative difference of the covariances, and its standard error
from 64 groups of chains (tests/test_gpu_vs_reference_order.py::test_nuts_mode0_full_size_distribution_vs_recursive_oracle)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import oracle as O
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.nuts import NUTS
C, nd, nc = 65536, 200, 100
init = init_with_seed(C, 3, 42) * 0.5
s = NUTS(RosenbrockND(3), init, 0.8, mode=0).set_seed(7).set_max_depth(10)
out = s._run(nc, nd, True, "numpy").astype(np.float64)
o = O.NUTS(O.rosenbrock_nd(3), init, 0.8, mode=0).use_engine_stream(7).set_max_depth(10)
ref = o.run(nc, nd, progress=True, n_threads=os.cpu_count())
a, b = out.reshape(-1, 3), ref.reshape(-1, 3)
ca, cb = np.cov(a.T), np.cov(b.T)
print("rel cov diff", np.abs(ca - cb) / np.sqrt(np.outer(np.diag(cb), np.diag(cb))))
# sampling error of the pooled covariance: chains in 64 groups
g = 64
da = np.stack([np.cov(out[i::g].reshape(-1, 3).T) for i in range(g)])
db = np.stack([np.cov(ref[i::g].reshape(-1, 3).T) for i in range(g)])
print("SE of a group-mean cov (rel)", da.std(axis=0) / np.sqrt(g) / np.sqrt(np.outer(np.diag(cb), np.diag(cb))))
print("paired SE of diff (rel)", (da - db).std(axis=0) / np.sqrt(g) / np.sqrt(np.outer(np.diag(cb), np.diag(cb))))
same = (s.leapfrog_counts().astype(np.int64) == np.array([o.chain_state(i)["n_leapfrog_total"] for i in range(C)]).astype(np.int64)).mean()
print("chains with equal total leapfrog counts", same)
print("kurtosis x3", ((b[:,2]-b[:,2].mean())**4).mean()/b[:,2].var()**2)
