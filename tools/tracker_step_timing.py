"""MultiChainTracker.step on an HMC sample [65536, 400, 3] (realistic accept flags: the p_accept certificate applies):
ms per call, back to back."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
x = h.run(400, 50, to="torch")
tr = S.MultiChainTracker(65536, 3)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(x)
torch.cuda.synchronize()
print("tracker.step on an HMC sample [65536, 400, 3]: %.4f ms" % ((time.perf_counter() - t0) / 20 * 1e3), tr.p_accept)
