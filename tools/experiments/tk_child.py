import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mini_mcmc_amd._lib as L
L.LIB_PATH = sys.argv[1]
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
x = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42).run(400, 50, to="torch")
tr = S.MultiChainTracker(65536, 3)
for _ in range(12):
    tr.step(x)
torch.cuda.synchronize()
