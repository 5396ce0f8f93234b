"""Times probe builds of the tracker (tools/experiments/tk/lib_*.so) on an HMC sample [65536, 400, 3]; one child per build."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, sys.argv[2])
import torch
import mini_mcmc_amd._lib as L
L.LIB_PATH = sys.argv[1]
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
x = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42).run(400, 50, to="torch")
tr = S.MultiChainTracker(65536, 3)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t0 = time.perf_counter(); tr.step(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{sys.argv[1].split('/')[-1]}: median {np.median(ts) * 1e3:.4f} ms  min {min(ts) * 1e3:.4f}", flush=True)
'''
for rep in range(2):
    for lib in sorted(glob.glob(os.path.join(HERE, "tk", "lib_*.so"))):
        subprocess.run([sys.executable, "-c", CHILD, lib, os.path.dirname(os.path.dirname(HERE))])
