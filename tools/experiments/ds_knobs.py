"""Times the builds of tools/experiments/ds_knobs.sh (lib_NN_RB.so) on PoissonReflect(4), 65 536 chains, run(1000, 100);
one child process per build."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, sys.argv[2])
import torch
import mini_mcmc_amd._lib as L
L.LIB_PATH = sys.argv[1]
from mini_mcmc_amd.discrete import DiscreteMetropolisHastings, PoissonReflect
s = DiscreteMetropolisHastings(PoissonReflect(4.0), np.zeros(65536, dtype=np.int32)).seed(42)
for _ in range(3):
    s.run(1000, 100, to="torch")
torch.cuda.synchronize()
ts = []
for _ in range(25):
    t0 = time.perf_counter(); s.run(1000, 100, to="torch"); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{sys.argv[1].split('/')[-1]}: median {np.median(ts) * 1e3:.4f} ms  min {min(ts) * 1e3:.4f}", flush=True)
'''
for lib in sorted(glob.glob(os.path.join(HERE, "ds", "lib_*.so"))):
    subprocess.run([sys.executable, "-c", CHILD, lib, os.path.dirname(os.path.dirname(HERE))])
