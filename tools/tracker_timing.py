import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
s = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
out = s.run(400, 50, to="torch")
torch.cuda.synchronize()
tr = S.MultiChainTracker(65536, 3)
tr.step(out); tr._stats(); torch.cuda.synchronize()
for k in (400, 40):
    tr = S.MultiChainTracker(65536, 3)
    t0 = time.perf_counter()
    for t in range(0, 400, k):
        tr.step(out, t0=t, k=k)
    r = tr._stats()
    torch.cuda.synchronize()
    print(f"tracker over [65536,400,3] in blocks of {k}: {(time.perf_counter()-t0)*1e3:.2f} ms", r[1], r[2])
t0 = time.perf_counter(); smp, st = s.run_progress(400, 50, every=100); torch.cuda.synchronize()
print(f"run_progress(400,50, every=100): {(time.perf_counter()-t0)*1e3:.2f} ms")
