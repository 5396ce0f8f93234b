import sys, os, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.hmc import HMC
for D, dt in ((32, np.float64), (32, np.float32), (16, np.float64), (8, np.float32)):
    g = GaussianND.ill_conditioned(D, 100.0, 7)
    C = 65536
    s = HMC(g, init_with_seed(C, D, 42, dt) * 0.1, 0.05, 10).set_seed(1)
    s.run(20, 5, to="torch", accept_counts=False); torch.cuda.synchronize()
    s.run(100, 20, to="torch", accept_counts=False); torch.cuda.synchronize()
    ms = s.timing()["kernel_ms"]
    print(json.dumps({"target": f"GaussianND D={D} {dt.__name__}", "chains": C, "run": "(100,20) L=10", "kernel_ms": ms,
                      "leapfrog_steps_per_s": C * 120 * 10 / (ms * 1e-3), "samples_per_s": C * 100 / (ms * 1e-3)}))
