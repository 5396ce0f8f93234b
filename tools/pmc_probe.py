"""One configuration per process, for rocprofv3 --pmc passes: python3 tools/pmc_probe.py {hmc|mh} {collect|nocollect} [variant]
or python3 tools/pmc_probe.py nuts5 <chains> [warmup draws]  (config-5 target, lane-group kernel)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
what, mode = sys.argv[1], sys.argv[2]
if what == "nuts3":  # one-chain-per-lane NUTS on RosenbrockND(3): python3 tools/pmc_probe.py nuts3 <scalar mode 0|1|2> [variant]
    from mini_mcmc_amd.nuts import NUTS
    s = NUTS(RosenbrockND(3), init_with_seed(65536, 3, 42) * 0.5, 0.8, mode=int(mode)).set_seed(1)
    if len(sys.argv) > 3:
        s.set_kernel_variant(int(sys.argv[3]))
    s._run(100, 100, True, "torch")
    torch.cuda.synchronize()
    print("nuts3", mode, s.kernel_variant, s.timing(), int(s.leapfrog_counts().sum()))
    sys.exit(0)
if what == "stats":  # split-R-hat / ESS of a [65536, 400, 3] f32 sample in HBM
    from mini_mcmc_amd import stats as S
    n = int(mode) if mode.isdigit() and int(mode) > 0 else 400  # python3 tools/pmc_probe.py stats <n> [kernel]
    if len(sys.argv) > 3:
        S.set_kernel(sys.argv[3])
    x = torch.randn(65536, n, 3, device="cuda").cumsum(dim=1) * 0.1
    for _ in range(3):
        S.split_rhat_mean_ess(x)
    torch.cuda.synchronize()
    print("stats done")
    sys.exit(0)
if what == "nuts5":
    from mini_mcmc_amd.distributions import GaussianND
    from mini_mcmc_amd.nuts import NUTS
    n = int(mode)
    nd, nc = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (50, 50)
    g = GaussianND.ill_conditioned(32, 1e4, 7)
    s = NUTS(g, init_with_seed(n, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10)
    if os.environ.get("NUTS_VARIANT"):
        s.set_kernel_variant(int(os.environ["NUTS_VARIANT"]))
    s.run_progress(nc, nd, to="torch")
    torch.cuda.synchronize()
    print("nuts5", n, s.timing(), int(s.leapfrog_counts().sum()))
    sys.exit(0)
variant = int(sys.argv[3]) if len(sys.argv) > 3 else None  # None: the sampler's default
C = 65536
if what == "hmc":
    s = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42)
    if variant is not None:
        s.set_kernel_variant(variant)
    nc, nd = (400, 50) if mode == "collect" else (0, 450)
else:
    s = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float32)).seed(42)
    if variant is not None:
        s.set_kernel_variant(variant)
    nc, nd = (1000, 100) if mode == "collect" else (0, 1100)
for _ in range(3):
    s.run(nc, nd, to="torch", accept_counts=False, collect=(mode == "collect"))
torch.cuda.synchronize()
print(what, mode, s.timing())
