"""Side measurement: the small kernels (integer-state MH, Gibbs mixture, tracker) at 65 536 chains."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.discrete import BinomialClamp, DiscreteMetropolisHastings, PoissonNonneg, PoissonReflect
from mini_mcmc_amd.gibbs import GibbsSampler, MixtureConditional

C = 65536


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


for name, model, init in (("PoissonReflect(4)", PoissonReflect(4.0), 0), ("BinomialClamp(10,0.3)", BinomialClamp(10, 0.3), 5),
                          ("PoissonNonneg(4)", PoissonNonneg(4.0), 0)):
    s = DiscreteMetropolisHastings(model, np.full(C, init, dtype=np.int32)).seed(42)
    dt = timed(lambda: s.run(1000, 100, to="torch"))
    print(json.dumps({"kernel": "mm_discrete_kernel", "model": name, "chains": C, "run": "(1000, 100)", "wall_ms": dt * 1e3,
                      "samples_per_s": C * 1000 / dt, "out_GBps": C * 1000 * 4 / dt / 1e9}))
g = GibbsSampler(MixtureConditional(-2.0, 1.0, 3.0, 1.5, 0.25), init_det(C, 2)).set_seed(42)
dt = timed(lambda: g.run(1000, 100, to="torch"))
print(json.dumps({"kernel": "mm_gibbs_mixture_kernel", "chains": C, "run": "(1000, 100)", "wall_ms": dt * 1e3,
                  "sweeps_per_s": C * 1100 / dt, "samples_per_s": C * 1000 / dt, "out_GBps": C * 1000 * 16 / dt / 1e9}))
# the tracker on white noise (every row differs: every flag "accepted" -- the one input on which the certificate of
# tracker_paccept_kernel cannot decide and the full 16 384-flag replay runs) and on what it is for, an HMC sample
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
hmc_sample = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42).run(400, 50, to="torch")
for name, x in (("white noise", torch.randn((C, 400, 3), device="cuda")), ("HMC sample of config 3", hmc_sample)):
    tr = S.MultiChainTracker(C, 3)
    dt = timed(lambda: tr.step(x))
    print(json.dumps({"kernel": "tracker_step_kernel + tracker_paccept_kernel", "input": name, "states": [C, 400, 3],
                      "wall_ms": dt * 1e3, "read_GBps": C * 400 * 3 * 4 / dt / 1e9}))
