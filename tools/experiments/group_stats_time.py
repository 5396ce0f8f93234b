import sys, time, numpy as np
sys.path.insert(0, '' + __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))) + '')
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.group import HMCGroup
for devs in ([0], [0, 0]):
    n = 65536 * len(devs)
    g = HMCGroup(RosenbrockND(3), init_with_seed(n, 3, 42, np.float32), 0.032, 10, devices=devs).set_seed(42)
    g.run_async(400, 50)
    g.sync()
    for _ in range(3):
        g.split_rhat_mean_ess()
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); g.split_rhat_mean_ess(); t.append(time.perf_counter() - t0)
    print(devs, "group stats ms median", sorted(t)[10] * 1e3)
