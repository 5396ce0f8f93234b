"""Per-leaf latency of the one-chain-per-lane NUTS kernel as a function of how many lanes of the wave hold a chain:
the SAME chains (global indices via set_chain_offset: same streams, same trees) run alone, as a wave of 8 / 64, and inside
the full launch.  kernel time / leapfrogs of the longest chain of the launch = cycles per sequential leaf."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS

for name, tgt, d in (("Rosenbrock3", RosenbrockND(3), 3), ("StdNormal8", StandardNormal(8), 8)):
    x0 = init_with_seed(65536, d, 42) * 0.5
    full = NUTS(tgt, x0, 0.8, mode=0).set_seed(1)
    full._run(100, 100, True, "torch")
    torch.cuda.synchronize()
    lf = full.leapfrog_counts().astype(np.float64)
    worst = int(lf.argmax())
    print(f"{name}: full launch {full.timing()['kernel_ms']:.2f} ms, longest chain {worst} with {lf[worst]:.0f} leapfrogs -> "
          f"{full.timing()['kernel_ms'] * 1e6 / lf[worst]:.0f} ns per leaf of the longest chain")
    for n in (1, 2, 8, 64, 4096):
        lo = worst - worst % n if n <= 64 else (worst // 64 * 64 // n) * n
        lo = min(lo, 65536 - n)
        s = NUTS(tgt, x0[lo:lo + n], 0.8, mode=0).set_seed(1).set_chain_offset(lo)
        ts = []
        for rep in range(3):
            s2 = NUTS(tgt, x0[lo:lo + n], 0.8, mode=0).set_seed(1).set_chain_offset(lo)
            s2._run(100, 100, True, "torch")
            torch.cuda.synchronize()
            ts.append(s2.timing()["kernel_ms"])
            l2 = s2.leapfrog_counts().astype(np.float64)
        assert l2[worst - lo] == lf[worst], (l2[worst - lo], lf[worst])
        print(f"   {n:5d} chains [{lo}, {lo + n}): {min(ts):.2f} ms, longest {l2.max():.0f} leapfrogs -> {min(ts) * 1e6 / l2.max():.0f} ns per leaf "
              f"(mean leapfrogs {l2.mean():.0f})")
