"""Round 6: how robust is the chain-of-handles HMC run on RosenbrockND(3) (hmc.run_chain_of_handles) to the noise seed and to the
(eps, L) schedule's band?  The first choice -- eps ~ U(0.01, 0.03), L ~ U{50..200}, 100 x (20 + 40) -- gave R-hat 1.004 / moments
within 0.1 % with one seed (profiles/r6m_converged_probe.jsonl) and R-hat 1.016 / var(x2) 2.5 % high with another (r6x bench)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import run_chain_of_handles
TM, TV = [0.78522, 0.84746, 1.06837], [0.22937, 0.35018, 1.63898]
CASES = {"A": ((0.01, 0.03), (50, 200), 100, 20, 40), "B": ((0.005, 0.03), (20, 300), 50, 40, 80),
         "C": ((0.005, 0.02), (50, 300), 100, 20, 40), "D": ((0.004, 0.016), (100, 400), 100, 20, 40),
         "E": ((0.004, 0.012), (150, 450), 100, 15, 30)}
for name in (sys.argv[1] if len(sys.argv) > 1 else "ABCDE"):
    eps, L, n_per, nb, nk = CASES[name]
    for seed, sched in ((42, 7), (1000, 7), (7, 3), (12345, 11)):
        t, info = run_chain_of_handles(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), eps, L, n_per, nb, nk, seed=seed, schedule_seed=sched)
        S.split_rhat_mean_ess(t)
        t0 = time.perf_counter(); rh, es = S.split_rhat_mean_ess(t); st = (time.perf_counter() - t0) * 1e3
        x = t.double().reshape(-1, 3)
        mean, var = x.mean(dim=0).cpu().numpy(), x.var(dim=0).cpu().numpy()
        err = max(np.abs(mean - TM) / np.array(TM)).item(), max(np.abs(var / np.array(TV) - 1)).item()
        r = float((1.0 / rh).max())
        print(json.dumps({"case": name, "eps": eps, "L": L, "n_per": n_per, "burn": nb, "keep": nk, "seed": seed, "schedule_seed": sched, "rhat": r,
                          "ess_min": float(es.min()), "kernel_ms": info["kernel_ms"], "stats_ms": st, "mean_rel_err": err[0], "var_rel_err": err[1],
                          "accept_rate": info["accept_rate"], "ok": bool(r <= 1.05 and max(err) <= 0.01),
                          "ess_per_s": float(es.min()) / ((info["kernel_ms"] + st) * 1e-3)}), flush=True)
        del t, x
