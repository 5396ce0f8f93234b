"""Round 6: can the REFERENCE's sampler -- HMC with fixed step size and trajectory length per handle
(hmc.rs:87-121, 304-431: no jitter) -- give a converged sample of config 3's own target, RosenbrockND(3), using only what its
surface allows?  Round 5 showed that ONE (eps, L) leaves var(x2) at 1.2-1.5x the exact value with split R-hat 1.08-1.98 at every
setting tried.  Tried here:
  (b) the trajectory length drawn per LAUNCH: `HMC::new(target, positions, eps, L)` is called again with the last positions
      (hmc.rs:87-109 takes any initial positions; the state continues through `positions`), L ~ U{Lmin..Lmax}, optionally eps
      from a narrow band as well, `run(n_per, 0)` each -- a chain of handles, every one of them the reference's sampler;
  (a) a GROUP of handles, each with its own fixed eps from a band (the reference takes one step_size per sampler), same L,
      run side by side over disjoint chains; R-hat per handle and pooled.
Yardstick: the exact moments of the target (its density factorises; one-dimensional quadrature, tools/converged_probe_r5.py).
One JSON line per case.   python3 tools/converged_probe_r6.py [cases as JSON]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

C = 65536
TRUE_MEAN, TRUE_VAR = [0.78522, 0.84746, 1.06837], [0.22937, 0.35018, 1.63898]
DEFAULT = [
    {"scheme": "b", "eps": [0.02, 0.02], "L": [50, 200], "n_per": 100, "burn_launches": 20, "keep_launches": 40},
    {"scheme": "b", "eps": [0.01, 0.03], "L": [50, 200], "n_per": 100, "burn_launches": 20, "keep_launches": 40},
    {"scheme": "b", "eps": [0.015, 0.015], "L": [100, 300], "n_per": 100, "burn_launches": 20, "keep_launches": 40},
    {"scheme": "b", "eps": [0.005, 0.03], "L": [20, 300], "n_per": 50, "burn_launches": 40, "keep_launches": 80},
    {"scheme": "a", "eps": [0.008, 0.03], "groups": 8, "L": 150, "n_discard": 3000, "n_collect": 3000},
]
CASES = json.loads(sys.argv[1]) if len(sys.argv) > 1 else DEFAULT


def moments(t):
    x = t.double()
    mean = [float(v) for v in x.mean(dim=(0, 1)).cpu()]
    var = [float(v) for v in x.reshape(-1, x.shape[-1]).var(dim=0).cpu()]
    return mean, var


def report(case, t, k_ms, extra):
    S.split_rhat_mean_ess(t)
    t1 = time.perf_counter()
    rhat, ess = S.split_rhat_mean_ess(t)
    st = (time.perf_counter() - t1) * 1e3
    r = float((1.0 / rhat).max())
    mean, var = moments(t)
    out = dict(case)
    out.update({"kernel_ms": k_ms, "stats_ms": st, "split_rhat_max_conventional": r, "ess_min": float(ess.min()),
                "ess": [float(v) for v in ess], "ess_per_s": float(ess.min()) / ((k_ms + st) * 1e-3), "mean": mean, "var": var,
                "mean_err_over_sd": [(mean[i] - TRUE_MEAN[i]) / TRUE_VAR[i] ** 0.5 for i in range(3)],
                "var_ratio": [var[i] / TRUE_VAR[i] for i in range(3)]})
    out["moments_within_1pct"] = bool(all(abs(v) < 0.01 for v in out["mean_err_over_sd"]) and all(abs(v - 1.0) < 0.01 for v in out["var_ratio"]))
    out["converged"] = bool(r <= 1.05 and out["moments_within_1pct"])
    out.update(extra)
    print(json.dumps(out), flush=True)


for case in CASES:
    rng = np.random.default_rng(7)
    if case["scheme"] == "b":
        state = init_with_seed(C, 3, 42, np.float32)
        n_per, nb, nk = case["n_per"], case["burn_launches"], case["keep_launches"]
        keep = torch.empty((C, nk * n_per, 3), dtype=torch.float32, device="cuda")
        k_ms, acc, n_tr, lf = 0.0, 0.0, 0, 0.0
        used = []
        for launch in range(nb + nk):
            eps = float(rng.uniform(*case["eps"]))
            L = int(rng.integers(case["L"][0], case["L"][1] + 1))
            # a NEW handle per launch, as the reference's constructor allows: the last positions, this launch's (eps, L); its own
            # seed, because a handle's stream starts at iteration 0
            h = HMC(RosenbrockND(3), state, eps, L).set_seed(1000 + launch)
            t = h.run(n_per, 0, to="torch")
            torch.cuda.synchronize()
            k_ms += float(h.timing()["kernel_ms"])
            lf += C * n_per * L
            if launch >= nb:
                keep[:, (launch - nb) * n_per:(launch - nb + 1) * n_per] = t
                acc += float(h.accept_counts.mean())
                n_tr += n_per
                used.append((round(eps, 4), L))
            state = h.state()
            del h, t
        report(case, keep, k_ms, {"accept_rate": acc / n_tr, "leapfrogs": lf, "launches": nb + nk,
                                  "kernel_ms_note": "all launches, burn-in included", "first_launches_eps_L": used[:6]})
        del keep
    else:
        G = case["groups"]
        epss = np.geomspace(case["eps"][0], case["eps"][1], G)
        init = init_with_seed(C, 3, 42, np.float32)
        per = C // G
        parts, k_ms, rh = [], 0.0, []
        for g in range(G):
            h = HMC(RosenbrockND(3), init[g * per:(g + 1) * per], float(epss[g]), case["L"]).set_seed(42)
            h.set_chain_offset(g * per)
            t = h.run(case["n_collect"], case["n_discard"], to="torch")
            torch.cuda.synchronize()
            k_ms += float(h.timing()["kernel_ms"])
            r, e = S.split_rhat_mean_ess(t)
            m, v = moments(t)
            rh.append({"eps": float(epss[g]), "split_rhat_max_conventional": float((1.0 / r).max()), "ess_min": float(e.min()),
                       "accept_rate": float(h.accept_counts.mean()) / (case["n_collect"] + case["n_discard"]),
                       "var_ratio": [v[i] / TRUE_VAR[i] for i in range(3)]})
            parts.append(t)
            del h
        report(case, torch.cat(parts, dim=0), k_ms, {"per_handle": rh})
        del parts
