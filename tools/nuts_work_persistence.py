"""Is a chain's NUTS work persistent?  Per-chain leapfrog counts of the first 25 / 100 / 200 transitions of the same run
(same seed: the shorter runs are prefixes of the longer one) and how well an early window predicts a later one -- the
premise of re-packing chains between the launches of a run by their recent work (csrc/mm_nuts_api.hip).
python3 tools/nuts_work_persistence.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS

C = 65536
for name, tgt, d in (("RosenbrockND(3)", RosenbrockND(3), 3), ("StandardNormal(8)", StandardNormal(8), 8)):
    init = init_with_seed(C, d, 42) * 0.5
    cum = {}
    for n_col, n_dis in ((0, 12), (0, 25), (0, 50), (0, 100), (100, 100)):
        s = NUTS(tgt, init, 0.8, mode=0).set_seed(1)
        if n_col == 0 and n_dis < 100:
            # the adaptation horizon must stay 100: run the prefix as warm-up of a run whose horizon is 100 -- not expressible
            # through run(); a horizon of n_dis gives the same first n_dis transitions (the horizon only ends the adaptation)
            pass
        s._run(n_col, n_dis, True, "numpy")
        cum[n_col + n_dis] = s.leapfrog_counts().astype(np.float64)
    w = {"0-12": cum[12], "12-25": cum[25] - cum[12], "25-50": cum[50] - cum[25], "50-100": cum[100] - cum[50], "100-200": cum[200] - cum[100],
         "25-200": cum[200] - cum[25], "12-200": cum[200] - cum[12]}
    def corr(a, b):
        return float(np.corrcoef(w[a], w[b])[0, 1])
    def top_overlap(a, b, frac=0.01):
        k = int(C * frac)
        ta, tb = set(np.argsort(-w[a])[:k].tolist()), set(np.argsort(-w[b])[:k].tolist())
        return len(ta & tb) / k
    print(json.dumps({"target": name, "mean_total": float(cum[200].mean()), "max_total": float(cum[200].max()),
                      "corr(0-12, 12-200)": corr("0-12", "12-200"), "corr(12-25, 25-200)": corr("12-25", "25-200"),
                      "corr(25-50, 50-100)": corr("25-50", "50-100"), "corr(50-100, 100-200)": corr("50-100", "100-200"),
                      "top1%_overlap(12-25, 25-200)": top_overlap("12-25", "25-200"),
                      "top1%_overlap(50-100, 100-200)": top_overlap("50-100", "100-200"),
                      "share_of_work": {k: float(v.sum() / cum[200].sum()) for k, v in w.items() if k in ("0-12", "12-25", "25-50", "50-100", "100-200")},
                      "max_over_mean": {k: float(v.max() / v.mean()) for k, v in w.items()}}), flush=True)
