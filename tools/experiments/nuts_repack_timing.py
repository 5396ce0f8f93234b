"""[round 5: the feature this script timed -- mmcmc_nuts_set_repacking, measured slower in round 4 -- has been removed from the library; kept as the record of how profiles/r4g_nuts_repack_timing.jsonl was made]
The asynchronous-lane NUTS kernel with and without re-packing the chains between the launches of a run
(mmcmc_nuts_set_repacking): kernel time of run_progress(100, 100) at 65 536 chains, same bits either way.
python3 tools/nuts_repack_timing.py [launch counts ...]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import DiffableGaussian2D, RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
launches = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 6]
for name, tgt, d in (("RosenbrockND(3)", RosenbrockND(3), 3), ("StandardNormal(8)", StandardNormal(8), 8),
                     ("DiffableGaussian2D", DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2)):
    for mode in (0,):
        ref = None
        for nl in launches:
            n = 65536
            s = NUTS(tgt, init_with_seed(n, d, 42) * 0.5, 0.8, mode=mode).set_seed(1).set_repacking(nl)
            best = None
            for rep in range(2):
                out = s._run(100, 100, True, "torch")
                torch.cuda.synchronize()
                ms = s.timing()["kernel_ms"]
                best = ms if best is None else min(best, ms)
                if rep == 0:
                    first = out.cpu().numpy()
                    lf = s.leapfrog_counts().copy()
            if ref is None:
                ref = (first, lf)
            same = bool(np.array_equal(first, ref[0]) and np.array_equal(lf, ref[1]))
            print(json.dumps({"target": name, "mode": mode, "launches": nl, "kernel_ms_first": None, "kernel_ms_best_of_2": best,
                              "n_launches": s.timing()["n_launches"], "bit_identical_to_one_launch": same}), flush=True)
