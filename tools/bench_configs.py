#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the driver's bench line): the other BASELINE.json configurations on one GPU.
  config 2: Gaussian2D MH, 65 536 chains, f32, run(1000, 100)
  config 3: RosenbrockND D=3 HMC (what bench.py measures) -- all kernel variants, and the PCIe-inclusive host-output rate
  config 5: NUTS, 32-D ill-conditioned Gaussian (cond 1e4), f64, max depth 10, chains scaled by --nuts-chains
  stats   : split-R-hat / ESS of a [65 536, 400, 3] sample
Prints one JSON object per line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nuts-chains", type=int, default=8192)
    ap.add_argument("--nuts-warmup", type=int, default=200)
    ap.add_argument("--nuts-draws", type=int, default=100)
    ap.add_argument("--nuts-variant", type=int, default=-1, help="-1: the default mapping; 0: lane per chain; 1: lane group / MFMA")
    ap.add_argument("--nuts-compaction-start", type=int, default=-1)
    ap.add_argument("--nuts-groups", type=int, default=0)
    ap.add_argument("--skip", default="")
    args = ap.parse_args()
    import torch

    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, GaussianND, IsotropicGaussian, RosenbrockND
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
    from mini_mcmc_amd.nuts import NUTS

    C = 65536
    skip = set(args.skip.split(","))

    def timed(sampler, nc, nd, reps=10):
        sampler.run(nc, nd, to="torch", accept_counts=False)
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            sampler.run(nc, nd, to="torch", accept_counts=False)
            ms.append(sampler.timing()["kernel_ms"])
        return float(np.median(ms)), float(np.min(ms))

    if "mh" not in skip:
        init = init_with_seed(C, 2, 42, np.float32)
        for variant in (0, 2):
            mh = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init).seed(42)
            mh.set_kernel_variant(variant)
            med, mn = timed(mh, 1000, 100)
            alg = C * 2 * 4 * (1000 + 2)
            print(json.dumps({"config": 2, "what": "Gaussian2D MH 65536 chains f32 run(1000,100)", "variant": variant,
                              "kernel_ms": med, "kernel_ms_min": mn, "samples_per_s": C * 1000 / (med * 1e-3),
                              "steps_per_s": C * 1100 / (med * 1e-3), "hbm_GBps": alg / (med * 1e-3) / 1e9,
                              "hbm_frac": alg / (med * 1e-3) / 8e12}))
    if "hmc" not in skip:
        init = init_with_seed(C, 3, 42, np.float32)
        for variant in (0, 2):
            h = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42).set_kernel_variant(variant)
            med, mn = timed(h, 400, 50)
            alg = C * 3 * 4 * 402
            print(json.dumps({"config": 3, "what": "RosenbrockND3 HMC 65536 chains f32 L=10 run(400,50)", "variant": variant,
                              "kernel_ms": med, "kernel_ms_min": mn, "samples_per_s": C * 400 / (med * 1e-3),
                              "hbm_GBps": alg / (med * 1e-3) / 1e9, "hbm_frac": alg / (med * 1e-3) / 8e12}))
        # PCIe-inclusive: the boundary hands back a HOST array (never the bench `value`)
        h = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42)
        h.run(400, 50, accept_counts=False)
        t0 = time.perf_counter()
        for _ in range(3):
            h.run(400, 50, accept_counts=False)
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({"config": 3, "what": "same, sample returned to a host numpy array (PCIe-inclusive, pageable)",
                          "wall_ms": dt * 1e3, "samples_per_s": C * 400 / dt}))
        # scaling in the number of chains (occupancy): one launch each
        for c in (16384, 65536, 262144, 1048576):
            initc = init_with_seed(c, 3, 42, np.float32)
            hc = HMC(RosenbrockND(3), initc, 0.032, 10).set_seed(42)
            med, mn = timed(hc, 100, 20, reps=5)
            print(json.dumps({"config": 3, "what": "chain-count sweep run(100,20)", "chains": c, "kernel_ms": med,
                              "chain_iters_per_s": c * 120 / (med * 1e-3)}))
    if "stats" not in skip:
        init = init_with_seed(C, 3, 42, np.float32)
        h = HMC(RosenbrockND(3), init, 0.032, 10).set_seed(42)
        t = h.run(400, 50, to="torch")
        torch.cuda.synchronize()
        S.split_rhat_mean_ess(t)
        t0 = time.perf_counter()
        for _ in range(5):
            rhat, ess = S.split_rhat_mean_ess(t)
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"what": "split_rhat_mean_ess [65536,400,3] f32 (device sample -> host rhat/ess)", "wall_ms": dt * 1e3,
                          "read_GBps": C * 400 * 3 * 4 / dt / 1e9, "ess": ess.tolist(), "rhat": rhat.tolist()}))
    if "nuts" not in skip:
        g = GaussianND.ill_conditioned(32, 1e4, 7)
        n = args.nuts_chains
        init = init_with_seed(n, 32, 42) * 0.1
        s = NUTS(g, init, 0.8, mode=2).set_seed(42).set_max_depth(10)
        if args.nuts_variant >= 0:
            s.set_kernel_variant(args.nuts_variant)
        if args.nuts_compaction_start >= 0 or args.nuts_groups > 0:
            s.set_compaction(args.nuts_compaction_start if args.nuts_compaction_start >= 0 else 5, args.nuts_groups)
        t0 = time.perf_counter()
        out = s.run_progress(args.nuts_draws, args.nuts_warmup, to="torch")[0]
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kms = s.timing()["kernel_ms"]
        lf = s.leapfrog_counts()
        hist = s.depth_histogram()
        flat = out.reshape(-1, 32).double().cpu().numpy()
        cov = np.cov(flat.T)
        true_cov = np.linalg.inv(g.precision)
        rel = np.abs(np.diag(cov) - np.diag(true_cov)) / np.diag(true_cov)
        rhat, ess = S.split_rhat_mean_ess(out)
        print(json.dumps({"config": 5, "what": f"NUTS 32-D Gaussian cond 1e4 f64, {n} chains, {args.nuts_warmup}+{args.nuts_draws}, max depth 10",
                          "variant": s.kernel_variant, "compaction_start": args.nuts_compaction_start, "groups": args.nuts_groups, "kernel_ms": kms, "wall_s": wall, "leapfrogs_total": int(lf.sum()),
                          "leapfrog_steps_per_s": float(lf.sum()) / (kms * 1e-3),
                          "samples_per_s": n * args.nuts_draws / (kms * 1e-3), "depth_hist": hist.tolist(),
                          "eps_mean": float(s.adapt_state()["epsilon"].mean()), "var_rel_err_max": float(rel.max()),
                          "ess_min": float(ess.min()), "rhat_conv_max": float((1 / rhat).max())}))


if __name__ == "__main__":
    main()
