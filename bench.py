#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: samples/sec (all chains) of 3-D Rosenbrock HMC.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]; configs[3] = the same per GPU at N = 8): RosenbrockND D = 3, 65 536 chains per
GPU, step size 0.032, 10 leapfrog steps, f32, init = init_with_seed(C, 3, 42), seed 42 (examples/minimal_hmc.rs).
One STEP = one `HMC::run(n_collect = 400, n_discard = 50)` of every chain (hmc.rs:137-158) through the C ABI, the
sample [C, 400, 3] written to HBM by the sampling kernel.  Chains are independent, so N GPUs run N shards of
65 536 chains with no data-path collective (weak scaling); after the timed region the split-R-hat / ESS reduction
exchanges only sufficient statistics (RCCL all-gather + all-reduce) and is reported next to the headline number.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on
the launch stream) and, at N = 1, `cpu_baseline` (the reference-ordered CPU restatement timed on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3   # vector f32 peak (v_pk_fma_f32)
SCLK_GHZ = 2.4                  # peak engine clock (MI355X_MICROARCH.md)

C_PER_GPU = 65536
DIM = 3
STEP_SIZE = 0.032
N_LEAPFROG = 10
N_COLLECT = 400
N_DISCARD = 50
SEED = 42
# algorithmic flops of one HMC transition at D = 3, L = 10 (48 per leapfrog step + momentum / energy / accept)
FLOP_PER_ITER = 48 * N_LEAPFROG + 16 + 30 + 4 * DIM + 25


CPU_CHAINS_PER_THREAD = 1024


def _cpu_sweep(kind, target, dim, scale, n_leapfrog, n_collect, n_discard, seconds_target, unit_per_run):
    """One workload on 1 / 8 / 64 / all usable hardware threads through oracle/cpu_bench.c: one spawn per measurement,
    every thread a private sampler over its own 1024 chains (state, noise and sample buffers private, first touched by
    the owner), `reps` x run(n_collect, n_discard) behind a common gate.  `unit_per_run` = units (samples) one thread
    produces per run.  Returns (sweep list, single-thread rate)."""
    import numpy as np

    import oracle as O

    cores = O.usable_cpus()
    cpt = CPU_CHAINS_PER_THREAD
    counts = sorted({n for n in (1, 8, 64, cores) if n <= cores})
    init_all = O.init_with_seed(cpt * cores, dim, SEED, np.float32)
    # calibrate on one thread, one run
    wall1, _, _ = O.cpu_bench(kind, target, init_all[:cpt], cpt, 1, scale, n_leapfrog, np.float32, n_collect, n_discard, reps=1, seed=SEED)
    share = seconds_target / (len(counts) + 1)  # the all-threads point gets a double share
    sweep = []
    for n in counts:
        budget = share * (2 if n == cores else 1)
        reps = max(1, int(round(budget / max(wall1, 1e-6))))
        wall, per, _ = O.cpu_bench(kind, target, init_all[:cpt * n], cpt, n, scale, n_leapfrog, np.float32, n_collect, n_discard,
                                   reps=reps, seed=SEED)
        sweep.append({"threads": n, "reps": reps, "wall_s": wall, "value": n * reps * unit_per_run / wall,
                      "per_thread": reps * unit_per_run / float(per.mean()),
                      "slowest_thread_s": float(per.max()), "fastest_thread_s": float(per.min())})
    return sweep, cores


def cpu_baseline(seconds_target: float = 15.0) -> dict:
    """The reference-ordered CPU restatement (oracle/mh_hmc.c: hmc.rs:304-431 op for op; xoshiro256++ / ziggurat
    noise) timed on this host's usable hardware threads on a bounded sample of the workload: what rayon's
    `par_iter_mut` over chains would do (core.rs:178-182) -- every thread carries its own block of 1024 chains through
    whole runs, nothing shared while the clock runs.  More parallelism than the reference's HMC extracts (its driver is
    single-threaded around burn tensor ops, hmc.rs:137-158), so the baseline errs on the CPU's side."""
    import oracle as O

    cpt = CPU_CHAINS_PER_THREAD
    sweep, cores = _cpu_sweep(O.CPU_BENCH_HMC, O.rosenbrock_nd(DIM), DIM, STEP_SIZE, N_LEAPFROG, N_COLLECT, N_DISCARD,
                              seconds_target, cpt * N_COLLECT)
    top, one = sweep[-1], sweep[0]
    ratio = top["per_thread"] / one["per_thread"]
    res = {
        "value": top["value"],
        "unit": "samples/s",
        "cores": top["threads"],
        "kind": "port",
        "sample": f"{top['reps']} x HMC::run({N_COLLECT},{N_DISCARD}) of {cpt} chains on each of {top['threads']} threads (same target, eps, L, "
                  f"f32) in {top['wall_s']:.1f} s; restatement of mini-mcmc's CPU path (the Rust reference cannot be built on this image)",
        "cores_how": f"len(sched_getaffinity) capped by the cgroup CPU quota = {cores} (os.cpu_count() = {os.cpu_count()})",
        "leapfrog_steps_per_s": top["value"] / N_COLLECT * (N_COLLECT + N_DISCARD) * N_LEAPFROG,
        "single_thread_samples_per_s": one["value"],
        "per_thread_samples_per_s_at_all_threads": top["per_thread"],
        "per_thread_over_single_thread": ratio,
        "threads_sweep": sweep,
    }
    if ratio < 0.5:
        res["per_thread_note"] = ("per-thread rate at all threads is under half the single-thread rate: the usable 'CPUs' are hardware "
                                  "threads (two per core share one FPU / L1) and the all-core clock is below the single-core boost; "
                                  "see threads_sweep for where the rate per thread drops")
    return res


def _cpu_side_baselines(seconds_each: float) -> dict:
    """BASELINE.md section 3's other CPU rows, on bounded samples, through the same harness (oracle/cpu_bench.c):
    config 2 = the reference-faithful MH restatement (oracle/mh_hmc.c with the reference's own stream: xoshiro256++ /
    ziggurat, every chain holding a clone of the proposal generator (quirk Q1), D + 1 normals per proposal (Q2), logp
    recomputed, q-terms kept -- metropolis_hastings.rs:150-153, 303-315; distributions.rs:364-372), 1024 chains per thread;
    config 5 = the recursive NUTS restatement (oracle/nuts.c, nuts.rs:550-946) on the same 32-D target and the same
    500 warm-up + 500 draws as the GPU leg, 32 chains per thread."""
    import numpy as np

    import oracle as O

    out = {}
    cpt = CPU_CHAINS_PER_THREAD
    sweep, cores = _cpu_sweep(O.CPU_BENCH_MH, O.gaussian2d([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), 2, 1.0, 0, 1000, 100,
                              seconds_each, cpt * 1000)
    top = sweep[-1]
    out["config2_mh"] = {"value": top["value"], "unit": "samples/s", "cores": top["threads"], "kind": "port",
                         "single_thread_samples_per_s": sweep[0]["value"], "threads_sweep": sweep,
                         "sample": f"{top['reps']} x run(1000,100) of {cpt} chains on each of {top['threads']} threads in {top['wall_s']:.1f} s, "
                                   "reference-faithful MH (Q1, Q2 kept)"}
    from mini_mcmc_amd.distributions import GaussianND

    g = GaussianND.ill_conditioned(32, 1e4, 7)
    tgt = O.gaussian_nd(g.precision)
    per_thread = 8  # 8 chains x 1000 transitions x ~185 leapfrog steps per thread: a few seconds
    init = O.init_with_seed(per_thread * cores, 32, SEED) * 0.1
    res5 = []
    for n in sorted({1, cores}):
        wall, per, lf = O.cpu_bench(O.CPU_BENCH_NUTS, tgt, init[:per_thread * n], per_thread, n, 0.8, 0, np.float64, 500, 500, reps=1,
                                    seed=SEED, nuts_mode=2, nuts_max_depth=10)
        res5.append({"threads": n, "wall_s": wall, "value": lf / wall, "draws_per_s": per_thread * n * 500 / wall})
    top5 = res5[-1]
    out["config5_nuts"] = {"value": top5["value"], "unit": "leapfrog-steps/s", "cores": top5["threads"], "kind": "port",
                           "draws_per_s": top5["draws_per_s"], "single_thread_leapfrog_steps_per_s": res5[0]["value"], "threads_sweep": res5,
                           "sample": f"{per_thread} chains on each of {top5['threads']} threads x (500 warm-up + 500 draws) in {top5['wall_s']:.1f} s, "
                                     "recursive NUTS restatement, f64, max depth 10"}
    return out


ROUND_TAG = "r6"  # profiles/*_kernel_*.json are quoted only when written this round (their "round" starts with this)


def _profile_json(name: str, kernel_name: str, variant: int):
    """profiles/<name> if it describes the kernel being timed: written this round, same kernel, same source fingerprint.
    Returns (dict, source label) or (None, reason)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        j = json.load(open(path))
    except (OSError, ValueError):
        return None, f"profiles/{name}: absent"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_fingerprint import sampling_kernel_sources_sha256

        fp = sampling_kernel_sources_sha256()
    except Exception:
        fp = None
    if not str(j.get("round", "")).startswith(ROUND_TAG):
        return None, f"profiles/{name}: written in round {j.get('round')!r}, not {ROUND_TAG}: not quoted"
    if kernel_name not in str(j.get("kernel", "")) or j.get("variant") != variant:
        return None, f"profiles/{name}: describes {j.get('kernel')!r}, not the kernel timed here: not quoted"
    if fp is None or j.get("sources_sha256") != fp:
        return None, f"profiles/{name}: the kernel's sources have changed since it was measured: not quoted"
    return j, f"profiles/{name} ({j.get('round')}): {j.get('source')} -- a profile of this kernel (same sources), not of this run"


def side_configs(dev, cpu_seconds: float = 0.0) -> dict:
    """BASELINE.json's other single-GPU configurations, measured once each after the headline run and reported as
    extra keys (they are parity-test cases, not bench lines): config 2 (Gaussian2D MH, 65 536 chains, f32,
    run(1000, 100)), config 5 (NUTS, 32-D Gaussian of condition number 1e4, 65 536 chains, f64, 200 + 100,
    tree-depth compaction), and config 3's long run (run(1000, 200): the ESS of a longer sample).  Kernel
    time by HIP events on the launch stream.  With cpu_seconds > 0 the CPU restatements of configs 2 and 5 are timed
    too.  Never fails the bench."""
    out = {}
    try:
        import numpy as np
        import torch

        from mini_mcmc_amd import stats as S
        from mini_mcmc_amd.core import init_with_seed
        from mini_mcmc_amd.distributions import Gaussian2D, GaussianND, IsotropicGaussian, RosenbrockND
        from mini_mcmc_amd.hmc import HMC
        from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
        from mini_mcmc_amd.nuts import NUTS

        mh = MetropolisHastings(Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), IsotropicGaussian(1.0),
                                init_with_seed(C_PER_GPU, 2, SEED, np.float32), device=dev.index or 0).seed(SEED)
        # measured like the headline: launches back to back into one resident buffer (stream time per launch, launch gaps
        # included), then a few launches bracketed by HIP events one at a time (the device idles between those: slower)
        import ctypes as C

        from mini_mcmc_amd import _lib as L

        lib = L.lib()
        mh_out = torch.empty((C_PER_GPU, 1000, 2), dtype=torch.float32, device=dev)
        mh_stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def mh_step():
            L.check(lib.mmcmc_mh_run(mh._h, 1000, 100, C.c_void_p(mh_out.data_ptr()), 1, None, mh_stream), "mmcmc_mh_run")

        mh.enable_timing(False)
        for _ in range(20):
            mh_step()
        torch.cuda.synchronize()
        n_b2b = 300
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n_b2b):
            mh_step()
        e1.record()
        torch.cuda.synchronize()
        k = (time.perf_counter() - t0) / n_b2b * 1e3
        kb = e0.elapsed_time(e1) / n_b2b  # as the headline's roofline.kernel_ms: HIP events on the launch stream around back-to-back launches
        mh.enable_timing(True)
        ms = []
        for _ in range(10):
            mh_step()
            ms.append(mh.timing()["kernel_ms"])
        ke = float(np.median(ms))
        mh_bytes = C_PER_GPU * 2 * 4 * (1000 + 2)
        # `kernel_ms` / `hbm_frac` are again what they were up to BENCH_r04 -- the median of launches timed one at a
        # time (comparable across rounds) -- and the throughput figure of launches queued back to back (the method of the
        # headline's roofline.kernel_ms, where one launch's tail overlaps the next one's head) has keys of its own
        out["config2_mh"] = {"kernel_ms": ke, "kernel_ms_how": "median of 10 launches timed one at a time by HIP events on the launch stream (the definition of BENCH_r01..r04; BENCH_r05 quoted the back-to-back figure under this key)",
                             "kernel_ms_back_to_back": kb,
                             "kernel_ms_back_to_back_how": f"HIP events on the launch stream around {n_b2b} back-to-back launches / {n_b2b} (as roofline.kernel_ms of the headline; what rocprofv3's average over the run agrees with)",
                             "ms_per_run_back_to_back": k,
                             "samples_per_s": C_PER_GPU * 1000 / (k * 1e-3),
                             "algorithmic_bytes_per_launch": mh_bytes,
                             "hbm_frac": mh_bytes / (ke * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "hbm_frac_back_to_back": mh_bytes / (kb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "stream": "the MH sampler's paired f32 stream at D <= 2 (csrc/mm_rng.h, round 5: one Philox block per two transitions)"}
        del mh, mh_out
        # config 3, long run: 1000 collected after 200; ESS / R-hat of that sample
        h = HMC(RosenbrockND(DIM), init_with_seed(C_PER_GPU, DIM, SEED, np.float32), STEP_SIZE, N_LEAPFROG,
                device=dev.index or 0).set_seed(SEED)
        t = h.run(1000, 200, to="torch", accept_counts=False)
        torch.cuda.synchronize()
        k = float(h.timing()["kernel_ms"])
        S.split_rhat_mean_ess(t)  # first call at this shape: work buffers
        t0 = time.perf_counter()
        rhat, ess = S.split_rhat_mean_ess(t)
        st_ms = (time.perf_counter() - t0) * 1e3
        out["config3_long"] = {"run": "run(1000, 200)", "kernel_ms": k, "stats_ms": st_ms, "ess_min": float(ess.min()),
                               "split_rhat_max_conventional": float((1.0 / rhat).max()),
                               ("ess_per_s" if float((1.0 / rhat).max()) <= 1.05 else "ess_per_s_unconverged"): float(ess.min()) / ((k + st_ms) * 1e-3),
                               "samples_per_s": C_PER_GPU * 1000 / (k * 1e-3)}
        del h, t
        # an ESS/s that means something: (a) the reference's own converging HMC configuration (hmc.rs:576-787: 2-D Gaussian
        # mean [0, 1], cov [[4, 2], [2, 3]], eps 0.1, L 10, 500 burn-in + 1000 draws; its test asserts split R-hat in
        # [0.95, 1.05]) at 65 536 chains; (b) config 3's target with trajectories long enough to cross the banana
        # (eps 0.02, L 150; profiles/r4b_converged_probe.jsonl: eps 0.032 / L 10 stays at R-hat 1.6-1.7 after 100 000
        # transitions per chain within any 4000-draw window).  ess_per_s only where the conventional split R-hat <= 1.05.
        from mini_mcmc_amd.distributions import DiffableGaussian2D

        def converged_case(target, dim, eps, n_leap, n_burn, n_keep, what):
            hh = HMC(target, init_with_seed(C_PER_GPU, dim, SEED, np.float32), eps, n_leap, device=dev.index or 0).set_seed(SEED)
            tt = hh.run(n_keep, n_burn, to="torch", accept_counts=False)
            torch.cuda.synchronize()
            kk = float(hh.timing()["kernel_ms"])
            S.split_rhat_mean_ess(tt)
            t0 = time.perf_counter()
            rh, es = S.split_rhat_mean_ess(tt)
            sm = (time.perf_counter() - t0) * 1e3
            rmax = float((1.0 / rh).max())
            res = {"workload": what, "run": f"run({n_keep}, {n_burn})", "kernel_ms": kk, "stats_ms": sm, "ess_min": float(es.min()),
                   "split_rhat_max_conventional": rmax, "converged": rmax <= 1.05,
                   "samples_per_s": C_PER_GPU * n_keep / (kk * 1e-3),
                   "leapfrog_steps_per_s": C_PER_GPU * (n_keep + n_burn) * n_leap / (kk * 1e-3)}
            res["ess_per_s" if rmax <= 1.05 else "ess_per_s_unconverged"] = float(es.min()) / ((kk + sm) * 1e-3)
            return res

        out["config3_converged"] = converged_case(DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2, 0.1, 10, 500, 1000,
                                                  "hmc.rs:576-787's configuration (DiffableGaussian2D, eps 0.1, L 10) at 65536 chains, f32")
        out["config3_rosenbrock_long_trajectories"] = converged_case(RosenbrockND(DIM), DIM, 0.02, 150, 6000, 2000,
                                                                     "RosenbrockND D=3 (config 3's target), eps 0.02, L 150, 65536 chains, f32")
        # (c) round 5: the metric's own target CONVERGED -- NUTS in the reference's type split (f32 tensors, f64 scalars:
        # NUTS<f64, Autodiff<NdArray>>), per-chain dual-averaging step size, target_accept_p 0.98, 1000 + 1000, 65 536 chains
        # (profiles/r5b_converged_probe.jsonl: fixed-step HMC leaves chains stuck in the stiff tail -- var(x2) 1.2-1.5x the exact
        # value, R-hat 1.08-1.23 at any length tried --; NUTS at 0.8 under-explores it; at 0.95 / 0.98 R-hat is 1.028 / 1.019).
        # The sample's moments are checked against the exact ones (quadrature over x0: x1 | x0 and x2 | x1 are Gaussian).
        nr = NUTS(RosenbrockND(DIM), init_with_seed(C_PER_GPU, DIM, SEED), 0.98, mode=0, device=dev.index or 0).set_seed(SEED)
        tn = nr._run(1000, 1000, True, "torch")
        torch.cuda.synchronize()
        kn = float(nr.timing()["kernel_ms"])
        S.split_rhat_mean_ess(tn)
        t0 = time.perf_counter()
        rh, es = S.split_rhat_mean_ess(tn)
        sm = (time.perf_counter() - t0) * 1e3
        rmax = float((1.0 / rh).max())
        xd = tn.double()
        true_mean, true_var = [0.78522, 0.84746, 1.06837], [0.22937, 0.35018, 1.63898]
        mean = [float(v) for v in xd.mean(dim=(0, 1)).cpu()]
        var = [float(v) for v in xd.reshape(-1, DIM).var(dim=0).cpu()]
        lfn = float(nr.leapfrog_counts().sum())
        res_n = {"workload": "RosenbrockND D=3 (config 3's target), NUTS mode 0 (f32 tensors, f64 scalars), target_accept_p 0.98, "
                             "65536 chains, 1000 warm-up + 1000 draws", "kernel_ms": kn, "stats_ms": sm, "ess_min": float(es.min()),
                 "split_rhat_max_conventional": rmax, "converged": rmax <= 1.05, "samples_per_s": C_PER_GPU * 1000 / (kn * 1e-3),
                 "leapfrog_steps_per_s": lfn / (kn * 1e-3),
                 "posterior_mean": mean, "posterior_var": var, "exact_mean": true_mean, "exact_var": true_var,
                 "max_rel_moment_error": max(max(abs(mean[i] - true_mean[i]) / true_mean[i] for i in range(DIM)),
                                             max(abs(var[i] - true_var[i]) / true_var[i] for i in range(DIM)))}
        res_n["ess_per_s" if rmax <= 1.05 else "ess_per_s_unconverged"] = float(es.min()) / ((kn + sm) * 1e-3)
        out["config3_rosenbrock_converged"] = res_n
        del nr, tn, xd
        # (d) round 6: the metric's own target CONVERGED under the metric's own sampler.  One fixed (eps, L) does not mix on
        # RosenbrockND(3) (profiles/r5b_, r6m_converged_probe.jsonl: R-hat 1.06-1.75 per handle at eps 0.0097-0.03, the trajectory
        # length jittered alone 1.11); the reference's HMC has no jitter, but `HMC::new` takes any initial positions, so launch
        # k is a NEW handle on the last positions with eps_k ~ U(0.004, 0.016), L_k ~ U{100..400} and run(100, 0): 20 launches
        # discarded, 40 kept (hmc.run_chain_of_handles).  ESS/s over all 60 launches' kernel time + the diagnostics.  The band is
        # the one that held over four noise / schedule seeds (profiles/r6s_converged_hmc_robustness.jsonl: R-hat <= 1.003, moments
        # within 0.04 %); the first choice, eps in (0.01, 0.03), met the 1 % with one seed of four -- steps of 0.03 leave chains
        # stuck in the stiff tail for a whole launch.
        from mini_mcmc_amd.hmc import run_chain_of_handles

        t0 = time.perf_counter()
        th, info = run_chain_of_handles(RosenbrockND(DIM), init_with_seed(C_PER_GPU, DIM, SEED, np.float32), (0.004, 0.016), (100, 400),
                                        100, 20, 40, seed=SEED, device=dev.index or 0)
        wall_h = (time.perf_counter() - t0) * 1e3
        S.split_rhat_mean_ess(th)
        t0 = time.perf_counter()
        rh, es = S.split_rhat_mean_ess(th)
        sm = (time.perf_counter() - t0) * 1e3
        rmax = float((1.0 / rh).max())
        xd = th.double()
        mean = [float(v) for v in xd.mean(dim=(0, 1)).cpu()]
        var = [float(v) for v in xd.reshape(-1, DIM).var(dim=0).cpu()]
        err = max(max(abs(mean[i] - true_mean[i]) / true_mean[i] for i in range(DIM)), max(abs(var[i] - true_var[i]) / true_var[i] for i in range(DIM)))
        res_h = {"workload": "RosenbrockND D=3 (config 3's target), HMC f32, 65536 chains: 60 launches of run(100, 0), each a new HMC handle on "
                             "the last positions with eps ~ U(0.004, 0.016), L ~ U{100..400}; 20 launches discarded, 4000 draws per chain kept",
                 "kernel_ms": info["kernel_ms"], "wall_ms": wall_h, "stats_ms": sm, "launches": info["launches"], "ess_min": float(es.min()),
                 "split_rhat_max_conventional": rmax, "accept_rate": info["accept_rate"], "leapfrog_steps_per_s": info["leapfrogs"] / (info["kernel_ms"] * 1e-3),
                 "posterior_mean": mean, "posterior_var": var, "exact_mean": true_mean, "exact_var": true_var, "max_rel_moment_error": err,
                 "converged": bool(rmax <= 1.05 and err <= 0.01)}
        res_h["ess_per_s" if res_h["converged"] else "ess_per_s_unconverged"] = float(es.min()) / ((info["kernel_ms"] + sm) * 1e-3)
        res_h["ess_per_s_wall_clock"] = float(es.min()) / ((wall_h + sm) * 1e-3)
        out["config3_hmc_converged"] = res_h
        del th, xd
        g = GaussianND.ill_conditioned(32, 1e4, 7)
        nuts = NUTS(g, init_with_seed(C_PER_GPU, 32, SEED) * 0.1, 0.8, mode=2, device=dev.index or 0).set_seed(SEED).set_max_depth(10)
        nuts._run(500, 500, True, "torch")
        torch.cuda.synchronize()
        tm = nuts.timing()
        k = float(tm["kernel_ms"])
        lf = float(nuts.leapfrog_counts().sum())
        # issue: (MFMA + vector-issue cycles) / wave cycles of the scheduler kernel from this round's SQ counters (tools/pmc_nuts.sh ->
        # profiles/nuts5_kernel_counters.json), quoted only while that summary is of this round
        nc5 = None
        try:
            j5 = json.load(open(os.path.join(ROOT, "profiles", "nuts5_kernel_counters.json")))
            if str(j5.get("round", "")).startswith(ROUND_TAG):
                nc5 = {kk: j5[kk] for kk in ("round", "source", "issue_frac", "mfma_busy_over_wave_cycles", "valu_active_over_wave_cycles",
                                             "wait_any_over_wave_cycles", "valu_instructions_per_leaf_iteration") if kk in j5}
        except (OSError, ValueError):
            pass
        out["config5_nuts"] = {"schedule": "500 warm-up + 500 draws (SURVEY 8d)", "kernel_ms": k, "n_launches": int(tm["n_launches"]),
                               "leapfrog_steps_per_s": lf / (k * 1e-3),
                               "draws_per_s": C_PER_GPU * 500 / (k * 1e-3), "kernel_variant": nuts.kernel_variant,
                               "f64_mfma_frac": lf * 2 * 32 * 32 / (k * 1e-3) / 78.6e12, "issue": nc5}
        del nuts
        # the same at round 4's shorter schedule (200 + 100), the figure VERDICT r4 set its <= 400 ms bar on
        nuts = NUTS(g, init_with_seed(C_PER_GPU, 32, SEED) * 0.1, 0.8, mode=2, device=dev.index or 0).set_seed(SEED).set_max_depth(10)
        nuts._run(100, 200, True, "torch")
        torch.cuda.synchronize()
        k = float(nuts.timing()["kernel_ms"])
        lf = float(nuts.leapfrog_counts().sum())
        out["config5_nuts_200_100"] = {"kernel_ms": k, "leapfrog_steps_per_s": lf / (k * 1e-3), "f64_mfma_frac": lf * 2 * 32 * 32 / (k * 1e-3) / 78.6e12}
        del nuts
        if cpu_seconds > 0:
            out["cpu_baselines"] = _cpu_side_baselines(cpu_seconds)
    except Exception as e:  # pragma: no cover -- side numbers must not take the headline down
        out["error"] = repr(e)
    return out


EXCHANGE_TEXT = {1: "rccl (ncclAllGather + ncclAllReduce inside libmmcmc.so)", 0: "host (a device is listed twice)",
                 -1: "host FALLBACK: no RCCL library could be loaded", -2: "host FALLBACK: ncclCommInitAll failed"}


def group_verdict(devices, pci_bus_ids, exchange_status: int, rccl_ranks: int):
    """(data, exit_code, problems) of a device-group line.  A run over N DISTINCT devices is a scaling measurement: its
    diagnostics must have travelled over RCCL with N ranks, and the N devices must be N different GPUs by PCI
    bus id -- otherwise the line is still printed, `data` says in capitals what happened, and the process exits non-zero
    so that no harness can take the line for a measurement.  A device listed twice (the one-GPU rehearsal,
    MMCMC_BENCH_GROUP_DEVICES=0,0) goes through the host by design: exit 0, and `data` says it is a rehearsal."""
    n = len(devices)
    distinct = len(set(devices)) == n
    problems = []
    if distinct:
        if exchange_status != 1:
            problems.append("DIAGNOSTICS FELL BACK TO THE HOST (" + EXCHANGE_TEXT.get(exchange_status, f"status {exchange_status}") + ")")
        if rccl_ranks != n:
            problems.append(f"RCCL COMMUNICATOR HAS {rccl_ranks} RANKS FOR {n} DEVICES")
        if len(set(pci_bus_ids)) != n:
            problems.append("TWO DEVICE ORDINALS SHARE A PCI BUS ID: " + ",".join(pci_bus_ids))
    data = "synthetic"
    if not distinct:
        data += "; REHEARSAL: a device is listed twice, its shards share one GPU and the statistics travel through the host by design -- not a scaling measurement"
    if problems:
        data += "; " + "; ".join(problems) + " -- NOT A VALID SCALING MEASUREMENT"
    return data, (3 if problems else 0), problems


def main_group(args) -> None:
    """--group: the same headline through the in-library device group -- one process, N devices (MMCMC_BENCH_GROUP_DEVICES =
    comma-separated device list overrides 0..N-1, e.g. "0,0" rehearses two shards on one GPU through the host exchange).
    A step = mmcmc_hmc_group_run of all N x 65 536 chains: every shard's kernel enqueued on its own device and stream, no
    wait between steps (as at N = 1); mmcmc_hmc_group_sync closes the timed region.  The diagnostics afterwards go through
    ncclAllGather / ncclAllReduce inside the library, and the line says which path they took."""
    import numpy as np
    import torch

    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.group import HMCGroup

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    devs = os.environ.get("MMCMC_BENCH_GROUP_DEVICES")
    devices = [int(d) for d in devs.split(",")] if devs else list(range(args.gpus))
    n = len(devices)
    init = init_with_seed(C_PER_GPU * n, DIM, SEED, np.float32)
    g = HMCGroup(RosenbrockND(DIM), init, STEP_SIZE, N_LEAPFROG, devices=devices).set_seed(SEED)

    exch_status, rccl_ranks = g.exchange()  # decided (and the RCCL communicators made) when the group was created

    def step():
        # nothing goes back to the host: every shard's launch is ENQUEUED on that shard's stream and the call returns
        # (include/mmcmc.h) -- the same back-to-back queueing as the N = 1 line's mmcmc_hmc_run on torch's stream
        g.run_async(N_COLLECT, N_DISCARD)

    t_pre, pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(100):
            step()
        g.sync()
        pre += 100
    for _ in range(args.warmup):
        step()
    g.sync()
    t0 = time.perf_counter()
    g.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms_per_device = g.timer_stop() / args.steps  # HIP events on every shard's own stream around the timed region
    g.sync()
    dt = time.perf_counter() - t0
    g.split_rhat_mean_ess()
    ts = time.perf_counter()
    phases = {"local_partials_ms": 0.0, "exchange_ms": 0.0, "finish_ms": 0.0}
    for _ in range(5):
        rhat, ess = g.split_rhat_mean_ess()
        for k, v in g.stats_phases().items():  # where the call's time goes: every shard's local statistics | the
            phases[k] += v / 5                 # collectives (+ cross-chain sums on shard 0) | the finish
    stats_s = (time.perf_counter() - ts) / 5
    pci = g.pci_bus_ids()
    data, exit_code, problems = group_verdict(devices, pci, g.exchange_status, rccl_ranks)
    samples = float(args.steps) * C_PER_GPU * n * N_COLLECT
    ms = dt / args.steps * 1e3
    alg_bytes = C_PER_GPU * DIM * 4 * (N_COLLECT + 2)
    rhat_max = float((1.0 / rhat).max())
    shards_on = {d: devices.count(d) for d in devices}
    k_ms = max(float(kernel_ms_per_device[i]) / shards_on[devices[i]] for i in range(n))
    res = {
        "metric": "samples/sec (all chains), 3D Rosenbrock HMC", "value": samples / dt, "unit": "samples/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": data,
        "config": {"workload": "BASELINE.json configs[2] per device (configs[3] at 8): RosenbrockND D=3 HMC, 65536 chains/GPU, "
                               "eps=0.032, L=10, f32, run(400, 50) per step", "chains_per_gpu": C_PER_GPU, "dim": DIM,
                   "n_leapfrog": N_LEAPFROG, "n_collect": N_COLLECT, "n_discard": N_DISCARD, "devices": devices,
                   "parallelism": f"in-library device group x{n}: one process, one stream per device, launches enqueued back to back, no data-path collective"},
        "leapfrog_steps_per_s": float(args.steps) * C_PER_GPU * n * (N_COLLECT + N_DISCARD) * N_LEAPFROG / dt,
        "ess_min": float(ess.min()), "split_rhat_max_conventional": rhat_max,
        "stats_ms": stats_s * 1e3, "stats_phases_ms": phases,
        "diagnostics_exchange": EXCHANGE_TEXT[g.exchange_status],
        "diagnostics_exchange_decided_at_create": exch_status,
        "preroll": {"seconds": args.preroll_seconds, "steps": pre},
        "rccl_ranks": rccl_ranks, "pci_bus_ids": pci, "problems": problems,
        "kernel_ms_per_device": [float(x) for x in kernel_ms_per_device],
        "kernel_ms_spread_max_over_min": float(max(kernel_ms_per_device) / max(min(kernel_ms_per_device), 1e-9)),
        # the kernel's launch duration per device: HIP events on each shard's own stream around the timed region / steps (with a
        # device listed twice its two streams share the device, so each stream's figure covers both shards' launches)
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "launches_per_step": n, "kernel_ms": k_ms,
                     "kernel_ms_how": "slowest device: HIP events on the shard's stream around the timed region / steps / shards on that device",
                     "whole_job": {"achieved": alg_bytes * n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * len(set(devices)),
                                   "how": "all shards' algorithmic bytes / ms_per_step (wall clock, host loop included)"},
                     "note": "vector-instruction issue binds the kernel (N = 1 line: fp32, issue)"},
    }
    conv = rhat_max <= 1.05
    res["ess_per_s" if conv else "ess_per_s_unconverged"] = float(ess.min()) * args.steps / (dt + stats_s * args.steps)
    print(json.dumps(res), flush=True)
    if exit_code:
        # printed first (the line is evidence of WHAT went wrong), then a non-zero exit: a scaling run cannot take the
        # host path, a short communicator or one GPU under two ordinals and still look like a measurement
        g.close()
        sys.exit(exit_code)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements of configs 2 and 5")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--preroll-seconds", type=float, default=1.0,
                    help="untimed pre-roll of step() before the counted warm-up (clock ramp); 0 disables it")
    ap.add_argument("--variant", type=int, default=5, help="kernel variant: 5 noise waves + transition waves, four waves per SIMD (default), 2 one wave per SIMD with paired + pipelined noise, 0 plain")
    ap.add_argument("--group", action="store_true",
                    help="ONE process drives all --gpus N devices through the library's device group (mmcmc_hmc_group_*: one "
                         "host thread and stream per device, RCCL inside the library for the diagnostics) instead of one "
                         "process per GPU under torch.distributed.run; same workload, same JSON line")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time sampler and diagnostics as a two-stream pipeline (a measured negative result, DESIGN.md 5.2: "
                         "slower than back to back; off by default so that a profiled run holds only undisturbed launches of the sampling kernel)")
    args = ap.parse_args()

    import numpy as np
    import torch

    # decided before anything touches a GPU (never by re-exec): --group, or a plain `python bench.py --gpus N` with N > 1
    # and no launcher environment, runs ONE process over N devices through the library's device group (RCCL inside
    # libmmcmc.so); under torch.distributed.run (WORLD_SIZE set) it is one process per GPU
    if args.group or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        return main_group(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         "or without a launcher for the in-library device group")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    # MMCMC_BENCH_DRYRUN_ONE_DEVICE=1: rehearsal of the N > 1 path on a box with ONE GPU (every rank on device 0, gloo for
    # the barrier and the diagnostics' reductions, since RCCL refuses one device twice); the line it prints says so and is
    # not a measurement
    dryrun = distributed and os.environ.get("MMCMC_BENCH_DRYRUN_ONE_DEVICE") == "1"
    if dryrun:
        local_rank = 0
    # a launcher that shows every rank ONE device (HIP_VISIBLE_DEVICES per rank) leaves ordinal 0 only: take the ordinal modulo
    # what this process sees -- whether the ranks really sit on different GPUs is decided below from the PCI bus ids
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd import stats as S

    # every rank owns global chains [rank*C, (rank+1)*C); the init stream is the reference's, sliced per rank
    init_all = init_with_seed(C_PER_GPU * world, DIM, SEED, np.float32)
    init = np.ascontiguousarray(init_all[rank * C_PER_GPU:(rank + 1) * C_PER_GPU])
    sampler = HMC(RosenbrockND(DIM), init, STEP_SIZE, N_LEAPFROG, device=local_rank).set_seed(SEED)
    sampler.set_chain_offset(rank * C_PER_GPU)
    sampler.set_kernel_variant(args.variant)

    out = torch.empty((C_PER_GPU, N_COLLECT, DIM), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    import ctypes as C
    from mini_mcmc_amd import _lib as L

    lib = L.lib()

    def step():
        st = lib.mmcmc_hmc_run(sampler._h, N_COLLECT, N_DISCARD, out.data_ptr(), 1, None, C.c_void_p(stream))
        L.check(st, "mmcmc_hmc_run")

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    sampler.enable_timing(False)  # no event packets between the back-to-back launches of the timed region
    # Untimed pre-roll of fixed DURATION before the counted warm-up: a process that has just initialised the GPU runs its
    # first milliseconds at ramping clocks, and `--steps 20 --warmup 5` is a 5 ms window -- without this the line depends
    # on who runs it (106 G by the driver's arguments against 127-138 G at 300 steps).  Same step(), same
    # buffers; `warmup` and `steps` are reported as passed and the timed region is untouched.
    preroll_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        preroll_steps += 100
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events on the launch stream (torch's current stream IS the stream the kernels are launched on) around the
    # timed region: elapsed / steps = the sampling kernel's average launch duration over exactly the launches that are
    # timed (back to back, one kernel per step)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms_timed_region = ev0.elapsed_time(ev1) / args.steps
    sampler.enable_timing(True)
    # the same kernel one launch at a time (the device idles between launches: slower clocks), reported beside it
    kernel_ms = []
    for _ in range(min(args.steps, 10)):
        step()
        kernel_ms.append(sampler.timing()["kernel_ms"])
    barrier()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    # every rank's own kernel time and the PCI bus id of the GPU it really ran on: rank 0 prints both, and refuses (non-zero
    # exit after printing) a launcher run in which two ranks shared a GPU or the process group is not RCCL
    pci_buf = C.create_string_buffer(32)
    L.check(lib.mmcmc_device_pci_bus_id(local_rank, pci_buf, 32), "mmcmc_device_pci_bus_id")
    mine = {"rank": rank, "device": local_rank, "pci_bus_id": pci_buf.value.decode(), "kernel_ms": float(kernel_ms_timed_region)}
    per_rank = [mine]
    if distributed:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # diagnostics on the last sample (outside the timed region): RCCL all-gather / all-reduce of sufficient statistics
    def diagnostics():
        return S.split_rhat_mean_ess_distributed(out) if distributed else S.split_rhat_mean_ess(out)

    diagnostics()  # untimed first call (work buffers, code objects), like the sampler's warm-up steps
    barrier()
    N_DIAG = 10  # the reduction is sub-millisecond: average a few calls
    ts = time.perf_counter()
    for _ in range(N_DIAG):
        rhat, ess = diagnostics()
    barrier()
    stats_s = (time.perf_counter() - ts) / N_DIAG

    # the same work as a pipeline (N = 1): while the sampler writes run k + 1 into one buffer, the statistics of run k are
    # reduced from the other one on a second stream -- the diagnostics leave the critical path
    pipe_ms = None
    if not distributed and args.pipelined:
        out2 = torch.empty_like(out)
        bufs = (out, out2)
        s_stats = torch.cuda.Stream(dev)
        done = [torch.cuda.Event(), torch.cuda.Event()]

        def step_into(buf):
            st = lib.mmcmc_hmc_run(sampler._h, N_COLLECT, N_DISCARD, buf.data_ptr(), 1, None, C.c_void_p(stream))
            L.check(st, "mmcmc_hmc_run")

        sampler.enable_timing(False)
        n_pipe = min(args.steps, 50)
        for timed in (False, True):
            barrier()
            tp = time.perf_counter()
            for k in range(n_pipe if timed else 3):
                step_into(bufs[k % 2])
                done[k % 2].record()
                if k > 0:
                    with torch.cuda.stream(s_stats):
                        s_stats.wait_event(done[(k - 1) % 2])
                        S.split_rhat_mean_ess(bufs[(k - 1) % 2])  # returns when the statistics of run k - 1 are on the host
            with torch.cuda.stream(s_stats):
                s_stats.wait_event(done[(n_pipe - 1) % 2 if timed else 0])
                rhat_p, ess_p = S.split_rhat_mean_ess(bufs[(n_pipe - 1) % 2 if timed else 0])
            barrier()
            if timed:
                pipe_ms = (time.perf_counter() - tp) / n_pipe * 1e3
        sampler.enable_timing(True)

    exit_code = 0
    if rank == 0:
        backend = dist.get_backend() if distributed else None
        if distributed:
            # one rank per GPU is the contract of this path, whatever ordinal each process calls its GPU: N ranks must show N
            # different PCI bus ids
            data, exit_code, problems = group_verdict([0] * world if dryrun else list(range(world)),
                                                      [r["pci_bus_id"] for r in per_rank], 1 if backend == "nccl" else 0,
                                                      dist.get_world_size() if backend == "nccl" else 0)
            if dryrun:
                data = "synthetic; DRY RUN: all ranks on one device over gloo, not a measurement"
        else:
            data, problems = "synthetic", []
        samples = float(args.steps) * C_PER_GPU * world * N_COLLECT
        ms_per_step = dt_max / args.steps * 1e3
        k_ms = float(kernel_ms_timed_region)
        alg_bytes = C_PER_GPU * DIM * 4 * (N_COLLECT + 2)  # sample written once + state loaded and stored once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        iters = C_PER_GPU * (N_COLLECT + N_DISCARD)
        valu_tflops = iters * FLOP_PER_ITER / (k_ms * 1e-3) / 1e12
        # roofline = the contract's figure: algorithmic bytes per launch / the kernel's launch duration (HIP events
        # over the timed region) / HBM peak.  What actually binds this kernel is vector-instruction issue (12 algorithmic
        # bytes against ~250 vector instructions per transition of 64 chains): `fp32` and `issue` say how close
        # to THOSE limits it runs.  Counter-derived figures are quoted from profiles/*.json only while the summary still
        # describes this kernel: same kernel name, this round's tag, same sha256 of the kernel's sources
        # (tools/summarize_pmc.py writes all three; nothing is assumed when they do not match).
        kernel_name = {0: "mm_run_kernel<float, mm_target<float, 4, 3>, 1, 0,", 1: "mm_run_kernel<float, mm_target<float, 4, 3>, 1, 10,",
                       2: "mm_run_kernel<float, mm_target<float, 4, 3>, 1, 10,",
                       5: "mm_run_split_kernel<float, mm_target<float, 4, 3>, 1, 10,"}[args.variant]
        traffic, traffic_source = _profile_json("hmc_kernel_traffic.json", kernel_name, args.variant)
        issue_counters, _ = _profile_json("hmc_kernel_counters.json", kernel_name, args.variant)
        issue = None
        if issue_counters and issue_counters.get("double_slot_share") is not None:
            valu_per_transition = float(issue_counters["valu_instructions_per_transition_of_64_chains"])
            double_slot_share = float(issue_counters["double_slot_share"])
            slots_per_launch = valu_per_transition * (1.0 + double_slot_share) * (C_PER_GPU / 64) * (N_COLLECT + N_DISCARD)
            issue_peak = 1024 * SCLK_GHZ * 1e9 / 2.0  # MI355X_MICROARCH.md: one wave64 vector instruction per 2 cycles per SIMD-32
            issue_achieved = slots_per_launch / (k_ms * 1e-3)
            issue = {"achieved": issue_achieved / 1e9, "peak": issue_peak / 1e9, "unit": "G vector-issue slots/s",
                     "frac": issue_achieved / issue_peak,
                     "frac_at_measured_issue_rate": issue_achieved / (1024 * SCLK_GHZ * 1e9 / 2.25),
                     "how": f"slots per launch = {valu_per_transition:.1f} vector instructions per transition of 64 chains (SQ_INSTS_VALU) x (1 + "
                            f"{double_slot_share:.3f} two-slot share, static from the kernel's disassembly) x {C_PER_GPU // 64} waves x "
                            f"{N_COLLECT + N_DISCARD} transitions / this run's kernel_ms; peak = 1024 SIMDs x {SCLK_GHZ} GHz / 2 cycles per "
                            "slot (MI355X_MICROARCH.md); tools/issue_rate.hip measures 2.25 cycles per slot from two waves per SIMD on "
                            "(frac_at_measured_issue_rate)",
                     "counters": {k: issue_counters[k] for k in ("kernel", "round", "sources_sha256", "source", "vgpr",
                                                                 "valu_instructions_per_transition_of_64_chains",
                                                                 "salu_instructions_per_transition_of_64_chains",
                                                                 "lds_instructions_per_transition_of_64_chains", "double_slot_share",
                                                                 "valu_active_over_wave_cycles", "wait_any_over_wave_cycles",
                                                                 "wait_inst_any_over_wave_cycles") if k in issue_counters}}
        res = {
            "metric": "samples/sec (all chains), 3D Rosenbrock HMC",
            "value": samples / dt_max,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {
                "workload": "BASELINE.json configs[2]: RosenbrockND D=3 HMC, 65536 chains/GPU, eps=0.032, L=10, f32, "
                            "run(n_collect=400, n_discard=50) per step, init_with_seed(C,3,42), seed 42",
                "chains_per_gpu": C_PER_GPU, "dim": DIM, "n_leapfrog": N_LEAPFROG, "n_collect": N_COLLECT,
                "n_discard": N_DISCARD, "parallelism": f"chains sharded x{world}, no data-path collective",
            },
            "leapfrog_steps_per_s": float(args.steps) * C_PER_GPU * world * (N_COLLECT + N_DISCARD) * N_LEAPFROG / dt_max,
            ("ess_per_s" if float((1.0 / rhat).max()) <= 1.05 else "ess_per_s_unconverged"):
                float(ess.min()) * args.steps / (dt_max + stats_s * args.steps),
            "ess_min": float(ess.min()),
            "split_rhat_max_conventional": float((1.0 / rhat).max()),
            "ess_note": "ESS of ONE run(400, 50) from init_with_seed: while the conventional split R-hat is above 1.05 the chains have "
                        "not converged (nor has the reference's example in 450 transitions) and the figure is named "
                        "ess_per_s_unconverged: it is not an efficiency.  The metric's ESS/s on a CONVERGED run of this target under "
                        "this sampler is side.config3_hmc_converged (R-hat <= 1.05, moments within 1 % of the exact ones): one fixed "
                        "(eps, L) does not converge on RosenbrockND(3) at any setting tried except eps <= 0.008, a sequence of handles "
                        "with (eps, L) drawn per launch does",
            "stats_ms": stats_s * 1e3,
            "diagnostics_exchange": (f"torch.distributed backend {backend}: two all-reduces of cross-chain sums "
                                     "(stats.split_rhat_mean_ess_distributed)" if distributed else "single device"),
            "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0) if distributed else None,
            "per_rank": per_rank, "problems": problems,
            "kernel_ms_spread_max_over_min": max(r["kernel_ms"] for r in per_rank) / max(min(r["kernel_ms"] for r in per_rank), 1e-9),
            "pipelined_ms_per_step": pipe_ms,
            "ess_per_s_pipelined_unconverged": (float(ess.min()) / (pipe_ms * 1e-3)) if pipe_ms else None,
            "preroll": {"seconds": args.preroll_seconds, "steps": preroll_steps,
                        "note": "untimed launches before the counted warm-up so that the timed region runs at steady clocks"},
            "roofline": {
                "kernel": kernel_name + " ...>(mm_run_args<float>)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                "traffic_source": traffic_source,
                "traffic_over_algorithmic": (traffic or {}).get("traffic_over_algorithmic"),
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_how": f"C x D x 4 x (n_collect + 2) = {C_PER_GPU} x {DIM} x 4 x ({N_COLLECT} + 2): the sample written once, the "
                                         "state loaded and stored once (SURVEY 8d)",
                "kernel_ms": k_ms,
                "kernel_ms_how": "HIP events on the launch stream around the timed region / steps",
                "kernel_ms_one_launch_at_a_time": float(np.mean(kernel_ms)),
                "launches_per_step": 1,
                "what_binds": "vector-instruction issue, not HBM: see fp32 (arithmetic) and issue (instruction slots); north_star's "
                              ">= 40 % of HBM would need < 125 vector instructions per transition of 64 chains, the ten leapfrog steps "
                              "alone are 120 (DESIGN.md 5.1)",
                "fp32": {"achieved": valu_tflops, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_tflops / FP32_VALU_PEAK_TFLOPS,
                         "flop_per_iteration": FLOP_PER_ITER},
                "issue": issue,
            },
        }
        if world == 1 and not args.no_side:
            res["side"] = side_configs(dev, 0.0 if args.no_cpu_baseline else max(3.0, args.cpu_seconds / 3))
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)  # after the line: a launcher run that shared GPUs or left RCCL cannot pass for a measurement


if __name__ == "__main__":
    main()
