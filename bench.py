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


def cpu_baseline(seconds_target: float = 15.0) -> dict:
    """The reference-ordered CPU restatement (oracle/mh_hmc.c: hmc.rs:304-431 op for op; xoshiro256++ / ziggurat
    noise) timed on this host's cores on a bounded sample of the workload.  Chains are block-partitioned over one
    thread per core for the whole run -- more parallelism than the reference itself extracts (its HMC driver is
    single-threaded around burn tensor ops), so the baseline errs on the CPU's side."""
    import numpy as np

    import oracle as O

    cores = os.cpu_count() or 1
    n_chains = 64 * cores
    init = O.init_with_seed(n_chains, DIM, SEED, np.float32)
    h = O.HMC(O.rosenbrock_nd(DIM), init, STEP_SIZE, N_LEAPFROG, np.float32).seed_blocked(SEED)
    h.run(20, 5, n_threads=cores, want_out=False)  # warm caches / thread pool
    t0 = time.perf_counter()
    reps = 0
    while True:
        h.run(N_COLLECT, N_DISCARD, n_threads=cores, want_out=True)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target:
            break
    samples = reps * n_chains * N_COLLECT
    return {
        "value": samples / dt,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{reps} x HMC::run({N_COLLECT},{N_DISCARD}) of {n_chains} chains (same target, eps, L, f32) in {dt:.1f} s; "
                  "restatement of mini-mcmc's CPU path (the Rust reference cannot be built on this image)",
        "leapfrog_steps_per_s": reps * n_chains * (N_COLLECT + N_DISCARD) * N_LEAPFROG / dt,
    }


def _cpu_side_baselines(seconds_each: float) -> dict:
    """BASELINE.md section 3's other CPU rows, on bounded samples: config 2 = the reference-faithful MH restatement
    (oracle/mh_hmc.c with the reference's own stream: xoshiro256++ / ziggurat, every chain holding a clone of the
    proposal generator (quirk Q1), D + 1 normals per proposal (Q2), logp recomputed, q-terms kept --
    metropolis_hastings.rs:150-153, 303-315; distributions.rs:364-372), chains block-partitioned over all cores;
    config 5 = the recursive NUTS restatement (oracle/nuts.c, nuts.rs:550-946) on the same 32-D target, scaled down to
    4 chains per core and 30 + 10 transitions."""
    import numpy as np

    import oracle as O

    cores = os.cpu_count() or 1
    out = {}
    n_chains = 64 * cores
    init = O.init_with_seed(n_chains, 2, SEED, np.float32)
    mh = O.MetropolisHastings(O.gaussian2d([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), 1.0, init, np.float32, proposal_seed=SEED).seed(SEED)
    mh.run(50, 10, n_threads=cores, want_out=False)
    t0, reps = time.perf_counter(), 0
    while True:
        mh.run(1000, 100, n_threads=cores, want_out=True)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_each:
            break
    out["config2_mh"] = {"value": reps * n_chains * 1000 / dt, "unit": "samples/s", "cores": cores, "kind": "port",
                         "sample": f"{reps} x run(1000,100) of {n_chains} chains in {dt:.1f} s, reference-faithful MH (Q1, Q2 kept)"}
    from mini_mcmc_amd.distributions import GaussianND

    g = GaussianND.ill_conditioned(32, 1e4, 7)
    n5 = 4 * cores
    nuts = O.NUTS(O.gaussian_nd(g.precision), O.init_with_seed(n5, 32, SEED) * 0.1, 0.8, mode=2).set_seed(SEED).set_max_depth(10)
    t0 = time.perf_counter()
    nuts.run(10, 30, progress=True, n_threads=cores)
    dt = time.perf_counter() - t0
    lf = sum(nuts.chain_state(i)["n_leapfrog_total"] for i in range(n5))
    out["config5_nuts"] = {"value": lf / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
                           "draws_per_s": n5 * 10 / dt,
                           "sample": f"{n5} chains x (30 warm-up + 10 draws) in {dt:.1f} s, recursive NUTS restatement, f64, max depth 10"}
    return out


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
        n_b2b = 100
        t0 = time.perf_counter()
        for _ in range(n_b2b):
            mh_step()
        torch.cuda.synchronize()
        k = (time.perf_counter() - t0) / n_b2b * 1e3
        mh.enable_timing(True)
        ms = []
        for _ in range(10):
            mh_step()
            ms.append(mh.timing()["kernel_ms"])
        ke = float(np.median(ms))
        out["config2_mh"] = {"kernel_ms": ke, "kernel_ms_how": "HIP events around one launch, median of 10 (as roofline.kernel_ms of the headline)",
                             "ms_per_run_back_to_back": k,
                             "samples_per_s": C_PER_GPU * 1000 / (k * 1e-3),
                             "hbm_frac": C_PER_GPU * 2 * 4 * (1000 + 2) / (ke * 1e-3) / 1e9 / HBM_PEAK_GBS}
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
                               "ess_per_s": float(ess.min()) / ((k + st_ms) * 1e-3),
                               "samples_per_s": C_PER_GPU * 1000 / (k * 1e-3)}
        del h, t
        g = GaussianND.ill_conditioned(32, 1e4, 7)
        nuts = NUTS(g, init_with_seed(C_PER_GPU, 32, SEED) * 0.1, 0.8, mode=2, device=dev.index or 0).set_seed(SEED).set_max_depth(10)
        nuts._run(100, 200, True, "torch")
        torch.cuda.synchronize()
        tm = nuts.timing()
        k = float(tm["kernel_ms"])
        lf = float(nuts.leapfrog_counts().sum())
        out["config5_nuts"] = {"kernel_ms": k, "n_launches": int(tm["n_launches"]), "leapfrog_steps_per_s": lf / (k * 1e-3),
                               "draws_per_s": C_PER_GPU * 100 / (k * 1e-3), "kernel_variant": nuts.kernel_variant,
                               "f64_mfma_frac": lf * 2 * 32 * 32 / (k * 1e-3) / 78.6e12}
        del nuts
        if cpu_seconds > 0:
            out["cpu_baselines"] = _cpu_side_baselines(cpu_seconds)
    except Exception as e:  # pragma: no cover -- side numbers must not take the headline down
        out["error"] = repr(e)
    return out


def main_group(args) -> None:
    """--group: the same headline through the in-library device group -- one process, N devices (MMCMC_BENCH_GROUP_DEVICES =
    comma-separated device list overrides 0..N-1, e.g. "0,0" rehearses two shards on one GPU through the host exchange).
    A step = mmcmc_hmc_group_run of all N x 65 536 chains (every shard's kernel on its own device and stream, the call
    returns when all have finished); the diagnostics afterwards go through ncclAllGather / ncclAllReduce inside the
    library, and the line says which path they took."""
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

    def step():
        g.run(N_COLLECT, N_DISCARD, to_host=False)

    t_pre, pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(20):
            step()
        pre += 20
    for _ in range(args.warmup):
        step()
    for d in set(devices):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for d in set(devices):
        torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    g.split_rhat_mean_ess()
    ts = time.perf_counter()
    for _ in range(5):
        rhat, ess = g.split_rhat_mean_ess()
    stats_s = (time.perf_counter() - ts) / 5
    samples = float(args.steps) * C_PER_GPU * n * N_COLLECT
    ms = dt / args.steps * 1e3
    alg_bytes = C_PER_GPU * DIM * 4 * (N_COLLECT + 2)
    res = {
        "metric": "samples/sec (all chains), 3D Rosenbrock HMC", "value": samples / dt, "unit": "samples/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2] per device (configs[3] at 8): RosenbrockND D=3 HMC, 65536 chains/GPU, "
                               "eps=0.032, L=10, f32, run(400, 50) per step", "chains_per_gpu": C_PER_GPU, "dim": DIM,
                   "n_leapfrog": N_LEAPFROG, "n_collect": N_COLLECT, "n_discard": N_DISCARD, "devices": devices,
                   "parallelism": f"in-library device group x{n}: one process, one host thread + stream per device, no data-path collective"},
        "leapfrog_steps_per_s": float(args.steps) * C_PER_GPU * n * (N_COLLECT + N_DISCARD) * N_LEAPFROG / dt,
        "ess_min": float(ess.min()), "ess_per_s": float(ess.min()) * args.steps / (dt + stats_s * args.steps),
        "stats_ms": stats_s * 1e3,
        "diagnostics_exchange": {1: "rccl (ncclAllGather + ncclAllReduce inside libmmcmc.so)", 0: "host (a device is listed twice)",
                                 -1: "host FALLBACK: no RCCL library could be loaded", -2: "host FALLBACK: ncclCommInitAll failed"}[g.exchange_status],
        "preroll": {"seconds": args.preroll_seconds, "steps": pre},
        # a group step includes the host's fan-out to N device threads and their join: per-device kernel time is not separated
        "roofline": {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G vector-issue slots/s", "frac": None, "traffic": None,
                     "hbm": {"achieved": alg_bytes * n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * n, "unit": "GB/s",
                             "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "from ms_per_step (host fan-out included)"}},
    }
    print(json.dumps(res))


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

    if args.group:
        return main_group(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    # MMCMC_BENCH_DRYRUN_ONE_DEVICE=1: rehearsal of the N > 1 path on a box with ONE GPU (every rank on device 0, gloo for
    # the barrier and the diagnostics' reductions, since RCCL refuses one device twice); the line it prints says so and is
    # not a measurement
    dryrun = distributed and os.environ.get("MMCMC_BENCH_DRYRUN_ONE_DEVICE") == "1"
    if dryrun:
        local_rank = 0
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

    if rank == 0:
        samples = float(args.steps) * C_PER_GPU * world * N_COLLECT
        ms_per_step = dt_max / args.steps * 1e3
        k_ms = float(kernel_ms_timed_region)
        alg_bytes = C_PER_GPU * DIM * 4 * (N_COLLECT + 2)  # sample written once + state loaded and stored once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        iters = C_PER_GPU * (N_COLLECT + N_DISCARD)
        valu_tflops = iters * FLOP_PER_ITER / (k_ms * 1e-3) / 1e12
        # What binds the kernel is vector-instruction issue, not HBM.  The instruction counts per
        # transition come from SQ-counter passes over this very kernel (tools/pmc_sq.sh -> profiles/hmc_kernel_counters.json:
        # counters cannot be read inside an un-profiled run); the launch duration they are divided by is THIS run's.
        # HBM traffic likewise is a profile of the kernel, not of this run: `traffic` stays null here and the profiled
        # figure is reported under its own key with its source.
        traffic_profile, issue = None, None
        tpath = os.path.join(ROOT, "profiles", "hmc_kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("variant", 2) == args.variant:
                    traffic_profile = {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                                       "source": "profiles/hmc_kernel_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                 "passes over this kernel (tools/pmc_hmc_traffic.sh), NOT this run"}
            except Exception:
                traffic_profile = None
        cpath = os.path.join(ROOT, "profiles", "hmc_kernel_counters.json")
        if os.path.exists(cpath):
            try:
                cj = json.load(open(cpath))
                if cj.get("variant", 2) == args.variant:
                    issue = cj
            except Exception:
                issue = None
        # vector-issue roofline of the kernel's own instruction mix: a SIMD issues one single-slot vector instruction per
        # ~2.25 cycles once several waves feed it, packed / 32x32-multiply instructions take two slots (tools/issue_rate.hip,
        # profiles/r2b_issue_rate_full.log); slots per transition of 64 chains from the SQ counters
        valu_per_transition = float((issue or {}).get("valu_instructions_per_transition_of_64_chains", 249.4))
        double_slot_share = float((issue or {}).get("double_slot_share", 0.45))
        slots_per_launch = valu_per_transition * (1.0 + double_slot_share) * (C_PER_GPU / 64) * (N_COLLECT + N_DISCARD)
        n_simd = 1024
        issue_peak = n_simd * SCLK_GHZ * 1e9 / 2.25  # issue slots per second, all SIMDs
        issue_achieved = slots_per_launch / (k_ms * 1e-3)
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
            "data": "synthetic" if not dryrun else "synthetic; DRY RUN: all ranks on one device over gloo, not a measurement",
            "config": {
                "workload": "BASELINE.json configs[2]: RosenbrockND D=3 HMC, 65536 chains/GPU, eps=0.032, L=10, f32, "
                            "run(n_collect=400, n_discard=50) per step, init_with_seed(C,3,42), seed 42",
                "chains_per_gpu": C_PER_GPU, "dim": DIM, "n_leapfrog": N_LEAPFROG, "n_collect": N_COLLECT,
                "n_discard": N_DISCARD, "parallelism": f"chains sharded x{world}, no data-path collective",
            },
            "leapfrog_steps_per_s": float(args.steps) * C_PER_GPU * world * (N_COLLECT + N_DISCARD) * N_LEAPFROG / dt_max,
            "ess_per_s": float(ess.min()) * args.steps / (dt_max + stats_s * args.steps),
            "ess_min": float(ess.min()),
            "split_rhat_max_conventional": float((1.0 / rhat).max()),
            "ess_note": "ESS of ONE run(400, 50) from init_with_seed: split R-hat ~1.9, the Rosenbrock chains have not "
                        "converged in 450 transitions (nor has the reference's example); side.config3_long is run(1000, 200)",
            "stats_ms": stats_s * 1e3,
            "pipelined_ms_per_step": pipe_ms,
            "ess_per_s_pipelined": (float(ess.min()) / (pipe_ms * 1e-3)) if pipe_ms else None,
            "preroll": {"seconds": args.preroll_seconds, "steps": preroll_steps,
                        "note": "untimed launches before the counted warm-up so that the timed region runs at steady clocks"},
            "roofline": {
                "kernel": {0: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=0, L=0>",
                           1: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=2, L=10>",
                           2: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=2, L=10>",
                           5: "mm_run_split_kernel<float, RosenbrockND<3>, HMC, L=10>"}[args.variant],
                # what binds this kernel: vector-instruction issue (12 algorithmic bytes against ~250 vector instructions per
                # transition of 64 chains); the HBM roofline BASELINE.json's north_star asks about is the `hbm` entry below
                "bound": "valu_issue",
                "achieved": issue_achieved / 1e9,
                "peak": issue_peak / 1e9,
                "unit": "G vector-issue slots/s",
                "frac": issue_achieved / issue_peak,
                "traffic": None,
                "how": f"slots per launch = {valu_per_transition:.1f} vector instructions per transition of 64 chains x (1 + "
                       f"{double_slot_share:.2f} two-slot share) x {C_PER_GPU // 64} waves x {N_COLLECT + N_DISCARD} transitions "
                       "(SQ counters, profiles/hmc_kernel_counters.json) / this run's kernel_ms; peak = 1024 SIMDs x "
                       f"{SCLK_GHZ} GHz / 2.25 cycles per slot (tools/issue_rate.hip)",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes, "traffic": None, "traffic_profile": traffic_profile},
                "valu_issue_counters": issue,
                "kernel_ms": k_ms,
                "kernel_ms_how": "HIP events on the launch stream around the timed region / steps",
                "kernel_ms_one_launch_at_a_time": float(np.mean(kernel_ms)),
                "launches_per_step": 1,
                "valu": {"achieved_tflops": valu_tflops, "peak_tflops": FP32_VALU_PEAK_TFLOPS,
                         "frac": valu_tflops / FP32_VALU_PEAK_TFLOPS, "flop_per_iteration": FLOP_PER_ITER},
            },
        }
        if world == 1 and not args.no_side:
            res["side"] = side_configs(dev, 0.0 if args.no_cpu_baseline else max(3.0, args.cpu_seconds / 3))
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
