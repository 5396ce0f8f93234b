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


def side_configs(dev) -> dict:
    """BASELINE.json's other single-GPU configurations, measured once each after the headline run and reported as
    extra keys (they are parity-test cases, not bench lines): config 2 (Gaussian2D MH, 65 536 chains, f32,
    run(1000, 100)) and config 5 (NUTS, 32-D Gaussian of condition number 1e4, 65 536 chains, f64, 200 + 100,
    tree-depth compaction).  Kernel time by HIP events on the launch stream.  Never fails the bench."""
    out = {}
    try:
        import numpy as np
        import torch

        from mini_mcmc_amd.core import init_with_seed
        from mini_mcmc_amd.distributions import Gaussian2D, GaussianND, IsotropicGaussian
        from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
        from mini_mcmc_amd.nuts import NUTS

        mh = MetropolisHastings(Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), IsotropicGaussian(1.0),
                                init_with_seed(C_PER_GPU, 2, SEED, np.float32), device=dev.index or 0).seed(SEED)
        ms = []
        for _ in range(4):
            mh.run(1000, 100, to="torch", accept_counts=False)
            ms.append(mh.timing()["kernel_ms"])
        k = float(np.median(ms[1:]))
        out["config2_mh"] = {"kernel_ms": k, "samples_per_s": C_PER_GPU * 1000 / (k * 1e-3),
                             "hbm_frac": C_PER_GPU * 2 * 4 * (1000 + 2) / (k * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del mh
        g = GaussianND.ill_conditioned(32, 1e4, 7)
        nuts = NUTS(g, init_with_seed(C_PER_GPU, 32, SEED) * 0.1, 0.8, mode=2, device=dev.index or 0).set_seed(SEED).set_max_depth(10)
        nuts._run(100, 200, True, "torch")
        torch.cuda.synchronize()
        k = float(nuts.timing()["kernel_ms"])
        lf = float(nuts.leapfrog_counts().sum())
        out["config5_nuts"] = {"kernel_ms": k, "leapfrog_steps_per_s": lf / (k * 1e-3), "draws_per_s": C_PER_GPU * 100 / (k * 1e-3),
                               "kernel_variant": nuts.kernel_variant,
                               "f64_mfma_frac": lf * 2 * 32 * 32 / (k * 1e-3) / 78.6e12}
    except Exception as e:  # pragma: no cover -- side numbers must not take the headline down
        out["error"] = repr(e)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements of configs 2 and 5")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--variant", type=int, default=5, help="kernel variant: 5 noise waves + transition waves, two waves per SIMD (default), 2 one wave per SIMD with paired + pipelined noise, 0 plain")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    sampler.enable_timing(True)
    # per-launch device time of the sampling kernel, HIP events on the launch stream (separate, un-timed pass so the
    # event queries do not perturb the timed region)
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

    if rank == 0:
        samples = float(args.steps) * C_PER_GPU * world * N_COLLECT
        ms_per_step = dt_max / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = C_PER_GPU * DIM * 4 * (N_COLLECT + 2)  # sample written once + state loaded and stored once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        iters = C_PER_GPU * (N_COLLECT + N_DISCARD)
        valu_tflops = iters * FLOP_PER_ITER / (k_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hmc_kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
            "data": "synthetic",
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
            "stats_ms": stats_s * 1e3,
            "roofline": {
                "kernel": {0: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=0, L=0>",
                           1: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=2, L=10>",
                           2: "mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=2, L=10>",
                           5: "mm_run_split_kernel<float, RosenbrockND<3>, HMC, L=10>"}[args.variant],
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": k_ms,
                "launches_per_step": 1,
                "note": "the HMC kernel is f32-VALU/issue bound, not HBM bound (about 270 VALU instructions per "
                        "12 algorithmic bytes); the HBM fraction is reported because BASELINE.json asks for it",
                "valu": {"achieved_tflops": valu_tflops, "peak_tflops": FP32_VALU_PEAK_TFLOPS,
                         "frac": valu_tflops / FP32_VALU_PEAK_TFLOPS, "flop_per_iteration": FLOP_PER_ITER},
            },
        }
        if world == 1 and not args.no_side:
            res["side"] = side_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
