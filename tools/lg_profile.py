#!/usr/bin/env python3
"""Where the lane-group NUTS kernel's cycles go (config-5 target).  Warm up with the product sampler, then run the
profiling build of the same kernel (tools/lg_profile.hip) from the adapted state and print per-section shares."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=16384)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--lib", default="liblgprof.so", help="liblgprof.so (every section stamped), liblgprof_coarse.so (-DMM_LG_PROFILE_MASK=0x3fu), ...")
    args = ap.parse_args()
    import torch

    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import GaussianND
    from mini_mcmc_amd.nuts import NUTS

    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), args.lib))
    lib.lgprof_scratch_doubles_per_wave.restype = C.c_ulonglong
    lib.lgprof_rec_doubles_per_chain.restype = C.c_ulonglong
    g = GaussianND.ill_conditioned(32, 1e4, 7)
    n = args.chains
    s = NUTS(g, init_with_seed(n, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_max_depth(10).set_kernel_variant(1)
    s._run(0, args.warmup, True, "numpy")
    pos = s.positions()
    ad = s.adapt_state()
    adapt = np.stack([ad["epsilon"], ad["epsilon_bar"], ad["h_bar"], ad["mu"]], axis=1).astype(np.float64)
    dev = torch.device("cuda", 0)
    waves = (n + 15) // 16
    t_mat = torch.tensor(np.asarray(g.precision, dtype=np.float64), device=dev)
    t_state = torch.tensor(pos, device=dev)
    t_adapt = torch.tensor(adapt, device=dev)
    t_scr = torch.empty(waves * lib.lgprof_scratch_doubles_per_wave(), dtype=torch.float64, device=dev)
    t_rec = torch.zeros(waves * 16 * lib.lgprof_rec_doubles_per_chain(), dtype=torch.float64, device=dev)
    t_prof = torch.zeros((waves, 16), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.lgprof_run(C.c_void_p(t_mat.data_ptr()), C.c_void_p(t_state.data_ptr()), C.c_void_p(t_adapt.data_ptr()),
                        C.c_ulonglong(n), C.c_ulonglong(42), C.c_uint(args.warmup), C.c_uint(args.steps),
                        C.c_uint(args.warmup), C.c_int(10), C.c_void_p(t_scr.data_ptr()), C.c_void_p(t_rec.data_ptr()),
                        C.c_void_p(t_prof.data_ptr()))
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    p = t_prof.cpu().numpy().astype(np.float64)
    names = ["transition prologue", "doubling prologue", "leapfrog", "walk", "doubling epilogue", "transition epilogue"]
    tot = p[:, :6].sum(axis=1) + p[:, 8:12].sum(axis=1)
    print(f"lib {args.lib}  raw ticks per leaf iteration by section: " + " ".join(f"[{i}] {p[:, i].sum() / max(p[:, 6].sum(), 1):.0f}" for i in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11)))
    print(f"chains {n}  steps {args.steps}  wall {dt * 1e3:.1f} ms  s_memtime ticks per wave: mean {tot.mean():.3e} max {tot.max():.3e}")
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {100 * p[:, i].sum() / tot.sum():6.2f} %")
    leaves, levels = p[:, 6], p[:, 7]
    L_ = leaves.sum()
    print(f"  per leaf iteration: records + kick {p[:, 8].sum() / L_:.0f}  MFMA + logp {p[:, 9].sum() / L_:.0f}  leaf bookkeeping {p[:, 2].sum() / L_:.0f}"
          f"  merges {p[:, 10].sum() / L_:.0f} ({p[:, 12].sum() / L_:.2f} per leaf, {p[:, 10].sum() / max(p[:, 12].sum(), 1):.0f} each)"
          f"  pushes {p[:, 11].sum() / L_:.0f} ({p[:, 13].sum() / L_:.2f} per leaf, {p[:, 11].sum() / max(p[:, 13].sum(), 1):.0f} each)"
          f"  walk glue {p[:, 3].sum() / L_:.0f}")
    print(f"  leaf iterations per wave per transition: {leaves.mean() / args.steps:.1f};  walk levels per leaf: {levels.sum() / leaves.sum():.2f}")
    print(f"  ticks per leaf iteration: leapfrog {p[:, 2].sum() / leaves.sum():.1f}  walk {p[:, 3].sum() / leaves.sum():.1f}"
          f"  (s_memtime ticks; the counter ran at ~2.1 GHz in these runs)")


if __name__ == "__main__":
    main()
