"""Differential fuzz on the GPU: kernels that must give the same bits are run on random shapes and compared.
  - integer-state MH: the four-waves-per-SIMD kernel against the one-wave kernel (random chain counts, run lengths, starts
    inside and outside the table, continued runs);
  - Gibbs mixture: likewise;
  - tracker: whole blocks (one wave per parameter, tiles of 16 rows) against step-by-step feeding (the plain kernel);
  - NUTS: asynchronous-lane pair kernel against the lanes in step, compiled and run-time compiled dimensions.
usage: python tools/fuzz_variants.py [seconds per family, default 40] [families: any of d g t n, default dgtn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mini_mcmc_amd import discrete as D, stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import IsotropicGaussian, RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
fam = sys.argv[2] if len(sys.argv) > 2 else "dgtn"
rng = np.random.default_rng(int(time.time()) & 0xffff)
print("seed", rng.bit_generator.state["state"]["state"] & 0xffff)

t0, n = time.time(), 0
while "d" in fam and time.time() - t0 < budget:
    model = [D.PoissonReflect(float(rng.uniform(0.5, 9))), D.BinomialClamp(int(rng.integers(1, 40)), float(rng.uniform(0.05, 0.95))),
             D.PoissonNonneg(float(rng.uniform(0.5, 9)))][int(rng.integers(3))]
    C = int(rng.integers(4096, 9000))
    start = rng.integers(0, 12, size=C).astype(np.int32)
    if rng.random() < 0.3:
        start[rng.integers(0, C, size=5)] = rng.integers(1000, 3000, size=5)
    seed = int(rng.integers(1 << 30))
    a = D.DiscreteMetropolisHastings(model, start).seed(seed)
    b = D.DiscreteMetropolisHastings(model, start).seed(seed).set_kernel_variant(0)
    for _ in range(2):
        nc, nd = int(rng.integers(0, 130)), int(rng.integers(0, 60))
        oa, ob = a.run(nc, nd), b.run(nc, nd)
        assert np.array_equal(oa, ob) and np.array_equal(a.state(), b.state()) and np.array_equal(a.accept_counts(), b.accept_counts()), (type(model).__name__, C, nc, nd, seed)
    n += 1
print(f"integer-state MH: {n} random cases, split kernel == one-wave kernel")

from mini_mcmc_amd.core import init_det
from mini_mcmc_amd.gibbs import GibbsSampler, MixtureConditional
t0, n = time.time(), 0
while "g" in fam and time.time() - t0 < budget / 2:
    C = int(rng.integers(4096, 9000))
    cond = MixtureConditional(float(rng.uniform(-3, 0)), float(rng.uniform(0.5, 2)), float(rng.uniform(0, 4)), float(rng.uniform(0.5, 2)), float(rng.uniform(0.05, 0.95)))
    seed = int(rng.integers(1 << 30))
    a = GibbsSampler(cond, init_det(C, 2)).set_seed(seed)
    b = GibbsSampler(cond, init_det(C, 2)).set_seed(seed).set_kernel_variant(0)
    for _ in range(2):
        nc, nd = int(rng.integers(0, 80)), int(rng.integers(0, 40))
        assert np.array_equal(a.run(nc, nd), b.run(nc, nd)), (C, nc, nd, seed)
    n += 1
print(f"Gibbs mixture: {n} random cases, split kernel == one-wave kernel")

t0, n = time.time(), 0
while "t" in fam and time.time() - t0 < budget:
    chains, steps, params = int(rng.integers(2, 700)), int(rng.integers(2, 90)), int(rng.integers(1, 9))
    x = rng.standard_normal((chains, steps, params)).astype(np.float32)
    keep = rng.random((chains, steps)) < rng.uniform(0, 0.9)
    for t in range(1, steps):
        x[:, t][keep[:, t]] = x[:, t - 1][keep[:, t]]
    if rng.random() < 0.3:
        x = x.astype(np.float64)
    dev = torch.as_tensor(x, device="cuda")
    ta = S.MultiChainTracker(chains, params).step(dev)
    tb = S.MultiChainTracker(chains, params)
    t = 0
    while t < steps:
        k = min(int(rng.integers(1, 16)), steps - t)
        tb.step(dev, t0=t, k=k)
        t += k
    assert np.array_equal(ta.rhat(), tb.rhat()) and ta.p_accept == tb.p_accept, (chains, steps, params, x.dtype)
    ra, mxa, pa = ta.chain_stats()
    rb, mxb, pb = tb.chain_stats()
    assert np.array_equal(ra, rb) and mxa == mxb and pa == pb, (chains, steps, params)
    n += 1
print(f"tracker: {n} random cases, whole blocks == step by step")

t0, n = time.time(), 0
while "n" in fam and time.time() - t0 < budget:
    d = int(rng.integers(1, 33))
    tgt = [StandardNormal(d), IsotropicGaussian(float(rng.uniform(0.5, 2)), d), RosenbrockND(max(d, 2))][int(rng.integers(3))]
    mode = int(rng.integers(3))
    C = int(rng.integers(1, 400))
    init = init_with_seed(C, tgt.dim, int(rng.integers(1000))) * 0.5
    seed = int(rng.integers(1 << 30))
    a = NUTS(tgt, init, 0.8, mode=mode).set_seed(seed)
    b = NUTS(tgt, init, 0.8, mode=mode).set_seed(seed)
    try:
        b.set_kernel_variant(0)
    except Exception:
        b.set_kernel_variant(6)  # a run-time compiled unit: its reference is the run-time-dimension kernel
    prog = bool(rng.integers(2))
    nc, nd = int(rng.integers(1, 12)), int(rng.integers(0, 12))
    assert np.array_equal(a._run(nc, nd, prog, "numpy"), b._run(nc, nd, prog, "numpy")), (type(tgt).__name__, tgt.dim, mode, C, nc, nd, seed, a.kernel_variant)
    assert np.array_equal(a.leapfrog_counts(), b.leapfrog_counts()) and np.array_equal(a.depth_histogram(), b.depth_histogram())
    sa, sb = a.adapt_state(), b.adapt_state()
    assert all(np.array_equal(sa[k], sb[k]) for k in sa), (type(tgt).__name__, tgt.dim, mode)
    n += 1
print(f"NUTS: {n} random cases, default kernel == lanes in step / run-time-dimension kernel")
