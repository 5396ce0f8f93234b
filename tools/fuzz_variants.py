"""Differential fuzz on the GPU: kernels that must give the same bits are run on random shapes and compared.
  - integer-state MH: the four-waves-per-SIMD kernel against the one-wave kernel (random chain counts, run lengths, starts
    inside and outside the table, continued runs);
  - Gibbs mixture: likewise;
  - tracker: whole blocks (one wave per parameter, tiles of 16 rows) against step-by-step feeding (the plain kernel);
  - NUTS: asynchronous-lane pair kernel against the lanes in step, compiled and run-time compiled dimensions;
  - lane-group NUTS (D = 16, 32, f64): the persistent scheduler against the single launch;
  - MH / HMC at the compiled dimensions up to 8: the four-waves-per-SIMD kernel (5) against the one-wave kernels (0, 2), any
    chain count, run length, iterations per launch, f32 and f64;
  - diagnostics: the power-spectrum kernel against the direct sums (R-hat / ESS to 1e-4 / 2e-3); x: long half-chains
    (N1 residues, any N1) against the direct sums.
usage: python tools/fuzz_variants.py [seconds per family, default 40] [families: any of d g t n l h s x m, default dgtnlhsm]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mini_mcmc_amd import discrete as D, stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import IsotropicGaussian, RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
fam = sys.argv[2] if len(sys.argv) > 2 else "dgtnlhsm"
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
    # NaN == NaN here: two chains that never moved in a parameter have no variance at all, and both feeding orders say so
    same = lambda u, v: np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True)
    assert same(ta.rhat(), tb.rhat()) and same(ta.p_accept, tb.p_accept), (chains, steps, params, x.dtype, ta.rhat(), tb.rhat())
    ra, mxa, pa = ta.chain_stats()
    rb, mxb, pb = tb.chain_stats()
    assert same(ra, rb) and same(mxa, mxb) and same(pa, pb), (chains, steps, params)
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

from mini_mcmc_amd.distributions import GaussianND
t0, n = time.time(), 0
while "l" in fam and time.time() - t0 < budget:
    # lane-group NUTS (dense Gaussian, f64, matrix cores): the persistent scheduler (3: queues, units, chains handed from wave to
    # wave, edges in registers) against the single launch in which every wave keeps its chains (1) -- random chain counts, depth
    # caps, first compaction levels, step-size targets, and a continued run
    d = int(rng.choice([16, 32]))
    tgt = GaussianND.ill_conditioned(d, float(10 ** rng.uniform(0, 4)), int(rng.integers(100)))
    C = int(rng.integers(2048, 5000))
    init = init_with_seed(C, d, int(rng.integers(1000))) * float(rng.uniform(0.05, 1.0))
    seed, p_acc, md = int(rng.integers(1 << 30)), float(rng.uniform(0.6, 0.95)), int(rng.integers(3, 9))
    a = NUTS(tgt, init, p_acc, mode=2).set_seed(seed).set_max_depth(md).set_kernel_variant(3)
    if rng.random() < 0.5:
        a.set_compaction(int(rng.integers(1, md + 1)))
    b = NUTS(tgt, init, p_acc, mode=2).set_seed(seed).set_max_depth(md).set_kernel_variant(1)
    for _ in range(2):
        nc, nd = int(rng.integers(1, 40)), int(rng.integers(0, 40))
        oa, ob = a._run(nc, nd, False, "numpy"), b._run(nc, nd, False, "numpy")
        assert np.array_equal(oa, ob), ("lane groups", d, C, seed, p_acc, md, nc, nd)
        assert np.array_equal(a.leapfrog_counts(), b.leapfrog_counts()) and np.array_equal(a.depth_histogram(), b.depth_histogram())
        sa, sb = a.adapt_state(), b.adapt_state()
        assert all(np.array_equal(sa[k], sb[k]) for k in sa), ("lane groups: adaptation state", d, C, seed)
    n += 1
if "l" in fam:
    print(f"lane-group NUTS: {n} random cases, persistent scheduler == single launch")

from mini_mcmc_amd.distributions import Gaussian2D, Rosenbrock2D
from mini_mcmc_amd.hmc import HMC
from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
t0, n = time.time(), 0
while "h" in fam and time.time() - t0 < budget:
    d = int(rng.integers(1, 9))
    dt = np.float32 if rng.random() < 0.6 else np.float64
    kind = int(rng.integers(4))
    tgt = [StandardNormal(d), IsotropicGaussian(float(rng.uniform(0.5, 2)), d), RosenbrockND(max(d, 2)), Gaussian2D([0.3, -0.2], [[1.0, 0.4], [0.4, 2.0]])][kind]
    C = int(rng.integers(1, 1500))
    init = (init_with_seed(C, tgt.dim, int(rng.integers(1000))) * 0.5).astype(dt)
    seed = int(rng.integers(1 << 30))
    hmc = bool(rng.integers(2))
    def make():
        if hmc:
            return HMC(tgt, init, float(rng_eps), int(rng_L)).set_seed(seed)
        return MetropolisHastings(tgt, IsotropicGaussian(float(rng_eps) * 4, tgt.dim), init).seed(seed)
    rng_eps, rng_L = rng.uniform(0.01, 0.2), rng.integers(1, 13)
    a, outs = make(), []
    nc, nd, ipl = int(rng.integers(0, 140)), int(rng.integers(0, 60)), int(rng.choice([0, 0, 7, 50]))
    for v in (None, 0, 2):
        s_ = make()
        try:
            if v is not None:
                s_.set_kernel_variant(v)
        except Exception:
            continue
        if ipl:
            s_.set_iters_per_launch(ipl)
        o1 = s_.run(nc, nd)
        o2 = s_.run(int(nc // 3), 3)  # continued
        outs.append((o1, o2, s_.state()))
    for o in outs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(o, outs[0])), (type(tgt).__name__, tgt.dim, dt.__name__, C, nc, nd, ipl, seed, hmc)
    n += 1
print(f"MH / HMC: {n} random cases, default kernel == one-wave kernels")

t0, n = time.time(), 0
while "s" in fam and time.time() - t0 < budget:
    c, m2, p = int(rng.integers(2, 400)), int(rng.integers(202, 2049)), int(rng.integers(1, 7))
    x = np.zeros((c, m2, p), dtype=np.float32)
    e = rng.standard_normal((c, m2, p)).astype(np.float32)
    phi = float(rng.uniform(0, 0.95))
    for t in range(1, m2):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    x += rng.uniform(-50, 50, size=(1, 1, p)).astype(np.float32)
    tt = torch.from_numpy(x).cuda()
    res = {}
    try:
        for k in ("fft", "direct"):
            S.set_kernel(k)
            res[k] = S.split_rhat_mean_ess(tt)
    finally:
        S.set_kernel("auto")
    assert np.allclose(res["fft"][0], res["direct"][0], rtol=1e-4) and np.allclose(res["fft"][1], res["direct"][1], rtol=2e-3), (c, m2, p, phi)
    n += 1
print(f"diagnostics: {n} random cases, power spectrum == direct sums (1e-4 / 2e-3)")

t0, n = time.time(), 0
while "x" in fam and time.time() - t0 < budget:
    # long half-chains: N1 = ceil(m / 1024) residues of 2048-point transforms, ANY N1 (round 6), against the direct sums
    c, m2, p = int(rng.integers(1, 24)), int(rng.integers(2050, 60000)), int(rng.integers(1, 5))
    if rng.random() < 0.3:
        m2 = 2048 * int(rng.integers(1, 24)) + int(rng.integers(-2, 3))  # at the edges of a residue count
    x = np.zeros((c, m2, p), dtype=np.float32)
    e = rng.standard_normal((c, m2, p)).astype(np.float32)
    phi = float(rng.uniform(0, 0.95))
    for t in range(1, m2):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    x += rng.uniform(-50, 50, size=(1, 1, p)).astype(np.float32)
    tt = torch.from_numpy(x).cuda()
    res = {}
    try:
        for k in ("auto", "direct"):
            S.set_kernel(k)
            res[k] = S.split_rhat_mean_ess(tt)
    finally:
        S.set_kernel("auto")
    assert np.allclose(res["auto"][0], res["direct"][0], rtol=1e-4) and np.allclose(res["auto"][1], res["direct"][1], rtol=5e-3), (c, m2, p, phi, res)
    n += 1
if "x" in fam:
    print(f"long diagnostics: {n} random cases, power spectrum (N1 residues) == direct sums (1e-4 / 5e-3)")

t0, n = time.time(), 0
while "m" in fam and time.time() - t0 < budget:
    d = int(rng.integers(1, 9))
    tgt = [StandardNormal(d), IsotropicGaussian(float(rng.uniform(0.5, 2)), d), RosenbrockND(max(d, 2))][int(rng.integers(3))]
    mode, C = int(rng.integers(3)), int(rng.integers(64, 6000))
    init = init_with_seed(C, tgt.dim, int(rng.integers(1000))) * 0.5
    seed, depth = int(rng.integers(1 << 30)), int(rng.integers(3, 11))
    prog, nc, nd = bool(rng.integers(2)), int(rng.integers(1, 10)), int(rng.integers(0, 10))
    outs = []
    for v in (5, 4, 0):
        s_ = NUTS(tgt, init, 0.8, mode=mode).set_seed(seed).set_max_depth(depth).set_kernel_variant(v)
        outs.append((s_._run(nc, nd, prog, "numpy"), s_.leapfrog_counts(), s_.depth_histogram()))
    for o in outs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(o, outs[0])), (type(tgt).__name__, tgt.dim, mode, C, depth, nc, nd, seed)
    n += 1
print(f"NUTS mappings 5 / 4 / 0 at the compiled dimensions, many chains, depth caps: {n} random cases, all equal")
