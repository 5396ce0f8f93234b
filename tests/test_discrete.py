"""Metropolis-Hastings on integer states: the reference's discrete targets / proposals
(tests/metrohast_poisson_test.rs, examples/poisson_mh.rs).

CPU: the oracle restatement (oracle/discrete.c) on the reference's own stream passes the reference's own assertions
(histogram within 0.05 of the pmf after run(20000, 2000) with seed 42), and on the engine's stream equals the host build
of the product's mm_discrete.h exactly -- states, samples and accept counts (integer work: bit-exact).
GPU: the kernel equals that host build exactly, at small and at full size, and reproduces the pmf."""
import math

import numpy as np
import pytest


def poisson_pmf(lam, k):
    return math.exp(-lam) * lam**k / math.factorial(k)


def binomial_pmf(n, p, k):
    return math.comb(n, k) * p**k * (1 - p) ** (n - k)


CASES = [
    # kind name, params, init, pmf
    ("POISSON_REFLECT", [4.0], 0, lambda k: poisson_pmf(4.0, k)),
    ("BINOMIAL_CLAMP", [10, 0.3], 5, lambda k: binomial_pmf(10, 0.3, k)),
    ("POISSON_NONNEG", [4.0], 0, lambda k: poisson_pmf(4.0, k)),
]


@pytest.mark.parametrize("name,params,init,pmf", CASES)
def test_oracle_passes_the_references_own_assertions(O, name, params, init, pmf):
    # test_poisson_mh / test_binomial_mh: one chain, .seed(42), run(20_000, 2_000), |freq - pmf| < 0.05 for k = 0..10
    kind = getattr(O, name)
    mh = O.DiscreteMH(kind, params, [init]).seed(42).proposal_seed(7)
    sample = mh.run(20_000, 2_000).ravel()
    for k in range(11):
        assert abs(np.mean(sample == k) - pmf(k)) < 0.05, (name, k)
    assert 0 < mh.accept_counts[0] <= 22_000


@pytest.mark.parametrize("name,params,init,pmf", CASES)
def test_host_build_equals_oracle_on_the_engine_stream(O, name, params, init, pmf):
    kind = getattr(O, name)
    rng = np.random.default_rng(3)
    start = rng.integers(0, 9, size=37).astype(np.int32)
    out_e, st_e, acc_e = O.engine_host_discrete_run(kind, params, start, 400, 60, seed=11, chain_offset=5)
    mh = O.DiscreteMH(kind, params, start).use_engine_stream(11, 5)
    out_o = mh.run(400, 60)
    # integer states: exact.  (The oracle sums ln k! on every call, the engine reads the table built with the same sum.)
    assert np.array_equal(out_e, out_o) and np.array_equal(st_e, mh.state()) and np.array_equal(acc_e, mh.accept_counts)
    assert out_e.min() >= 0 and (name != "BINOMIAL_CLAMP" or out_e.max() <= 10)


def test_discrete_abi_symbols_and_no_device():
    import ctypes as C

    import mini_mcmc_amd

    lib = mini_mcmc_amd.lib()
    for sym in ("create", "seed", "set_chain_offset", "run", "state", "accept_counts", "sync", "destroy"):
        assert hasattr(lib, "mmcmc_mh_discrete_" + sym)
    import torch

    if not torch.cuda.is_available():
        h = C.c_void_p()
        p = (C.c_double * 1)(4.0)
        init = (C.c_int32 * 1)(0)
        assert lib.mmcmc_mh_discrete_create(C.byref(h), 16, p, init, 1, 0) == -4  # MMCMC_ERR_NO_DEVICE: no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,init,pmf", CASES)
def test_gpu_discrete_bit_exact_and_pmf(O, name, params, init, pmf):
    from mini_mcmc_amd import discrete as D

    model = {"POISSON_REFLECT": D.PoissonReflect(4.0), "BINOMIAL_CLAMP": D.BinomialClamp(10, 0.3),
             "POISSON_NONNEG": D.PoissonNonneg(4.0)}[name]
    kind = getattr(O, name)
    rng = np.random.default_rng(5)
    for C_, nc, nd, off in ((1, 300, 20, 0), (130, 257, 63, 0), (70, 64, 0, 1 << 33), (64, 1, 5, 9)):
        start = rng.integers(0, 9, size=C_).astype(np.int32)
        s = D.DiscreteMetropolisHastings(model, start).seed(21)
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        ref, st, acc = O.engine_host_discrete_run(kind, params, start, nc, nd, seed=21, chain_offset=off)
        assert out.shape == (C_, nc, 1) and out.dtype == np.int32
        assert np.array_equal(out[:, :, 0], ref) and np.array_equal(s.state(), st), (name, C_)
        assert np.array_equal(s.accept_counts(), acc), (name, C_)  # accept counts: exact
        # a second run continues the chain and the stream
        out2 = s.run(33, 7)
        ref2, st2, acc2 = O.engine_host_discrete_run(kind, params, st, 33, 7, seed=21, chain_offset=off, iter0=nc + nd)
        assert np.array_equal(out2[:, :, 0], ref2) and np.array_equal(s.accept_counts(), acc + acc2)
    # full size: 65 536 chains from the reference's start; the histogram is the pmf to 3 decimals, a checksum of the
    # device sample equals the host build's on a sub-block, and sharding by chain offset reproduces the unsharded run
    C_ = 65536
    start = np.full(C_, init, dtype=np.int32)
    s = D.DiscreteMetropolisHastings(model, start).seed(42)
    out = s.run(200, 300, to="torch")
    flat = out.reshape(-1).cpu().numpy()
    for k in range(11):
        assert abs(np.mean(flat == k) - pmf(k)) < 2e-3, (name, k)
    ref, _, acc = O.engine_host_discrete_run(kind, params, start[:512], 200, 300, seed=42)
    assert np.array_equal(out[:512, :, 0].cpu().numpy(), ref) and np.array_equal(s.accept_counts()[:512], acc)
    hi = D.DiscreteMetropolisHastings(model, start[40000:]).seed(42).set_chain_offset(40000).run(200, 300)
    assert np.array_equal(hi, out[40000:].cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,init,pmf", CASES)
def test_gpu_discrete_split_kernel_batches_tiles_and_band(O, name, params, init, pmf):
    """The four-waves-per-SIMD kernel (from 4096 chains on; round 3: tabulated accept ratios, 4-byte draws, batches of
    12 transitions cut at n_discard, tiles of 48 samples) against the one-wave kernel and the host build: run lengths
    around every batch / tile boundary, a ragged last workgroup, continued runs, chains that start outside the table
    (their wave keeps the plain transition), and -- over all 65 536 chains -- the transitions whose ratio falls inside
    the accept filter's band (about 4 in 10^6: the batch is run again by the exact loop)."""
    from mini_mcmc_amd import discrete as D

    model = {"POISSON_REFLECT": D.PoissonReflect(4.0), "BINOMIAL_CLAMP": D.BinomialClamp(10, 0.3),
             "POISSON_NONNEG": D.PoissonNonneg(4.0)}[name]
    kind = getattr(O, name)
    rng = np.random.default_rng(11)
    C_ = 4096 + 77
    start = rng.integers(0, 9, size=C_).astype(np.int32)
    start[4000:4003] = 5000  # beyond the table: log-density -inf
    if name != "BINOMIAL_CLAMP":
        start[130] = 1023  # the last tabulated state
    for nc, nd in ((1, 0), (11, 1), (12, 12), (13, 11), (47, 13), (48, 0), (49, 24), (100, 5), (96, 36), (7, 100)):
        a = D.DiscreteMetropolisHastings(model, start).seed(3)
        b = D.DiscreteMetropolisHastings(model, start).seed(3).set_kernel_variant(0)
        oa, ob = a.run(nc, nd), b.run(nc, nd)
        assert np.array_equal(oa, ob), (name, nc, nd)
        assert np.array_equal(a.state(), b.state()) and np.array_equal(a.accept_counts(), b.accept_counts()), (name, nc, nd)
        ref, st, acc = O.engine_host_discrete_run(kind, params, start[3900:4200], nc, nd, seed=3, chain_offset=3900)
        assert np.array_equal(oa[3900:4200, :, 0], ref) and np.array_equal(a.accept_counts()[3900:4200], acc), (name, nc, nd)
        oa2, ob2 = a.run(50, 3), b.run(50, 3)  # continued: stream position and state carry over
        assert np.array_equal(oa2, ob2) and np.array_equal(a.accept_counts(), b.accept_counts()), (name, nc, nd)
    # no output buffer: states and counts only
    a = D.DiscreteMetropolisHastings(model, start).seed(3)
    b = D.DiscreteMetropolisHastings(model, start).seed(3).set_kernel_variant(0)
    a.run(0, 77), b.run(0, 77)
    assert np.array_equal(a.state(), b.state()) and np.array_equal(a.accept_counts(), b.accept_counts())
    # every chain of the full-size run, both kernels (3e7 transitions: ~100 of them inside the band)
    C_ = 65536
    start = np.full(C_, init, dtype=np.int32)
    a = D.DiscreteMetropolisHastings(model, start).seed(42)
    b = D.DiscreteMetropolisHastings(model, start).seed(42).set_kernel_variant(0)
    oa, ob = a.run(400, 100, to="torch"), b.run(400, 100, to="torch")
    assert bool((oa == ob).all()) and np.array_equal(a.accept_counts(), b.accept_counts())
