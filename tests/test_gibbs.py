"""Gibbs sampling with the built-in mixture conditional (the part that needs no Rust toolchain):
GibbsSampler (gibbs.rs:139-203) over MixtureConditional (gibbs.rs:231-285).

CPU: the oracle (oracle/gibbs.c) on the reference's stream passes the reference's own assertions
(test_gibbs_sampler_mixture_1 / _2: pooled mean and variance of x within 10 % of the mixture's); on the engine's stream
it agrees with the host build of the product's mm_gibbs.h -- z exactly, x to an ulp (the oracle adds mean + std * z
unfused).  GPU: the kernel equals that host build bit for bit, and reproduces the mixture's moments."""
import numpy as np
import pytest

SETS = [(-2.0, 1.0, 3.0, 1.5, 0.5), (-42.0, 69.0, 1.0, 2.0, 0.123)]


def _theory(mu0, s0, mu1, s1, pi0):
    mean = pi0 * mu0 + (1 - pi0) * mu1
    var = pi0 * (s0**2 + (mu0 - mean) ** 2) + (1 - pi0) * (s1**2 + (mu1 - mean) ** 2)
    return mean, var


@pytest.mark.parametrize("params", SETS)
def test_oracle_passes_the_references_mixture_assertions(O, params):
    # assert_mixture_simulation (gibbs.rs:330-375): 4 chains from init_det(4, 2), conditional rng and sampler seed 42,
    # run(100_000, 10_000); |mean - theo| < |theo| / 10, |var - theo| < theo / 10 over the pooled x
    g = O.GibbsMixture(params, O.init_det(4, 2), conditional_seed=42)
    x = g.run(100_000, 10_000)[:, :, 0].ravel()
    mean, var = _theory(*params)
    assert abs(x.mean() - mean) < abs(mean) / 10 and abs(x.var(ddof=1) - var) < var / 10
    zs = g.run(1000, 0)[:, :, 1]
    assert set(np.unique(zs)) <= {0.0, 1.0}
    assert np.array_equal(zs[0], zs[1])  # every chain holds a clone of the conditional's generator: identical draws


@pytest.mark.parametrize("params", SETS)
def test_host_build_agrees_with_oracle_on_the_engine_stream(O, params):
    init = O.init_det(33, 2)
    out_e, st_e = O.engine_host_gibbs_mixture_run(params, init, 300, 40, seed=5, chain_offset=3)
    g = O.GibbsMixture(params, init).use_engine_stream(5, 3)
    out_o = g.run(300, 40)
    assert np.array_equal(out_e[:, :, 1], out_o[:, :, 1])  # the latent indicator: exact
    # one rounding of sigma * z before the add: an ulp of |mu| + |sigma z| (up to ~300 in the second set)
    np.testing.assert_allclose(out_e[:, :, 0], out_o[:, :, 0], rtol=4e-16, atol=1e-13)
    np.testing.assert_allclose(st_e, g.state(), rtol=4e-16, atol=1e-13)


def test_gibbs_abi_symbols_and_no_device():
    import ctypes as C

    import mini_mcmc_amd

    lib = mini_mcmc_amd.lib()
    for sym in ("create", "seed", "set_chain_offset", "run", "state", "sync", "destroy"):
        assert hasattr(lib, "mmcmc_gibbs_mixture_" + sym)
    import torch

    if not torch.cuda.is_available():
        h = C.c_void_p()
        p = (C.c_double * 5)(-2.0, 1.0, 3.0, 1.5, 0.5)
        init = (C.c_double * 2)(0.0, 0.0)
        assert lib.mmcmc_gibbs_mixture_create(C.byref(h), p, init, 1, 0) == -4  # MMCMC_ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("params", SETS)
def test_gpu_gibbs_bit_exact_and_moments(O, params):
    from mini_mcmc_amd.core import init_det, init_with_seed
    from mini_mcmc_amd.gibbs import GibbsSampler, MixtureConditional

    cond = MixtureConditional(*params)
    for C_, nc, nd, off in ((4, 100, 10, 0), (130, 37, 5, 0), (70, 16, 0, 1 << 33), (64, 1, 3, 7)):
        init = init_with_seed(C_, 2, 11)
        s = GibbsSampler(cond, init).set_seed(42)
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        ref, st = O.engine_host_gibbs_mixture_run(params, init, nc, nd, seed=42, chain_offset=off)
        assert np.array_equal(out, ref) and np.array_equal(s.state(), st), (C_, nc)
        out2 = s.run(9, 2)  # continues the chains and the stream
        ref2, _ = O.engine_host_gibbs_mixture_run(params, st, 9, 2, seed=42, chain_offset=off, iter0=nc + nd)
        assert np.array_equal(out2, ref2)
    # the reference's own assertion at scale: 4096 chains from init_det, run(500, 100): pooled moments within 10 % ...
    s = GibbsSampler(cond, init_det(4096, 2)).set_seed(42)
    sample, stats = s.run_progress(500, 100)
    x = sample[:, :, 0].ravel()
    mean, var = _theory(*params)
    assert abs(x.mean() - mean) < abs(mean) / 10 and abs(x.var(ddof=1) - var) < var / 10
    # ... in fact within 1 % (north_star); the modes are far apart, so a chain switches rarely and the reference's
    # split-R-hat sqrt(W / var+) sits visibly below 1
    assert abs(x.mean() - mean) < 0.01 * np.sqrt(var) + 0.01 * abs(mean) and abs(x.var(ddof=1) - var) < 0.02 * var
    assert abs(sample[:, :, 1].mean() - (1 - params[4])) < 0.01
    assert 0.8 < stats.rhat.min <= stats.rhat.max <= 1.0 and stats.ess.min > 4096
    # full size (the four-waves-per-SIMD kernel, mm_gibbs_mixture_split_kernel, from 4096 chains on): a sub-block equals the
    # host build, the whole sample the one-wave kernel's, and sharding by chain offset reproduces the unsharded run; odd
    # counts leave ragged batches and tiles
    import os

    C_ = 20000 + 37
    init = init_with_seed(C_, 2, 3)
    g = GibbsSampler(cond, init).set_seed(42)
    out = g.run(131, 17)
    ref, st = O.engine_host_gibbs_mixture_run(params, init[:300], 131, 17, seed=42)
    assert np.array_equal(out[:300], ref) and np.array_equal(g.state()[:300], st)
    p = GibbsSampler(cond, init).set_seed(42).set_kernel_variant(0)  # one wave per SIMD: bit-identical
    assert np.array_equal(p.run(131, 17), out) and np.array_equal(p.state(), g.state())
    hi = GibbsSampler(cond, init[9000:]).set_seed(42).set_chain_offset(9000).run(131, 17)
    assert np.array_equal(hi, out[9000:])
    out2 = g.run(7, 0)
    ref2, _ = O.engine_host_gibbs_mixture_run(params, st, 7, 0, seed=42, iter0=148)
    assert np.array_equal(out2[:300], ref2)
