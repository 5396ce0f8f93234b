"""Any dimension (run-time D, mini_mcmc_amd/csrc/mm_generic.h): the reference's N-dimensional targets take any D
(RosenbrockND distributions.rs:531-547, IsotropicGaussian :350-402, StandardNormal nuts.rs:1027-1037); the
register-resident kernels exist for D in {1..8, 16, 32}, every other dimension runs through the store-based path.

CPU: the host build of the run-time-D path equals the host build of the fixed-D path bit for bit wherever both exist
(so the two kernels families define ONE function of (seed, chain, iteration)), and agrees with the reference-ordered
oracle to rounding at D = 10, 24, 100.  GPU (-m gpu): the kernels equal that host build bit for bit, in LDS and in the
HBM store, and the run-time-D kernel equals the fixed-D kernels on the GPU itself."""
import numpy as np
import pytest


def _kinds(O, dim, rng):
    A = rng.standard_normal((dim, dim))
    A = A @ A.T / dim + np.eye(dim)
    return [
        ("RosenbrockND", O.ROSENBROCK_ND, [], None, 0.01, lambda: O.rosenbrock_nd(dim)),
        ("StandardNormal", O.STANDARD_NORMAL, [], None, 0.3, lambda: O.standard_normal(dim)),
        ("IsotropicGaussian", O.ISOTROPIC_GAUSSIAN, [1.5], None, 0.4, lambda: O.isotropic_gaussian(1.5, dim)),
        ("GaussianND", O.GAUSSIAN_ND, [], A, 0.15, lambda: O.gaussian_nd(A)),
    ]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_generic_path_equals_host_fixed_path(O, dtype):
    rng = np.random.default_rng(1)
    for dim in (1, 2, 3, 5, 8, 16, 32):
        for name, kind, params, mat, scale, _ in _kinds(O, dim, rng):
            init = O.init_with_seed(20, dim, 3, dtype) * dtype(0.5)
            for smp in ("mh", "hmc"):
                a = O.engine_host_run(smp, kind, dim, params, init, scale, 15, 3, seed=7, n_leapfrog=6, matrix=mat, dtype=dtype,
                                      chain_offset=9, n_threads=2)
                b = O.engine_host_run(smp + "_generic", kind, dim, params, init, scale, 15, 3, seed=7, n_leapfrog=6, matrix=mat,
                                      dtype=dtype, chain_offset=9, n_threads=2)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, dim, smp)
            x = (rng.standard_normal((33, dim)) * 0.7).astype(dtype)
            lp, g = O.engine_host_logp_grad(kind, dim, params, x, matrix=mat, dtype=dtype)
    # noise: a prefix property of the schedule -- the first 8 normals of a 10-dimensional draw are the 8-dimensional draw
    z10, u10 = O.engine_host_noise(5, 2, 3, 6, 10, dtype)
    z8, u8 = O.engine_host_noise(5, 2, 3, 6, 8, dtype)
    assert np.array_equal(z10[:, :8], z8) and np.array_equal(u10, u8)


@pytest.mark.parametrize("dim", [10, 24, 100])
def test_host_generic_path_vs_reference_order(O, dim):
    """f64, D = 10 / 24 / 100: against oracle/mh_hmc.c (hmc.rs:304-431, metropolis_hastings.rs:303-315 in the reference's
    order) on the same noise: identical accept counts, samples to rounding."""
    rng = np.random.default_rng(dim)
    for name, kind, params, mat, scale, mk in _kinds(O, dim, rng):
        sc = scale / np.sqrt(dim / 3.0)
        init = O.init_with_seed(32, dim, 6, np.float64) * 0.5
        for smp in ("mh", "hmc"):
            out, st, acc = O.engine_host_run(smp, kind, dim, params, init, sc, 20, 5, seed=3, n_leapfrog=5, matrix=mat,
                                             dtype=np.float64, n_threads=2)
            o = (O.MetropolisHastings(mk(), sc, init, np.float64) if smp == "mh" else O.HMC(mk(), init, sc, 5, np.float64)).use_engine_stream(3)
            ref = o.run(20, 5, n_threads=2)
            assert np.array_equal(acc, o.accept_counts), (name, dim, smp)
            np.testing.assert_allclose(out, ref, rtol=1e-7, atol=1e-7, err_msg=f"{name} D={dim} {smp}")
            assert 0 < acc.mean() <= 25, (name, dim, smp, acc.mean())


# ------------------------------------------------------------------ GPU


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


def _targets(M, O, dim, rng):
    A = rng.standard_normal((dim, dim))
    A = A @ A.T / dim + np.eye(dim)
    return [
        (M.dist.RosenbrockND(dim), O.ROSENBROCK_ND, [], None, 0.01),
        (M.dist.StandardNormal(dim), O.STANDARD_NORMAL, [], None, 0.3),
        (M.dist.IsotropicGaussian(1.5, dim), O.ISOTROPIC_GAUSSIAN, [1.5], None, 0.4),
        (M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 0.15),
    ]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [10, 24, 100, 9, 257])
def test_gpu_any_dimension_bit_exact_vs_host_build(M, O, dtype, dim):
    """Dimensions without a register-resident kernel (RosenbrockND(10), IsotropicGaussian(sigma, 24), GaussianND(100), ...):
    samples, states and accept counts equal the host build bit for bit; the chain vectors sit in LDS (D = 9, 10, 24) or
    in the HBM store (D = 100, 257; f64 D = 24 too); ragged chain counts, chain offsets, continued handles."""
    rng = np.random.default_rng(dim)
    for tgt, kind, params, mat, scale in _targets(M, O, dim, rng)[: (2 if dim > 100 else 4)]:
        sc = scale / np.sqrt(dim / 3.0)
        for sampler, C, nc, nd, off in (("hmc", 70, 9, 3, 0), ("mh", 130, 12, 1, 1 << 33)):
            init = M.core.init_with_seed(C, dim, 7, dtype) * dtype(0.5)
            if sampler == "mh":
                s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(sc), init).seed(99)
            else:
                s = M.hmc.HMC(tgt, init, sc, 4).set_seed(99)
            if off:
                s.set_chain_offset(off)
            out = s.run(nc, nd)
            ref, st, acc = O.engine_host_run(sampler, kind, dim, params, init, sc, nc, nd, seed=99, chain_offset=off, n_leapfrog=4,
                                             matrix=mat, dtype=dtype)
            name = f"{sampler} {type(tgt).__name__} D={dim} {dtype.__name__}"
            assert out.shape == (C, nc, dim) and np.array_equal(out, ref), name
            assert np.array_equal(s.accept_counts, acc) and np.array_equal(s.state(), st), name
            out2 = s.run(3, 2)  # continues the chains and the stream
            ref2, _, acc2 = O.engine_host_run(sampler, kind, dim, params, st, sc, 3, 2, seed=99, chain_offset=off, iter0=nc + nd,
                                              n_leapfrog=4, matrix=mat, dtype=dtype)
            assert np.array_equal(out2, ref2) and np.array_equal(s.accept_counts, acc2), name
        # densities and gradients, and the noise, at this dimension
        x = (rng.standard_normal((70, dim)) * 0.7).astype(dtype)
        lp, g = tgt.unnorm_logp_batch(x, dtype, with_grad=True)
        lpr, gr = O.engine_host_logp_grad(kind, dim, params, x, matrix=mat, dtype=dtype)
        assert np.array_equal(lp, lpr) and np.array_equal(g, gr), (type(tgt).__name__, dim)
    z, u = M.core.draw_noise(11, 5, 7, 100, dim, dtype)
    zr, ur = O.engine_host_noise(11, 5, 7, 100, dim, dtype)
    assert np.array_equal(z, zr) and np.array_equal(u, ur)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gpu_generic_kernel_equals_fixed_dimension_kernels(M, O, dtype):
    """Kernel variant 6 (run-time D) on dimensions that also have register-resident kernels: identical bits on the GPU."""
    rng = np.random.default_rng(4)
    for dim in (3, 8):
        for tgt, kind, params, mat, scale in _targets(M, O, dim, rng):
            init = M.core.init_with_seed(100, dim, 5, dtype) * dtype(0.5)
            res = []
            for variant in (None, 6):
                s = M.hmc.HMC(tgt, init, scale / 3, 7).set_seed(4)
                if variant is not None:
                    s.set_kernel_variant(variant)
                res.append((s.run(21, 4), s.accept_counts.copy(), s.state()))
            assert all(np.array_equal(a, b) for a, b in zip(*res)), (type(tgt).__name__, dim)
    with pytest.raises(Exception):  # the 2-D special targets have no run-time-D form
        M.hmc.HMC(M.dist.Rosenbrock2D(1.0, 100.0), M.core.init_with_seed(4, 2, 1, dtype), 0.1, 5).set_kernel_variant(6)
    with pytest.raises(Exception):  # a dimension that only the run-time-D path serves has no other variant
        M.hmc.HMC(M.dist.RosenbrockND(10), M.core.init_with_seed(4, 10, 1, dtype), 0.01, 5).set_kernel_variant(2)


@pytest.mark.gpu
def test_gpu_any_dimension_vs_reference_order_and_posterior(M, O):
    """D = 10 / 24 / 100 in f64 against the reference-ordered oracle directly (identical accept counts, samples to
    rounding); IsotropicGaussian(2.0, 20) under HMC: pooled mean 0 and variance sigma^2 within 1 %."""
    for dim in (10, 24, 100):
        rng = np.random.default_rng(dim)
        A = rng.standard_normal((dim, dim))
        A = A @ A.T / dim + np.eye(dim)
        for tgt, otgt, scale in ((M.dist.RosenbrockND(dim), O.rosenbrock_nd(dim), 0.01), (M.dist.GaussianND(A), O.gaussian_nd(A), 0.15),
                                 (M.dist.IsotropicGaussian(1.5, dim), O.isotropic_gaussian(1.5, dim), 0.4)):
            sc = scale / np.sqrt(dim / 3.0)
            init = M.core.init_with_seed(64, dim, 6, np.float64) * 0.5
            s = M.hmc.HMC(tgt, init, sc, 5).set_seed(3)
            out = s.run(20, 5)
            o = O.HMC(otgt, init, sc, 5, np.float64).use_engine_stream(3)
            ref = o.run(20, 5, n_threads=2)
            assert np.array_equal(s.accept_counts, o.accept_counts), (type(tgt).__name__, dim)
            np.testing.assert_allclose(out, ref, rtol=1e-7, atol=1e-7)
            m = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(sc), init).seed(3)
            outm = m.run(20, 5)
            om = O.MetropolisHastings(otgt, sc, init, np.float64).use_engine_stream(3)
            refm = om.run(20, 5, n_threads=2)
            assert np.array_equal(m.accept_counts, om.accept_counts), (type(tgt).__name__, dim)
            np.testing.assert_allclose(outm, refm, rtol=1e-9, atol=1e-9)
    s = M.hmc.HMC(M.dist.IsotropicGaussian(2.0, 20), M.core.init_with_seed(4096, 20, 1, np.float32), 0.5, 8).set_seed(2)
    x = s.run(60, 40).reshape(-1, 20).astype(np.float64)
    assert np.abs(x.mean(axis=0)).max() < 0.02 and np.abs(x.var(axis=0) / 4.0 - 1.0).max() < 0.01
    assert 0.5 < s.accept_counts.mean() / 100 <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gpu_nuts_more_dimensions_bit_exact_vs_host_build(M, O, mode):
    """NUTS keeps its vectors in registers, so its dimensions are compile-time instances (csrc/mm_nuts_dims.h): every
    dimension up to 8 and 10, 16, 20 for the N-dimensional built-ins (24 / 32 for the dense Gaussian).  The added ones
    against the host build: samples, tree shapes, adaptation state."""
    from mini_mcmc_amd.nuts import NUTS

    rng = np.random.default_rng(8)

    def spd(d):
        A = rng.standard_normal((d, d))
        return A @ A.T / d + np.eye(d)

    A24, A5 = spd(24), spd(5)
    cases = [
        (M.dist.RosenbrockND(10), O.ROSENBROCK_ND, [], None),
        (M.dist.RosenbrockND(5), O.ROSENBROCK_ND, [], None),
        (M.dist.IsotropicGaussian(1.5, 20), O.ISOTROPIC_GAUSSIAN, [1.5], None),
        (M.dist.IsotropicGaussian(0.7, 6), O.ISOTROPIC_GAUSSIAN, [0.7], None),
        (M.dist.StandardNormal(16), O.STANDARD_NORMAL, [], None),
        (M.dist.GaussianND(A24), O.GAUSSIAN_ND, [], A24),
        (M.dist.GaussianND(A5), O.GAUSSIAN_ND, [], A5),
    ]
    for tgt, kind, params, mat in cases:
        init = M.core.init_with_seed(70, tgt.dim, 31) * 0.6
        s = NUTS(tgt, init, 0.8, mode=mode).set_seed(77)
        out = s._run(7, 6, False, "numpy")
        ref, pos, ad, nlf = O.engine_host_nuts_run(mode, kind, tgt.dim, params, init, 0.8, 7, 6, seed=77, matrix=mat)
        name = f"{type(tgt).__name__} D={tgt.dim} mode={mode}"
        assert np.array_equal(out, ref) and np.array_equal(s.positions(), pos), name
        assert np.array_equal(s.leapfrog_counts(), nlf), name
        a = s.adapt_state()
        assert np.array_equal(a["epsilon"], ad[:, 0]) and np.array_equal(a["h_bar"], ad[:, 2]), name


def _spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_generic_nuts_twin_equals_fixed_dimension_twin(O, mode):
    """mm_nuts_generic.h (run-time D, vectors in a store) is mm_nuts.h operation for operation: at the dimensions both
    have, the two host builds give the same bits -- samples, positions, adaptation state, tree shapes."""
    rng = np.random.default_rng(4)
    A6 = _spd(rng, 6)
    for kind, dim, params, mat in ((O.ROSENBROCK_ND, 3, [], None), (O.ROSENBROCK_ND, 10, [], None), (O.STANDARD_NORMAL, 5, [], None),
                                   (O.ISOTROPIC_GAUSSIAN, 7, [1.5], None), (O.GAUSSIAN_ND, 6, [], A6), (O.STANDARD_NORMAL, 1, [], None)):
        init = O.init_with_seed(9, dim, 12) * 0.7
        for progress in (False, True):
            a = O.engine_host_nuts_run(mode, kind, dim, params, init, 0.8, 9, 6, seed=31, matrix=mat, progress=progress, n_threads=2)
            b = O.engine_host_nuts_run(mode + 4, kind, dim, params, init, 0.8, 9, 6, seed=31, matrix=mat, progress=progress, n_threads=2)
            for u, v in zip(a, b):
                assert np.array_equal(u, v), (kind, dim, mode, progress)


def test_generic_nuts_twin_vs_recursive_oracle_at_new_dimensions(O):
    """Dimensions without a fixed-D instance (RosenbrockND(11), GaussianND(40), IsotropicGaussian(.., 100)): the
    run-time-D twin against the recursive restatement of nuts.rs (pinned by the reference's seeded vectors) on the
    engine's stream, f64 -- identical tree shapes, samples to the rounding a trajectory amplifies."""
    rng = np.random.default_rng(5)
    A40 = _spd(rng, 40)
    for kind, dim, params, mat, otgt in ((O.ROSENBROCK_ND, 11, [], None, O.rosenbrock_nd(11)), (O.GAUSSIAN_ND, 40, [], A40, O.gaussian_nd(A40)),
                                         (O.ISOTROPIC_GAUSSIAN, 100, [1.5], None, O.isotropic_gaussian(1.5, 100)),
                                         (O.STANDARD_NORMAL, 33, [], None, O.standard_normal(33))):
        init = O.init_with_seed(6, dim, 17) * 0.5
        out_e, pos_e, ad_e, nlf = O.engine_host_nuts_run(2, kind, dim, params, init, 0.8, 12, 10, seed=5, matrix=mat, n_threads=2)
        s = O.NUTS(otgt, init, 0.8, mode=2).use_engine_stream(5).set_max_depth(10)
        out_o = s.run(12, 10, n_threads=2)
        lf_o = np.array([s.chain_state(i)["n_leapfrog_total"] for i in range(6)])
        assert np.array_equal(nlf.astype(np.int64), lf_o.astype(np.int64)), (kind, dim)
        np.testing.assert_allclose(out_e, out_o, rtol=1e-6, atol=1e-6, err_msg=f"kind {kind} D {dim}")
        eps_o = np.array([s.chain_state(i)["epsilon"] for i in range(6)])
        np.testing.assert_allclose(ad_e[:, 0], eps_o, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gpu_nuts_any_dimension_bit_exact_vs_host_build(M, O, mode):
    """NUTS at dimensions without a fixed-D instance runs the run-time-D kernel (mm_nuts_generic.h, variant 6): bit-exact
    against its host build; and where both exist (variant 6 forced) it equals the fixed-D kernel."""
    from mini_mcmc_amd.nuts import NUTS

    rng = np.random.default_rng(8)
    A40, A5 = _spd(rng, 40), _spd(rng, 5)
    for tgt, kind, params, mat in ((M.dist.RosenbrockND(11), O.ROSENBROCK_ND, [], None), (M.dist.IsotropicGaussian(1.5, 100), O.ISOTROPIC_GAUSSIAN, [1.5], None),
                                   (M.dist.StandardNormal(33), O.STANDARD_NORMAL, [], None), (M.dist.GaussianND(A40), O.GAUSSIAN_ND, [], A40)):
        init = M.core.init_with_seed(70, tgt.dim, 31) * 0.5
        s = NUTS(tgt, init, 0.8, mode=mode).set_seed(77)
        # up to dimension 32 the functor is compiled into the one-chain-per-lane kernel on first use (variant 7), beyond
        # that the run-time-D kernel runs (6); both must equal the host build
        assert s.kernel_variant == (7 if tgt.dim <= 32 else 6)
        if tgt.dim <= 32:
            g6 = NUTS(tgt, init, 0.8, mode=mode).set_seed(77).set_kernel_variant(6)
            assert np.array_equal(g6._run(7, 6, False, "numpy"), s._run(7, 6, False, "numpy")) and np.array_equal(g6.positions(), s.positions())
            s = NUTS(tgt, init, 0.8, mode=mode).set_seed(77)
        out = s._run(7, 6, False, "numpy")
        ref, pos, ad, nlf = O.engine_host_nuts_run(mode, kind, tgt.dim, params, init, 0.8, 7, 6, seed=77, matrix=mat)
        name = f"{type(tgt).__name__} D={tgt.dim} mode={mode}"
        assert np.array_equal(out, ref) and np.array_equal(s.positions(), pos), name
        assert np.array_equal(s.leapfrog_counts(), nlf), name
        a = s.adapt_state()
        assert np.array_equal(a["epsilon"], ad[:, 0]) and np.array_equal(a["h_bar"], ad[:, 2]), name
        # a second run continues the chain (self.m, adaptation state) exactly like the host build does
        out2 = s._run(5, 0, False, "numpy")
        ref2, pos2, ad2, nlf2 = O.engine_host_nuts_run(mode, kind, tgt.dim, params, pos, 0.8, 5, 0, seed=77, matrix=mat, m0=12, adapt=ad)
        assert np.array_equal(out2, ref2) and np.array_equal(s.positions(), pos2), name
    # a different depth cap (the store is re-sized), run_progress stepping, one dimension, a chain count that is not a multiple of 64
    tgt = M.dist.StandardNormal(1)
    init = M.core.init_with_seed(37, 1, 2) * 0.5
    s = NUTS(tgt, init, 0.8, mode=mode).set_seed(3).set_kernel_variant(6).set_max_depth(4)
    out = s._run(5, 5, True, "numpy")
    ref, pos, ad, nlf = O.engine_host_nuts_run(mode + 4, O.STANDARD_NORMAL, 1, [], init, 0.8, 5, 5, seed=3, progress=True, max_depth=4)
    assert np.array_equal(out, ref) and np.array_equal(s.leapfrog_counts(), nlf)
    s.set_max_depth(12)
    out2 = s._run(5, 0, True, "numpy")
    ref2, _, _, _ = O.engine_host_nuts_run(mode + 4, O.STANDARD_NORMAL, 1, [], pos, 0.8, 5, 0, seed=3, progress=True, max_depth=12, m0=10, adapt=ad)
    assert np.array_equal(out2, ref2)
    for tgt in (M.dist.RosenbrockND(3), M.dist.GaussianND(A5), M.dist.IsotropicGaussian(0.7, 6)):
        init = M.core.init_with_seed(130, tgt.dim, 3) * 0.6
        a = NUTS(tgt, init, 0.8, mode=mode).set_seed(5)
        b = NUTS(tgt, init, 0.8, mode=mode).set_seed(5).set_kernel_variant(6)
        assert np.array_equal(a._run(6, 8, True, "numpy"), b._run(6, 8, True, "numpy")), type(tgt).__name__
        assert np.array_equal(a.leapfrog_counts(), b.leapfrog_counts())
        assert np.array_equal(a.depth_histogram(), b.depth_histogram())


@pytest.mark.gpu
def test_gpu_nuts_any_dimension_vs_recursive_oracle_and_posterior(M, O):
    """The run-time-D NUTS kernel directly against the recursive restatement of nuts.rs (f64, engine stream) at D = 11 /
    40 / 100: identical tree shapes, samples to rounding; and IsotropicGaussian(2, 20): posterior mean and variance."""
    from mini_mcmc_amd.nuts import NUTS

    rng = np.random.default_rng(9)
    A40 = _spd(rng, 40)
    for tgt, otgt in ((M.dist.RosenbrockND(11), O.rosenbrock_nd(11)), (M.dist.GaussianND(A40), O.gaussian_nd(A40)),
                      (M.dist.IsotropicGaussian(1.5, 100), O.isotropic_gaussian(1.5, 100))):
        init = M.core.init_with_seed(48, tgt.dim, 17) * 0.5
        s = NUTS(tgt, init, 0.8, mode=2).set_seed(5)
        out = s._run(10, 8, False, "numpy")
        o = O.NUTS(otgt, init, 0.8, mode=2).use_engine_stream(5).set_max_depth(10)
        ref = o.run(10, 8, n_threads=2)
        lf_o = np.array([o.chain_state(i)["n_leapfrog_total"] for i in range(48)])
        assert np.array_equal(s.leapfrog_counts().astype(np.int64), lf_o.astype(np.int64)), type(tgt).__name__
        np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    s = NUTS(M.dist.IsotropicGaussian(2.0, 20), M.core.init_with_seed(4096, 20, 1) * 0.5, 0.8, mode=0).set_seed(2)
    x = s._run(60, 60, True, "numpy").reshape(-1, 20).astype(np.float64)
    assert np.abs(x.mean(axis=0)).max() < 0.03 and np.abs(x.var(axis=0) / 4.0 - 1.0).max() < 0.02


@pytest.mark.gpu
def test_gpu_nuts_run_time_compiled_units_at_the_dimensions_the_fuzz_found(M, O):
    """tools/fuzz_variants.py (round 3) found run-time compiled NUTS kernels that did not reproduce themselves: the unit's
    lanes-in-step kernel at RosenbrockND(19) / (23) in f64 (wrong samples) and StandardNormal(25) in f32 (memory fault),
    cause not found (the same template compiled into the library is correct).  That kernel is no longer launched and every unit is verified before use
    (mm_nuts_api.hip: rtc_unit_verified); here: the default mapping at those dimensions equals the run-time-D kernel and
    itself, bit for bit."""
    from mini_mcmc_amd.nuts import NUTS

    for tgt, mode in ((M.dist.RosenbrockND(19), 2), (M.dist.RosenbrockND(23), 2), (M.dist.StandardNormal(25), 0), (M.dist.StandardNormal(25), 2)):
        init = M.core.init_with_seed(77, tgt.dim, 31) * 0.5
        outs = []
        for v in (None, None, 6):
            s = NUTS(tgt, init, 0.8, mode=mode).set_seed(5)
            if v is not None:
                s.set_kernel_variant(v)
            else:
                assert s.kernel_variant in (7, 6)  # 6 if the unit's check refused it
            outs.append((s._run(4, 7, False, "numpy"), s.leapfrog_counts()))
        for o, l in outs[1:]:
            assert np.array_equal(o, outs[0][0]) and np.array_equal(l, outs[0][1]), (type(tgt).__name__, tgt.dim, mode)
