"""GPU (through the C ABI) against the REFERENCE-ORDERED oracle, directly -- no host build of the product's headers
in between.  Run with `-m gpu`.

tests/test_gpu_parity.py proves hipcc + gfx950 == g++ for the engine's own source (bit for bit); what is proved HERE
is that the kernels compute the reference's algorithm: the other side of every comparison is oracle/mh_hmc.c
(hmc.rs:304-431, metropolis_hastings.rs:303-315 op for op: unmerged half-kicks, unfused arithmetic, logp recomputed,
q-terms kept), oracle/nuts.c (the RECURSIVE build_tree of nuts.rs:764-946, pinned by the reference's seeded vectors
nuts.rs:1050-1222), oracle/discrete.c and oracle/gibbs.c, all driven by the engine's counter-based stream as restated
independently in oracle/orng.c.  Agreement is to rounding (fused vs unfused arithmetic), bit-exact for integer work:
accept counts, tree shapes (leapfrog counts), integer states.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAUSS = [0.0, 1.0, 4.0, 2.0, 2.0, 3.0]


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


# ---------------------------------------------------------------- the stream, against scipy / numpy (no shared header)


def test_gpu_noise_against_scipy_and_numpy(M, O):
    """The oracle restates the engine's stream but reads the inverse-CDF table and mm_log / mm_sincos2pi from the
    product's own headers (oracle/orng.c), so a wrong table row or polynomial would be invisible to both.  Here the
    GPU's normals and uniforms are held against functions neither side owns: the Philox words come from the oracle's
    independent restatement (pinned by Random123's known-answer vectors in tests/test_engine_stream.py), the f32 normals
    are -Phi^-1(n 2^-25) by scipy.special.ndtri, the f64 ones Box-Muller by numpy's libm."""
    import ctypes as C

    from scipy.special import ndtri

    seed, off, it, n, dim = 987654321, 5, 41, 4096, 8
    z, u = M.core.draw_noise(seed, off, it, n, dim, np.float32)
    L = O.lib()
    words = np.zeros((n, dim // 4, 4), dtype=np.uint32)
    buf = (C.c_uint32 * 4)()
    for c in range(n):
        for b in range(dim // 4):
            L.o_engine_block(seed, off + c, it, b, buf)
            words[c, b] = list(buf)
    w = words.reshape(n, dim).astype(np.uint64)
    nn = ((w >> np.uint64(8)) | np.uint64(1)).astype(np.float64)
    sign = np.where((w >> np.uint64(8)) & np.uint64(1), -1.0, 1.0)
    ref = sign * -ndtri(nn * 2.0**-25)
    assert np.max(np.abs(z.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))) < 2e-7
    assert np.abs(z).max() <= 5.42 and abs(z.mean()) < 0.02 and abs(z.var() - 1.0) < 0.02
    s = (w[:, 0] & np.uint64(255)) | ((w[:, 1] & np.uint64(255)) << np.uint64(8)) | ((w[:, 2] & np.uint64(255)) << np.uint64(16))
    assert np.array_equal(u, ((s + np.uint64(1)).astype(np.float64) * 2.0**-24).astype(np.float32))
    # f64: z[2b], z[2b + 1] = Box-Muller of the 53-bit uniforms of block b; accept uniform = u53 of the AUX block
    z64, u64 = M.core.draw_noise(seed, off, it, 512, 4, np.float64)
    for c in range(0, 512, 37):
        for b in range(2):
            L.o_engine_block(seed, off + c, it, b, buf)
            w0, w1, w2, w3 = [int(v) for v in buf]
            u1 = (((w0 << 21) | (w1 >> 11)) + 1) * 2.0**-53
            u2 = (((w2 << 21) | (w3 >> 11)) + 1) * 2.0**-53
            r = np.sqrt(-2.0 * np.log(u1))
            np.testing.assert_allclose(z64[c, 2 * b], r * np.cos(2 * np.pi * u2), rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(z64[c, 2 * b + 1], r * np.sin(2 * np.pi * u2), rtol=1e-12, atol=1e-13)
        L.o_engine_block(seed, off + c, it, 0x40000000, buf)
        assert u64[c] == (((int(buf[0]) << 21) | (int(buf[1]) >> 11)) + 1) * 2.0**-53


# ---------------------------------------------------------------- HMC / MH (hmc.rs:304-431, metropolis_hastings.rs:303-315)


def test_hmc_config3_f64_vs_reference_order(M, O):
    """BASELINE config 3 target and integrator (RosenbrockND(3), eps 0.032, L 10) in f64, 4096 chains, run(40, 5):
    identical accept counts for every chain, samples equal to the rounding a stiff trajectory amplifies."""
    C, nc, nd = 4096, 40, 5
    init = M.core.init_with_seed(C, 3, 42, np.float64)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    g = h.run(nc, nd)
    o = O.HMC(O.rosenbrock_nd(3), init, 0.032, 10, np.float64).use_engine_stream(42)
    ref = o.run(nc, nd)
    assert np.array_equal(h.accept_counts, o.accept_counts)  # 4096 x 45 accept decisions, all equal
    err = np.abs(g - ref) / np.maximum(1.0, np.abs(ref))
    # merged vs separate half-kicks and fused vs unfused products differ by ~1e-16 per operation; 450 leapfrog steps of
    # the Rosenbrock valley amplify that to 1e-6 in the worst of 4096 chains (2e-6 measured on the CPU), 1e-12 typically
    assert err.max() < 1e-5
    assert np.median(err.reshape(C, -1).max(axis=1)) < 1e-10
    np.testing.assert_allclose(h.state(), o.state(), rtol=1e-5, atol=1e-5)
    # the first transition from the initial point: nothing amplified yet
    h1 = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    g1 = h1.run(1, 0)
    o1 = O.HMC(O.rosenbrock_nd(3), init, 0.032, 10, np.float64).use_engine_stream(42)
    np.testing.assert_allclose(g1, o1.run(1, 0), rtol=1e-10, atol=1e-10)  # worst of 4096 chains: 4e-12
    assert np.array_equal(h1.accept_counts, o1.accept_counts)


def test_hmc_config3_f32_full_size_vs_reference_order(M, O):
    """BASELINE config 3 at full size (65 536 chains, f32, 400 + 50): in f32 single trajectories of a chaotic target
    drift apart, so the bar is north_star's: accept counts agree for 98.5 % of the chains (measured on the CPU, host build
    of the engine's headers against oracle/mh_hmc.c: 0.98495; the GPU equals that host build bit for bit, so the
    assertion is the measured figure minus a 0.5 % margin) and the pooled posterior mean / covariance are within 1 % of
    the CPU reference order's."""
    C, nc, nd = 65536, 400, 50
    init = M.core.init_with_seed(C, 3, 42, np.float32)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    g = h.run(nc, nd)
    o = O.HMC(O.rosenbrock_nd(3), init, 0.032, 10, np.float32).use_engine_stream(42)
    ref = o.run(nc, nd)
    assert np.mean(h.accept_counts == o.accept_counts) > 0.98  # measured 0.98495
    assert abs(h.accept_counts.mean() - o.accept_counts.mean()) < 0.1  # measured 0.021 of 450 transitions
    fg, fo = g.reshape(-1, 3).astype(np.float64), ref.reshape(-1, 3).astype(np.float64)
    co = np.cov(fo.T)
    sd = np.sqrt(np.diag(co))
    assert np.all(np.abs(fg.mean(axis=0) - fo.mean(axis=0)) < 0.01 * sd)  # measured on the CPU: 3e-4 sd
    assert np.all(np.abs(np.cov(fg.T) - co) < 0.01 * np.outer(sd, sd))    # measured: 1.2e-3
    # the first 8 transitions (before the drift): same accept decisions almost everywhere, samples to f32 rounding
    h8 = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    g8 = h8.run(8, 0)
    o8 = O.HMC(O.rosenbrock_nd(3), init, 0.032, 10, np.float32).use_engine_stream(42)
    r8 = o8.run(8, 0)
    same = h8.accept_counts == o8.accept_counts
    assert same.mean() > 0.999  # measured 0.99960
    err = (np.abs(g8[same][:, 0] - r8[same][:, 0]) / np.maximum(1.0, np.abs(r8[same][:, 0]))).max(axis=1)
    # CPU measurement of the same comparison: median 1.2e-7, 99.9 % quantile 8e-5, worst of 65 536 chains 4e-3 (starting
    # points far out in the tails, where one transition already amplifies an f32 ulp)
    assert np.median(err) < 1e-6 and np.quantile(err, 0.999) < 1e-3 and err.max() < 5e-2


@pytest.mark.parametrize("params", [[0.0, 0.0, 1.0, 0.0, 0.0, 1.0], GAUSS], ids=["config2", "correlated"])
def test_mh_config2_f32_full_size_vs_reference_order(M, O, params):
    """BASELINE config 2 at full size and in its own dtype (Gaussian2D, IsotropicGaussian(1) proposal, 65 536 chains,
    f32, run(1000, 100)) through the C ABI against oracle/mh_hmc.c (metropolis_hastings.rs:303-315 op for op: logp of
    the current point recomputed, q-terms kept, strict >) on the engine stream -- no host build of the product's headers
    in between.  Measured on the CPU first (host build of the engine's headers, which the GPU equals bit for bit,
    against the same oracle; both targets): 65 532 of 65 536 chains give the oracle's sample in EVERY one of the 1000
    collected draws (0.99994; a decision flips only where the cached f32 ratio and the recomputed one straddle ln u),
    accept counts differ by at most 13 on the four others, pooled mean / covariance agree to 2e-6 of a standard
    deviation, the first 8 transitions are bit-identical for all chains.  Asserted: those figures with a margin."""
    C, nc, nd = 65536, 1000, 100
    init = M.core.init_with_seed(C, 2, 42, np.float32)
    tgt = M.dist.Gaussian2D(params[:2], [params[2:4], params[4:6]])
    s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(1.0), init).seed(42)
    g = s.run(nc, nd)
    o = O.MetropolisHastings(O.gaussian2d(params[:2], [params[2:4], params[4:6]]), 1.0, init, np.float32).use_engine_stream(42)
    ref = o.run(nc, nd)
    same = s.accept_counts == o.accept_counts
    assert same.mean() >= 0.9998  # measured 0.99994 (4 chains of 65 536)
    assert np.abs(s.accept_counts.astype(np.int64) - o.accept_counts.astype(np.int64)).max() <= 40  # measured 13 / 8
    identical = (g == ref).all(axis=(1, 2))
    assert identical.mean() >= 0.9998  # whole 1000-draw samples, bit for bit
    fg, fo = g.reshape(-1, 2).astype(np.float64), ref.reshape(-1, 2).astype(np.float64)
    co = np.cov(fo.T)
    sd = np.sqrt(np.diag(co))
    assert np.all(np.abs(fg.mean(axis=0) - fo.mean(axis=0)) < 1e-4 * sd)          # north_star: 1 %; measured 2e-6
    assert np.all(np.abs(np.cov(fg.T) - co) < 1e-4 * np.outer(sd, sd))            # measured 1e-6
    # and against the exact posterior (north_star: moments within 1 % at 65 536 chains)
    mean, cov = np.array(params[:2]), np.array([params[2:4], params[4:6]])
    sde = np.sqrt(np.diag(cov))
    assert np.all(np.abs(fg.mean(axis=0) - mean) < 0.01 * sde)
    assert np.all(np.abs(np.cov(fg.T) - cov) < 0.01 * np.outer(sde, sde))
    # the first 8 transitions: every decision and every bit
    s8 = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(1.0), init).seed(42)
    g8 = s8.run(8, 0)
    o8 = O.MetropolisHastings(O.gaussian2d(params[:2], [params[2:4], params[4:6]]), 1.0, init, np.float32).use_engine_stream(42)
    r8 = o8.run(8, 0)
    assert np.array_equal(s8.accept_counts, o8.accept_counts)
    assert np.array_equal(g8, r8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mh_and_hmc_single_transition_vs_reference_order(M, O, dtype):
    """One transition of every built-in target from identical states on identical noise: the accept decision of every
    chain identical, new states within the stated rounding bound (3e-5 f32 / 1e-12 f64 relative)."""
    rng = np.random.default_rng(11)
    A = rng.standard_normal((8, 8))
    A = A @ A.T / 8 + np.eye(8)
    f32 = dtype == np.float32
    cases = [
        ("mh", M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 1.0, 0),
        ("mh", M.dist.IsotropicGaussian(1.3, 3), O.isotropic_gaussian(1.3, 3), 0.8, 0),
        ("mh", M.dist.RosenbrockND(4), O.rosenbrock_nd(4), 0.05, 0),
        ("mh", M.dist.GaussianND(A), O.gaussian_nd(A), 0.3, 0),
        ("hmc", M.dist.RosenbrockND(3), O.rosenbrock_nd(3), 0.032, 10),
        ("hmc", M.dist.Rosenbrock2D(1.0, 100.0), O.rosenbrock2d(1.0, 100.0), 0.01, 20),
        ("hmc", M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]),
         O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=f32), 0.1, 10),
        ("hmc", M.dist.StandardNormal(5), O.standard_normal(5), 0.3, 5),
        ("hmc", M.dist.GaussianND(A), O.gaussian_nd(A), 0.2, 6),
    ]
    for sampler, tgt, otgt, scale, L in cases:
        init = M.core.init_with_seed(512, tgt.dim, 5, dtype) * dtype(0.7)
        if sampler == "mh":
            s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(scale), init).seed(9)
            o = O.MetropolisHastings(otgt, scale, init, dtype).use_engine_stream(9)
        else:
            s = M.hmc.HMC(tgt, init, scale, L).set_seed(9)
            o = O.HMC(otgt, init, scale, L, dtype).use_engine_stream(9)
        s.run(1, 0)
        o.run(1, 0, n_threads=2)
        name = f"{sampler} {type(tgt).__name__} D={tgt.dim} {dtype.__name__}"
        assert np.array_equal(s.accept_counts, o.accept_counts), name
        st_g, st_o = s.state().astype(np.float64), o.state().astype(np.float64)
        tol = 3e-5 if f32 else 1e-12
        assert np.max(np.abs(st_g - st_o) / np.maximum(1.0, np.abs(st_o))) < tol, name
        assert 0.02 < s.accept_counts.mean() <= 1.0, name  # both branches taken


def test_short_f64_runs_vs_reference_order(M, O):
    """f64, 64 chains, run(40, 5), every built-in target: identical accept counts, samples to 1e-7."""
    rng = np.random.default_rng(11)
    A = rng.standard_normal((8, 8))
    A = A @ A.T / 8 + np.eye(8)
    cases = [
        ("mh", M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 1.0, 0),
        ("mh", M.dist.RosenbrockND(4), O.rosenbrock_nd(4), 0.05, 0),
        ("mh", M.dist.GaussianND(A), O.gaussian_nd(A), 0.3, 0),
        ("hmc", M.dist.RosenbrockND(3), O.rosenbrock_nd(3), 0.032, 10),
        ("hmc", M.dist.Rosenbrock2D(1.0, 100.0), O.rosenbrock2d(1.0, 100.0), 0.01, 20),
        ("hmc", M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]),
         O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 0.1, 10),
        ("hmc", M.dist.StandardNormal(5), O.standard_normal(5), 0.3, 5),
        ("hmc", M.dist.GaussianND(A), O.gaussian_nd(A), 0.2, 6),
    ]
    for sampler, tgt, otgt, scale, L in cases:
        init = M.core.init_with_seed(64, tgt.dim, 6, np.float64) * 0.7
        if sampler == "mh":
            s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(scale), init).seed(3)
            o = O.MetropolisHastings(otgt, scale, init, np.float64).use_engine_stream(3)
        else:
            s = M.hmc.HMC(tgt, init, scale, L).set_seed(3)
            o = O.HMC(otgt, init, scale, L, np.float64).use_engine_stream(3)
        g = s.run(40, 5)
        ref = o.run(40, 5, n_threads=2)
        name = f"{sampler} {type(tgt).__name__} D={tgt.dim}"
        assert np.array_equal(s.accept_counts, o.accept_counts), name
        np.testing.assert_allclose(g, ref, rtol=1e-7, atol=1e-7, err_msg=name)


def test_hmc_lane_group_mfma_vs_reference_order(M, O):
    """The MFMA lane-group HMC kernels (dense Gaussian, D = 16 / 32, mm_hmc_lg.h) against hmc.rs:304-431 as restated in
    oracle/mh_hmc.c: other summation order (MFMA accumulation, 4-lane butterflies), same algorithm."""
    for dim, cond, eps in ((32, 100.0, 0.05), (16, 50.0, 0.1)):
        g = M.dist.GaussianND.ill_conditioned(dim, cond, 7)
        init = M.core.init_with_seed(200, dim, 3, np.float64) * 0.3
        s = M.hmc.HMC(g, init, eps, 10).set_seed(19)
        assert s.kernel_variant == 3
        out = s.run(30, 5)
        o = O.HMC(O.gaussian_nd(g.precision), init, eps, 10, np.float64).use_engine_stream(19)
        ref = o.run(30, 5, n_threads=2)
        assert np.array_equal(s.accept_counts, o.accept_counts), dim
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9)
        # f32 kernel (v_mfma_f32_16x16x4_f32): first transition to f32 rounding, accept counts of a short run
        init32 = init.astype(np.float32)
        s32 = M.hmc.HMC(g, init32, eps, 10).set_seed(19)
        assert s32.kernel_variant == 3
        o32 = O.HMC(O.gaussian_nd(g.precision), init32, eps, 10, np.float32).use_engine_stream(19)
        a, b = s32.run(1, 0), o32.run(1, 0, n_threads=2)
        same = s32.accept_counts == o32.accept_counts
        assert same.mean() > 0.97, dim
        np.testing.assert_allclose(a[same], b[same], rtol=1e-3, atol=2e-4)


# ---------------------------------------------------------------- NUTS (nuts.rs:550-691, 764-946)


def _nuts_cases(M, O):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((4, 4))
    A = A @ A.T / 4 + np.eye(4)
    return [
        (M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]),
         lambda f32: O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=f32)),
        (M.dist.Rosenbrock2D(1.0, 100.0), lambda f32: O.rosenbrock2d(1.0, 100.0)),
        (M.dist.StandardNormal(3), lambda f32: O.standard_normal(3)),
        (M.dist.RosenbrockND(3), lambda f32: O.rosenbrock_nd(3)),
        (M.dist.GaussianND(A), lambda f32: O.gaussian_nd(A)),
    ]


def _oracle_nuts(O, otgt, init, mode, seed, nc, nd, progress=False, max_depth=10, offset=0):
    s = O.NUTS(otgt, init, 0.8, mode=mode).use_engine_stream(seed, offset).set_max_depth(max_depth)
    out = s.run(nc, nd, progress=progress, n_threads=2)
    n = init.shape[0]
    lf = np.array([s.chain_state(i)["n_leapfrog_total"] for i in range(n)]).astype(np.int64)
    eps = np.array([s.chain_state(i)["epsilon"] for i in range(n)])
    return out, lf, eps


@pytest.mark.parametrize("variant", [0, 4])
def test_nuts_f64_tree_shapes_equal_recursive_oracle(M, O, variant):
    """mode 2 (f64 tensors and scalars), both one-chain-per-lane mappings (lock-step lanes / asynchronous lanes): the
    kernel's iterative build_tree takes exactly the doublings, merges and early exits of the recursive nuts.rs:764-946
    -- identical leapfrog counts for every chain -- and the samples, step sizes agree to amplified rounding."""
    from mini_mcmc_amd.nuts import NUTS

    for tgt, mk in _nuts_cases(M, O):
        init = M.core.init_with_seed(70, tgt.dim, 17) * 0.8
        for progress in (False, True):
            s = NUTS(tgt, init, 0.8, mode=2).set_seed(5).set_max_depth(10).set_kernel_variant(variant)
            out = s._run(25, 15, progress, "numpy")
            ref, lf, eps = _oracle_nuts(O, mk(False), init, 2, 5, 25, 15, progress)
            name = f"{type(tgt).__name__} D={tgt.dim} variant={variant} progress={progress}"
            assert np.array_equal(s.leapfrog_counts().astype(np.int64), lf), name
            np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6, err_msg=name)
            np.testing.assert_allclose(s.adapt_state()["epsilon"], eps, rtol=1e-6, err_msg=name)
        # first transitions, nothing amplified: 1e-8 (rounding of fused vs unfused arithmetic only)
        s = NUTS(tgt, init, 0.8, mode=2).set_seed(5).set_kernel_variant(variant)
        out = s.run(4, 0)
        ref, lf, _ = _oracle_nuts(O, mk(False), init, 2, 5, 4, 0)
        assert np.array_equal(s.leapfrog_counts().astype(np.int64), lf)
        np.testing.assert_allclose(out, ref, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("mode", [0, 1])
def test_nuts_f32_modes_first_transitions_vs_recursive_oracle(M, O, mode):
    """modes 0 / 1 (f32 tensors; the reference's NUTS<f64, ..> and NUTS<f32, ..>): the first transitions from the
    initial point, before f32 rounding can flip a tree decision: row 0 is the initial position (nuts.rs:457-471), tree
    shapes agree for >= 75 % of the chains (CPU measurement of the same arithmetic: > 90 %) and those chains' samples
    agree to f32 rounding."""
    from mini_mcmc_amd.nuts import NUTS

    for tgt, mk in _nuts_cases(M, O):
        init = M.core.init_with_seed(128, tgt.dim, 23) * 0.8
        s = NUTS(tgt, init, 0.8, mode=mode).set_seed(9).set_max_depth(10)
        out = s.run(4, 0)
        ref, lf, _ = _oracle_nuts(O, mk(mode == 1), init, mode, 9, 4, 0)
        same = s.leapfrog_counts().astype(np.int64) == lf
        name = f"{type(tgt).__name__} D={tgt.dim} mode={mode}"
        assert same.mean() >= 0.75, (name, same.mean())
        np.testing.assert_allclose(out[same], ref[same], rtol=5e-4, atol=5e-4, err_msg=name)
        assert np.array_equal(out[:, 0].astype(np.float64), init.astype(np.float32).astype(np.float64)), name


@pytest.mark.parametrize("which", ["DiffableGaussian2D", "RosenbrockND3"])
def test_nuts_mode0_full_size_distribution_vs_recursive_oracle(M, O, which):
    """The reference's own type configuration, NUTS<f64, Autodiff<NdArray>> (mode 0: f32 tensors, f64 scalars;
    nuts.rs:1123-1222), at 65 536 chains, 200 warm-up + 100 recorded transitions (run_progress, nuts.rs:491-522), against
    the recursive restatement oracle/nuts.c in the same mode on the same stream.  f32 rounding separates individual
    trajectories after a few transitions (the test above), so what is compared is what the two runs must share as
    DISTRIBUTIONS: the histogram of tree depths over all 19.7 M transitions (chi-square of the two histograms), the
    adapted step sizes (mean and spread of epsilon_bar over the chains), and the pooled posterior mean / covariance of the
    6.5 M recorded draws within 1 % -- north_star's clause for the reference's actual NUTS types."""
    import os

    from mini_mcmc_amd.nuts import NUTS

    if which == "DiffableGaussian2D":
        tgt = M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
        otgt = O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=False)
    else:
        tgt, otgt = M.dist.RosenbrockND(3), O.rosenbrock_nd(3)
    C, nd, nc = 65536, 200, 100
    init = M.core.init_with_seed(C, tgt.dim, 42) * 0.5
    s = NUTS(tgt, init, 0.8, mode=0).set_seed(7).set_max_depth(10)
    out = s._run(nc, nd, True, "numpy").astype(np.float64)
    o = O.NUTS(otgt, init, 0.8, mode=0).use_engine_stream(7).set_max_depth(10)
    ref = o.run(nc, nd, progress=True, n_threads=os.cpu_count() or 1)
    h_ref, eps_ref, ebar_ref = o.depth_histogram()
    h_gpu = s.depth_histogram().astype(np.float64)
    h_ref = h_ref[:h_gpu.size].astype(np.float64)
    assert h_gpu.sum() == h_ref.sum() == C * (nd + nc)
    # two histograms of the same multinomial: sum (a - b)^2 / (a + b) ~ chi-square with (bins - 1) degrees of freedom
    big = (h_gpu + h_ref) >= 50
    chi2 = float((((h_gpu - h_ref) ** 2)[big] / (h_gpu + h_ref)[big]).sum())
    dof = int(big.sum()) - 1
    assert chi2 < 3.0 * dof + 10.0, (which, chi2, dof, h_gpu.tolist(), h_ref.tolist())
    ad = s.adapt_state()
    np.testing.assert_allclose(np.mean(ad["epsilon_bar"]), np.mean(ebar_ref), rtol=2e-3, err_msg=which)
    np.testing.assert_allclose(np.std(ad["epsilon_bar"]), np.std(ebar_ref), rtol=2e-2, err_msg=which)
    np.testing.assert_allclose(np.mean(ad["epsilon"]), np.mean(eps_ref), rtol=2e-3, err_msg=which)
    lf_gpu, lf_ref = float(s.leapfrog_counts().sum()), float(sum(o.chain_state(i)["n_leapfrog_total"] for i in range(0, C, 64)))
    assert lf_gpu > 0 and lf_ref > 0
    a, b = out.reshape(-1, tgt.dim), ref.reshape(-1, tgt.dim)
    sd = b.std(axis=0)
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) <= 0.01 * sd), (which, a.mean(axis=0), b.mean(axis=0))
    ca, cb = np.cov(a.T), np.cov(b.T)
    scale = np.sqrt(np.outer(np.diag(cb), np.diag(cb)))
    # 1 % -- or, where 6.5 M draws of a heavy-tailed posterior (RosenbrockND(3): kurtosis 10 in x3, chains 300 transitions
    # from their start) do not pin a second moment to 1 %, three standard errors of the difference of the two runs,
    # estimated from 64 groups of chains (measured: difference 1.15 %, its standard error 1.36 %: tools/nuts_mode0_cov_noise.py)
    g = 64
    da = np.stack([np.cov(out[i::g].reshape(-1, tgt.dim).T) for i in range(g)])
    db = np.stack([np.cov(ref[i::g].reshape(-1, tgt.dim).T) for i in range(g)])
    se = (da - db).std(axis=0) / np.sqrt(g)
    assert np.all(np.abs(ca - cb) <= np.maximum(0.01 * scale, 3.0 * se)), (which, np.abs(ca - cb) / scale, se / scale)
    assert np.all(np.abs(ca - cb) <= 0.03 * scale), (which, np.abs(ca - cb) / scale)


def test_rosenbrock3_converged_posterior_equals_the_exact_moments(M):
    """BASELINE's metric target, CONVERGED (round 5): RosenbrockND(3) at 65 536 chains under NUTS in the reference's type split
    (mode 0), target_accept_p 0.98, 1000 warm-up + 1000 draws.  The density factorises -- x1 | x0 and x2 | x1 are Gaussian
    (precisions 202 and 200), x0 has the marginal exp(-(1 - x0)^2 - (100 / 101) (1 - x0^2)^2) -- so its moments are exact by
    one-dimensional quadrature, independent of any sampler: the pooled sample must show them within north_star's 1 %, and
    the conventional split R-hat over the 131 072 half-chains must be under 1.05 (what lets bench.py call its ESS/s one)."""
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.nuts import NUTS

    x0 = np.linspace(-5.0, 6.0, 220001)
    w = np.exp(-(1 - x0) ** 2 - (100.0 / 101.0) * (1 - x0 ** 2) ** 2)
    w /= w.sum()
    mu1, s1 = (100.0 * x0 ** 2 + 1.0) / 101.0, 1.0 / 202.0
    m0 = (w * x0).sum()
    m1 = (w * mu1).sum()
    e1sq = (w * (mu1 ** 2 + s1)).sum()
    e14 = (w * (mu1 ** 4 + 6 * mu1 ** 2 * s1 + 3 * s1 ** 2)).sum()
    mean = np.array([m0, m1, e1sq])
    var = np.array([(w * (x0 - m0) ** 2).sum(), e1sq - m1 ** 2, e14 + 1.0 / 200.0 - e1sq ** 2])
    s = NUTS(M.dist.RosenbrockND(3), M.core.init_with_seed(65536, 3, 42), 0.98, mode=0).set_seed(42)
    t = s._run(1000, 1000, True, "torch")
    rhat, ess = S.split_rhat_mean_ess(t)
    assert float((1.0 / rhat).max()) <= 1.05, rhat
    assert float(ess.min()) > 1e6, ess
    x = t.double().reshape(-1, 3)
    got_mean, got_var = x.mean(dim=0).cpu().numpy(), x.var(dim=0).cpu().numpy()
    np.testing.assert_allclose(got_mean, mean, rtol=0.01)
    np.testing.assert_allclose(got_var, var, rtol=0.01)


def test_rosenbrock3_hmc_converges_as_a_chain_of_handles(M):
    """BASELINE's metric names HMC on RosenbrockND(3): ONE fixed (eps, L) -- all the reference's HMC has, hmc.rs:87-121 -- leaves
    var(x2) at 1.2-1.9x the exact value with split R-hat 1.06-1.75 (profiles/r6m_converged_probe.jsonl).  A sequence of handles,
    each `HMC::new(target, last positions, eps_k, L_k)` + `run(100, 0)` with eps_k ~ U(0.004, 0.016), L_k ~ U{100..400}
    (hmc.run_chain_of_handles: 20 launches discarded, 40 kept), must give the exact moments (quadrature, as in the NUTS test
    above) within north_star's 1 % and a conventional split R-hat <= 1.05 over the 131 072 half-chains.  The band is the one
    that held over four noise / schedule seeds (profiles/r6s_converged_hmc_robustness.jsonl: R-hat <= 1.003, moments within
    0.04 %, ESS 2.1e7-4.8e7); eps up to 0.03 met the 1 % with one seed of four."""
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.hmc import run_chain_of_handles

    x0 = np.linspace(-5.0, 6.0, 220001)
    w = np.exp(-(1 - x0) ** 2 - (100.0 / 101.0) * (1 - x0 ** 2) ** 2)
    w /= w.sum()
    mu1, s1 = (100.0 * x0 ** 2 + 1.0) / 101.0, 1.0 / 202.0
    m0, m1 = (w * x0).sum(), (w * mu1).sum()
    e1sq = (w * (mu1 ** 2 + s1)).sum()
    e14 = (w * (mu1 ** 4 + 6 * mu1 ** 2 * s1 + 3 * s1 ** 2)).sum()
    mean = np.array([m0, m1, e1sq])
    var = np.array([(w * (x0 - m0) ** 2).sum(), e1sq - m1 ** 2, e14 + 1.0 / 200.0 - e1sq ** 2])
    t, info = run_chain_of_handles(M.dist.RosenbrockND(3), M.core.init_with_seed(65536, 3, 42, np.float32), (0.004, 0.016), (100, 400),
                                   100, 20, 40, seed=42)
    assert t.shape == (65536, 4000, 3) and info["launches"] == 60 and 0.85 < info["accept_rate"] < 0.999
    rhat, ess = S.split_rhat_mean_ess(t)
    assert float((1.0 / rhat).max()) <= 1.05, rhat  # measured 1.0012
    assert float(ess.min()) > 5e6, ess               # measured 4.5e7 of 2.6e8 draws
    x = t.double().reshape(-1, 3)
    np.testing.assert_allclose(x.mean(dim=0).cpu().numpy(), mean, rtol=0.01)  # measured: 2e-4
    np.testing.assert_allclose(x.var(dim=0).cpu().numpy(), var, rtol=0.01)    # measured: 2e-4


def test_nuts_depth_cap_matches_oracle_cap(M, O):
    from mini_mcmc_amd.nuts import NUTS

    init = M.core.init_with_seed(64, 3, 1) * 0.5
    for cap in (1, 3):
        s = NUTS(M.dist.StandardNormal(3), init, 0.8, mode=2).set_seed(2).set_max_depth(cap)
        out = s.run(10, 10)
        ref, lf, _ = _oracle_nuts(O, O.standard_normal(3), init, 2, 2, 10, 10, max_depth=cap)
        assert np.array_equal(s.leapfrog_counts().astype(np.int64), lf)
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_nuts_lane_group_mfma_tree_shapes_equal_recursive_oracle(M, O, variant):
    """The MFMA lane-group mapping (mm_nuts_lg.h; variant 1 one launch, 2 compaction by launches, 3 persistent
    scheduler) on the config-5 shaped target against the recursive oracle: identical tree shapes (leapfrog counts) for
    every chain while rounding has not been amplified into a different tree decision -- 15 transitions at condition
    number 1e4 (~2000 leapfrog steps per chain), 30 at 1e2 -- samples to the amplified rounding; over 30 transitions at
    1e4 a few chains' trajectories have separated (the sequential-order CPU twin: 92.5 % still equal), the rest agree."""
    from mini_mcmc_amd.nuts import NUTS

    def pair(dim, cond, C, nc, nd):
        g = M.dist.GaussianND.ill_conditioned(dim, cond, 7)
        init = M.core.init_with_seed(C, dim, 42) * 0.1
        s = NUTS(g, init, 0.8, mode=2).set_seed(42).set_kernel_variant(variant)
        out = s.run(nc, nd)
        ref, lf, eps = _oracle_nuts(O, O.gaussian_nd(g.precision), init, 2, 42, nc, nd)
        return out, s.leapfrog_counts().astype(np.int64), s.adapt_state()["epsilon"], ref, lf, eps

    for dim, cond, C, nc, nd, tol in ((32, 1e4, 40, 4, 6, 1e-9), (32, 1e4, 40, 5, 10, 1e-4), (16, 100.0, 70, 10, 20, 1e-4),
                                      (16, 100.0, 70, 4, 0, 1e-12)):
        out, glf, geps, ref, lf, eps = pair(dim, cond, C, nc, nd)
        name = f"D={dim} variant={variant} run({nc}, {nd})"
        assert np.array_equal(glf, lf), name
        np.testing.assert_allclose(out, ref, rtol=tol, atol=tol, err_msg=name)
        np.testing.assert_allclose(geps, eps, rtol=max(tol, 1e-9), err_msg=name)
    out, glf, geps, ref, lf, eps = pair(32, 1e4, 40, 10, 20)
    same = glf == lf
    assert same.mean() >= 0.85, same.mean()
    np.testing.assert_allclose(geps[same], eps[same], rtol=2e-2)


def test_nuts_config5_full_size_scheduler_equals_single_launch_and_twin(M, O):
    """BASELINE config 5 at its full chain count (65 536 chains, 32-D, condition number 1e4, f64; a shortened
    60 + 40 schedule): the persistent scheduler (variant 3: 8 queue shards, ring laps, chains advancing independently)
    against the single-launch lock-step kernel (variant 1) on EVERY sample, leapfrog count and adaptation state; the
    first 256 chains against the host twin bit for bit; and, on a 10 + 5 schedule (before rounding drift can flip a tree
    decision), the first 64 chains' tree shapes against the recursive oracle."""
    from mini_mcmc_amd.nuts import NUTS

    C, nc, nd = 65536, 40, 60
    g = M.dist.GaussianND.ill_conditioned(32, 1e4, 7)
    init = M.core.init_with_seed(C, 32, 42) * 0.1
    res = {}
    for variant in (3, 1):
        s = NUTS(g, init, 0.8, mode=2).set_seed(42).set_kernel_variant(variant)
        out = s.run(nc, nd)
        res[variant] = (out, s.leapfrog_counts(), s.positions(), s.adapt_state(), s.depth_histogram())
    a, b = res[3], res[1]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(a[3][k], b[3][k]) for k in ("epsilon", "epsilon_bar", "h_bar", "mu"))
    assert np.array_equal(a[4], b[4]) and int(a[4].sum()) == C * (nc + nd - 1)
    assert np.isfinite(a[0]).all()
    ref, pos, ad, nlf = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, 32, [], init[:256], 0.8, nc, nd, seed=42, matrix=g.precision)
    assert np.array_equal(a[0][:256], ref) and np.array_equal(a[1][:256], nlf) and np.array_equal(a[2][:256], pos)
    s = NUTS(g, init, 0.8, mode=2).set_seed(42).set_kernel_variant(3)
    short = s.run(5, 10)
    oref, lf, _ = _oracle_nuts(O, O.gaussian_nd(g.precision), init[:64], 2, 42, 5, 10)
    assert np.array_equal(s.leapfrog_counts()[:64].astype(np.int64), lf)
    np.testing.assert_allclose(short[:64], oref, rtol=1e-4, atol=1e-4)


def test_nuts_config5_at_surveys_schedule_500_plus_500(M, O):
    """BASELINE config 5 at SURVEY 8(d)'s schedule -- 65 536 chains, 500 warm-up + 500 draws, adaptation stopping at n_discard
    (nuts.rs:676-690) -- which the shortened tests above never reach (VERDICT r4 missing #3): the first 48 chains against the
    host twin BIT FOR BIT over all 1000 transitions (samples, final positions, leapfrog counts: every tree decision of
    ~190 000 leapfrog steps per chain), every chain's adapted step size frozen after the warm-up, and -- the target being
    Gaussian -- the pooled posterior against its exact moments: mean 0 and covariance A^-1 (north_star's 1 %, here on the
    33.5 M post-adaptation draws)."""
    from mini_mcmc_amd.nuts import NUTS

    C, nc, nd = 65536, 500, 500
    g = M.dist.GaussianND.ill_conditioned(32, 1e4, 7)
    init = M.core.init_with_seed(C, 32, 42) * 0.1
    s = NUTS(g, init, 0.8, mode=2).set_seed(42).set_max_depth(10)
    t = s._run(nc, nd, True, "torch")
    ad = s.adapt_state()
    assert np.array_equal(ad["epsilon"], ad["epsilon_bar"])  # m > n_discard: epsilon = epsilon_bar (nuts.rs:688)
    assert int(s.depth_histogram().sum()) == C * (nc + nd)
    ref, pos, _, nlf = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, 32, [], init[:48], 0.8, nc, nd, seed=42, matrix=g.precision, progress=True)
    out48 = t[:48].cpu().numpy()
    assert np.array_equal(out48, ref) and np.array_equal(s.leapfrog_counts()[:48], nlf) and np.array_equal(s.positions()[:48], pos)
    x = t.reshape(-1, 32)
    cov = np.linalg.inv(np.asarray(g.precision, dtype=np.float64))
    sd = np.sqrt(np.diag(cov))
    mean = x.mean(dim=0).cpu().numpy()
    assert np.all(np.abs(mean) <= 0.01 * sd), np.abs(mean / sd).max()
    xc = x - x.mean(dim=0, keepdim=True)
    got = (xc.T @ xc / (x.shape[0] - 1)).cpu().numpy()
    assert np.all(np.abs(got - cov) <= 0.01 * np.sqrt(np.outer(np.diag(cov), np.diag(cov)))), \
        (np.abs(got - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))).max()


# ---------------------------------------------------------------- integer-state MH and Gibbs (f2, f4)


@pytest.mark.parametrize("name,params", [("POISSON_REFLECT", [4.0]), ("BINOMIAL_CLAMP", [10, 0.3]), ("POISSON_NONNEG", [4.0])])
def test_discrete_mh_vs_reference_order(M, O, name, params):
    """Integer work: the kernel's states, samples and accept counts equal oracle/discrete.c (metropolis_hastings.rs:
    303-315 over the reference's discrete targets: log-pmf summed on every call, q-terms kept) exactly."""
    from mini_mcmc_amd import discrete as D

    model = {"POISSON_REFLECT": D.PoissonReflect(4.0), "BINOMIAL_CLAMP": D.BinomialClamp(10, 0.3),
             "POISSON_NONNEG": D.PoissonNonneg(4.0)}[name]
    rng = np.random.default_rng(3)
    for C_, nc, nd, off in ((37, 400, 60, 5), (2000, 150, 50, 0)):
        start = rng.integers(0, 9, size=C_).astype(np.int32)
        s = D.DiscreteMetropolisHastings(model, start).seed(11)
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        o = O.DiscreteMH(getattr(O, name), params, start).use_engine_stream(11, off)
        ref = o.run(nc, nd)
        assert np.array_equal(out[:, :, 0], ref), (name, C_)
        assert np.array_equal(s.state().ravel(), o.state()) and np.array_equal(s.accept_counts(), o.accept_counts), (name, C_)


@pytest.mark.parametrize("params", [(-2.0, 1.0, 3.0, 1.5, 0.5), (-42.0, 69.0, 1.0, 2.0, 0.123)])
def test_gibbs_mixture_vs_reference_order(M, O, params):
    """GibbsSampler over MixtureConditional (gibbs.rs:139-203, 231-285) as restated in oracle/gibbs.c: the latent
    indicator exact, x to one rounding of sigma * z (the oracle adds mean + std * z unfused)."""
    from mini_mcmc_amd.gibbs import GibbsSampler, MixtureConditional

    for C_, nc, nd, off in ((33, 300, 40, 3), (1500, 80, 20, 0)):
        init = M.core.init_with_seed(C_, 2, 42)
        s = GibbsSampler(MixtureConditional(*params), init).set_seed(5)
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        o = O.GibbsMixture(params, init).use_engine_stream(5, off)
        ref = o.run(nc, nd)
        assert np.array_equal(out[:, :, 1], ref[:, :, 1])
        np.testing.assert_allclose(out[:, :, 0], ref[:, :, 0], rtol=4e-16, atol=1e-13)
        np.testing.assert_allclose(s.state(), o.state(), rtol=4e-16, atol=1e-13)
