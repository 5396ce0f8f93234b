"""GPU parity tests proper: everything goes through the C ABI (libmmcmc.so) on a real MI355X and is compared with
the CPU checker (oracle/).  Run with `-m gpu`.

  level C (here): GPU == host build of the engine's arithmetic, bit for bit -- samples, states, accept counts,
                  at small sizes, at ragged sizes, across launch partitions and chain offsets, and at
                  BASELINE.json's full size through a checksum of per-chain checksums
  level D (here): GPU vs the reference-ordered restatement (rand-compatible stream): posterior moments within
                  the reference's own tolerances
Levels A (oracle vs the reference's known answers) and B (engine arithmetic vs reference-ordered arithmetic on
identical noise) run on the CPU: tests/test_oracle_pins.py, tests/test_step_parity.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


GAUSS = [0.0, 1.0, 4.0, 2.0, 2.0, 3.0]


# ---------------------------------------------------------------- stream and densities


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [1, 2, 3, 5, 8, 32])
def test_noise_bit_exact(M, O, dtype, dim):
    z, u = M.core.draw_noise(1234567, 10, 77, 300, dim, dtype)
    zr, ur = O.engine_host_noise(1234567, 10, 77, 300, dim, dtype)
    assert np.array_equal(z, zr) and np.array_equal(u, ur)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_logp_grad_bit_exact_and_close_to_reference_order(M, O, dtype):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((8, 8))
    A = A @ A.T + np.eye(8)
    cases = [
        (M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.GAUSSIAN2D, GAUSS, None, O.gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])),
        (M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.DIFFABLE_GAUSSIAN2D, GAUSS, None,
         O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=(dtype == np.float32))),
        (M.dist.IsotropicGaussian(1.5, 4), O.ISOTROPIC_GAUSSIAN, [1.5], None, O.isotropic_gaussian(1.5, 4)),
        (M.dist.Rosenbrock2D(1.0, 100.0), O.ROSENBROCK2D, [1.0, 100.0], None, O.rosenbrock2d(1.0, 100.0)),
        (M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], None, O.rosenbrock_nd(3)),
        (M.dist.RosenbrockND(7), O.ROSENBROCK_ND, [], None, O.rosenbrock_nd(7)),
        (M.dist.StandardNormal(5), O.STANDARD_NORMAL, [], None, O.standard_normal(5)),
        (M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, O.gaussian_nd(A)),
    ]
    for tgt, kind, params, mat, otgt in cases:
        x = (rng.standard_normal((257, tgt.dim)) * 0.8).astype(dtype)
        lp, g = tgt.unnorm_logp_batch(x, dtype, with_grad=True)
        lpr, gr = O.engine_host_logp_grad(kind, tgt.dim, params, x, matrix=mat, dtype=dtype)
        assert np.array_equal(lp, lpr) and np.array_equal(g, gr), type(tgt).__name__
        # against the reference-ordered (unfused) arithmetic: agreement to rounding
        tol = 2e-5 if dtype == np.float32 else 1e-12
        for i in range(0, 257, 37):
            lo, go = otgt.logp_grad(x[i], dtype)
            scale = max(1.0, abs(float(lo)))
            assert abs(float(lp[i]) - float(lo)) <= tol * scale
            np.testing.assert_allclose(g[i], go, rtol=tol * 10, atol=tol * 10 * max(1.0, np.abs(go).max()))


# ---------------------------------------------------------------- samplers, bit exact


def _run_pair(M, O, sampler, tgt, kind, params, init, scale, nc, nd, seed, L=0, mat=None, offset=0, ipl=0, variant=None):
    dtype = init.dtype.type
    if sampler == "mh":
        s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(scale), init).seed(seed)
    else:
        s = M.hmc.HMC(tgt, init, scale, L).set_seed(seed)
    if offset:
        s.set_chain_offset(offset)
    if ipl:
        s.set_iters_per_launch(ipl)
    if variant is not None:
        s.set_kernel_variant(variant)
    out = s.run(nc, nd)
    ref, ref_state, ref_acc = O.engine_host_run(sampler, kind, tgt.dim, params, init, scale, nc, nd, seed=seed,
                                                chain_offset=offset, n_leapfrog=L, matrix=mat, dtype=dtype)
    return s, out, ref, ref_state, ref_acc


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_chains,nc,nd", [(4, 1000, 100), (1, 7, 0), (63, 33, 5), (64, 48, 0), (65, 49, 1), (300, 97, 3)])
def test_mh_gaussian2d_bit_exact(M, O, dtype, n_chains, nc, nd):
    # BASELINE.json config 1 (4 x (1000+100), init_det) and ragged shapes around the 64-lane / tile boundaries
    init = M.core.init_det(n_chains, 2, dtype)
    tgt = M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    s, out, ref, ref_state, ref_acc = _run_pair(M, O, "mh", tgt, O.GAUSSIAN2D, GAUSS, init, 1.0, nc, nd, 42)
    assert out.shape == (n_chains, nc, 2)
    assert np.array_equal(out, ref)
    assert np.array_equal(s.accept_counts, ref_acc)
    assert np.array_equal(s.state(), ref_state)
    assert np.array_equal(out[:, -1, :], ref_state)  # MarkovChain::current_state == last collected row


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_chains,nc,nd", [(4, 400, 50), (130, 65, 7), (64, 32, 0), (64, 31, 2), (200, 1, 0)])
def test_hmc_rosenbrock3d_bit_exact(M, O, dtype, n_chains, nc, nd):
    # BASELINE.json config 3 shape (examples/minimal_hmc.rs: eps 0.032, L 10, 400 + 50) at small chain counts
    init = M.core.init_det(n_chains, 3, dtype)
    s, out, ref, ref_state, ref_acc = _run_pair(M, O, "hmc", M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], init,
                                                0.032, nc, nd, 42, L=10)
    assert np.array_equal(out, ref)
    assert np.array_equal(s.accept_counts, ref_acc)
    assert np.array_equal(s.state(), ref_state)


def test_other_targets_and_dims_bit_exact(M, O):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((32, 32))
    A = A @ A.T / 32 + np.eye(32)
    A16 = A[:16, :16].copy()
    cases = [
        ("hmc", M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.DIFFABLE_GAUSSIAN2D, GAUSS, None, 0.1, 10),
        ("hmc", M.dist.Rosenbrock2D(1.0, 100.0), O.ROSENBROCK2D, [1.0, 100.0], None, 0.01, 50),
        ("hmc", M.dist.RosenbrockND(2), O.ROSENBROCK_ND, [], None, 0.02, 8),
        ("hmc", M.dist.RosenbrockND(8), O.ROSENBROCK_ND, [], None, 0.01, 6),
        ("hmc", M.dist.StandardNormal(5), O.STANDARD_NORMAL, [], None, 0.3, 4),
        ("hmc", M.dist.IsotropicGaussian(2.0, 6), O.ISOTROPIC_GAUSSIAN, [2.0], None, 0.5, 3),
        ("hmc", M.dist.GaussianND(A16), O.GAUSSIAN_ND, [], A16, 0.2, 5),
        ("hmc", M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 0.2, 5),
        ("mh", M.dist.IsotropicGaussian(1.0, 3), O.ISOTROPIC_GAUSSIAN, [1.0], None, 0.7, 0),
        ("mh", M.dist.RosenbrockND(4), O.ROSENBROCK_ND, [], None, 0.1, 0),
        ("mh", M.dist.Rosenbrock2D(1.0, 100.0), O.ROSENBROCK2D, [1.0, 100.0], None, 0.2, 0),
        ("mh", M.dist.StandardNormal(16), O.STANDARD_NORMAL, [], None, 0.4, 0),
        ("mh", M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 0.15, 0),
    ]
    for sampler, tgt, kind, params, mat, scale, L in cases:
        for dtype in (np.float32, np.float64):
            init = M.core.init_with_seed(70, tgt.dim, 7, dtype)
            # HMC on the dense Gaussian at dim 16 / 32 defaults to the lane-group kernels, which have their own twins
            # (test_hmc_lane_group_*): here the one-chain-per-lane kernels, both forms
            variants = (2, 0) if (sampler == "hmc" and kind == O.GAUSSIAN_ND and tgt.dim >= 16) else (None,)
            for variant in variants:
                s, out, ref, ref_state, ref_acc = _run_pair(M, O, sampler, tgt, kind, params, init, scale, 37, 4, 99, L=L,
                                                            mat=mat, variant=variant)
                assert np.array_equal(out, ref), (sampler, type(tgt).__name__, tgt.dim, dtype.__name__, variant)
            name = f"{sampler} {type(tgt).__name__} D={tgt.dim} {dtype.__name__}"
            assert np.array_equal(out, ref), name
            assert np.array_equal(s.accept_counts, ref_acc), name


@pytest.mark.parametrize("nc,nd", [(0, 7), (5, 0), (6, 3), (9, 4), (33, 32), (1, 1)])
def test_runs_without_output_leave_the_same_state(M, O, nc, nd):
    """run(collect=False) takes the branch-free burn-in loop for all nc + nd transitions (odd totals end with a half pair
    whose second transition is masked): final states and accept counts equal the collecting run's and the host build's."""
    init = M.core.init_with_seed(130, 3, 11, np.float32)
    for make in (lambda: M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(5),
                 lambda: M.mh.MetropolisHastings(M.dist.RosenbrockND(3), M.dist.IsotropicGaussian(0.1), init).seed(5)):
        a = make()
        a.run(nc, nd, collect=False)
        b = make()
        out = b.run(nc, nd)
        assert np.array_equal(a.state(), b.state()) and np.array_equal(a.accept_counts, b.accept_counts)
        if nc:
            assert np.array_equal(out[:, -1, :], b.state())
    ref, ref_state, ref_acc = O.engine_host_run("hmc", O.ROSENBROCK_ND, 3, [], init, 0.032, nc, nd, seed=5, n_leapfrog=10)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(5)
    h.run(nc, nd, collect=False)
    assert np.array_equal(h.state(), ref_state) and np.array_equal(h.accept_counts, ref_acc)


def test_results_independent_of_launch_partition_and_sharding(M, O):
    # the stream is keyed by (seed, GLOBAL chain id, iteration): splitting the run over launches, continuing a
    # handle, or sharding chains over handles (= GPUs) must not change a single bit
    init = M.core.init_with_seed(192, 3, 42, np.float32)
    whole = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).run(60, 9)
    split = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).set_iters_per_launch(13).run(60, 9)
    assert np.array_equal(whole, split)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1)
    a = h.run(25, 9)
    b = h.run(35, 0)  # second run() continues the chains, like the Rust struct
    assert np.array_equal(np.concatenate([a, b], axis=1), whole)
    lo = M.hmc.HMC(M.dist.RosenbrockND(3), init[:100], 0.032, 10).set_seed(1).run(60, 9)
    hi = M.hmc.HMC(M.dist.RosenbrockND(3), init[100:], 0.032, 10).set_seed(1).set_chain_offset(100).run(60, 9)
    assert np.array_equal(np.concatenate([lo, hi], axis=0), whole)
    # step(): hmc.rs:304-377
    h2 = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1)
    for _ in range(9):
        h2.step()
    h2.step()
    assert np.array_equal(h2.positions, whole[:, 0, :])


def test_kernel_variants_agree_bit_for_bit(M, O):
    # variant 2 (noise of two iterations packed + software-pipelined, the default) and variant 0 (plain) compute the same
    # pure function of (seed, chain, iteration): every output must be identical, including ragged tails
    rng = np.random.default_rng(9)
    A = rng.standard_normal((16, 16))
    A = A @ A.T / 16 + np.eye(16)
    cases = [
        ("hmc", M.dist.RosenbrockND(3), 0.032, 10, 333, 101, 7),
        ("hmc", M.dist.RosenbrockND(3), 0.032, 10, 64, 3, 0),
        ("hmc", M.dist.RosenbrockND(3), 0.05, 7, 100, 40, 3),
        ("hmc", M.dist.GaussianND(A), 0.2, 4, 70, 19, 2),
        ("mh", M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 1.0, 0, 333, 205, 9),
        ("mh", M.dist.StandardNormal(7), 0.5, 0, 65, 50, 1),
    ]
    for dtype in (np.float32, np.float64):
        for sampler, tgt, scale, L, C, nc, nd in cases:
            init = M.core.init_with_seed(C, tgt.dim, 21, dtype)
            outs = []
            for variant in (0, 2) + ((5,) if tgt.dim <= 8 else ()):  # 5: noise waves + transition waves (dim <= 8)
                if sampler == "mh":
                    s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(scale), init).seed(77)
                else:
                    s = M.hmc.HMC(tgt, init, scale, L).set_seed(77)
                s.set_kernel_variant(variant)
                outs.append((s.run(nc, nd), s.accept_counts.copy(), s.state()))
            for other in outs[1:]:
                for a, b in zip(outs[0], other):
                    assert np.array_equal(a, b), (sampler, type(tgt).__name__, dtype.__name__)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_split_role_kernel_bit_exact_vs_host_build(M, O, dtype):
    """Kernel variant 5 (mm_split_kernels.h: 512-thread workgroups, four transition waves fed by four noise waves through
    an LDS ring, two waves per SIMD) against the host build: every target it is instantiated for (dim <= 8), ragged
    chain counts around the 64 / 256 boundaries, collected counts around the ring-half and tile sizes, odd burn-in,
    runs without output, continued handles, chain offsets beyond 2^32, split launches."""
    rng = np.random.default_rng(2)
    A = rng.standard_normal((8, 8))
    A = A @ A.T / 8 + np.eye(8)
    cases = [
        ("hmc", M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], None, 0.032, 10),
        ("hmc", M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], None, 0.05, 7),
        ("hmc", M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.DIFFABLE_GAUSSIAN2D, GAUSS, None, 0.1, 10),
        ("hmc", M.dist.Rosenbrock2D(1.0, 100.0), O.ROSENBROCK2D, [1.0, 100.0], None, 0.01, 10),
        ("hmc", M.dist.StandardNormal(1), O.STANDARD_NORMAL, [], None, 0.3, 4),
        ("hmc", M.dist.StandardNormal(5), O.STANDARD_NORMAL, [], None, 0.3, 4),
        ("hmc", M.dist.IsotropicGaussian(2.0, 7), O.ISOTROPIC_GAUSSIAN, [2.0], None, 0.5, 3),
        ("hmc", M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 0.2, 5),
        ("mh", M.dist.Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.GAUSSIAN2D, GAUSS, None, 1.0, 0),
        ("mh", M.dist.RosenbrockND(4), O.ROSENBROCK_ND, [], None, 0.1, 0),
        ("mh", M.dist.StandardNormal(6), O.STANDARD_NORMAL, [], None, 0.4, 0),
        ("mh", M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 0.15, 0),
    ]
    shapes = [(300, 37, 4, 0), (1, 1, 0, 0), (64, 16, 1, 0), (65, 17, 8, 1 << 33), (257, 8, 0, 0), (256, 49, 3, 5), (70, 0, 9, 0)]
    for ci, (sampler, tgt, kind, params, mat, scale, L) in enumerate(cases):
        for C, nc, nd, off in (shapes if ci in (0, 8) else shapes[ci % len(shapes):][:2]):
            init = M.core.init_with_seed(C, tgt.dim, 7, dtype)
            s, out, ref, ref_state, ref_acc = _run_pair(M, O, sampler, tgt, kind, params, init, scale, nc, nd, 99, L=L,
                                                        mat=mat, offset=off, variant=5)
            name = f"{sampler} {type(tgt).__name__} D={tgt.dim} {dtype.__name__} C={C} run({nc}, {nd})"
            assert np.array_equal(out, ref), name
            assert np.array_equal(s.accept_counts, ref_acc) and np.array_equal(s.state(), ref_state), name
    # continued handle, launch partition, run without output
    init = M.core.init_with_seed(333, 3, 42, dtype)
    whole = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).set_kernel_variant(2).run(60, 9)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).set_kernel_variant(5)
    a = h.run(25, 9)
    b = h.run(35, 0)
    assert np.array_equal(np.concatenate([a, b], axis=1), whole)
    split = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).set_kernel_variant(5).set_iters_per_launch(13).run(60, 9)
    assert np.array_equal(split, whole)
    q = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(1).set_kernel_variant(5)
    q.run(60, 9, collect=False)
    assert np.array_equal(q.state(), whole[:, -1, :])
    with pytest.raises(Exception):  # no instance above dim 8
        M.hmc.HMC(M.dist.StandardNormal(16), M.core.init_with_seed(8, 16, 1, dtype), 0.1, 5).set_kernel_variant(5)


def test_torch_device_output_matches_host_output(M, O):
    import torch

    init = M.core.init_with_seed(500, 3, 3, np.float32)
    a = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(5).run(70, 5)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(5)
    t = h.run(70, 5, to="torch")
    torch.cuda.synchronize()
    assert t.is_cuda and tuple(t.shape) == (500, 70, 3)
    assert np.array_equal(t.cpu().numpy(), a)
    tm = h.timing()
    assert tm["kernel_ms"] > 0 and tm["n_launches"] == 1 and tm["out_bytes"] == 500 * 70 * 3 * 4


# ---------------------------------------------------------------- full BASELINE sizes: checksum of checksums


def _chain_checksums(sample_u32):
    # per-chain xor-rotate checksum over the raw bits, then a checksum of the per-chain checksums
    c = np.zeros(sample_u32.shape[0], dtype=np.uint64)
    flat = sample_u32.reshape(sample_u32.shape[0], -1).astype(np.uint64)
    for k in range(flat.shape[1]):
        c = ((c << np.uint64(7)) | (c >> np.uint64(57))) ^ (flat[:, k] * np.uint64(0x9E3779B97F4A7C15))
    return c


def test_config3_full_size_checksum(M, O):
    # BASELINE.json config 3: 3-D Rosenbrock HMC, 65 536 chains, L = 10, f32, 400 + 50 (examples/minimal_hmc.rs)
    C, nc, nd = 65536, 400, 50
    init = M.core.init_with_seed(C, 3, 42, np.float32)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    out = h.run(nc, nd)
    ref, _, ref_acc = O.engine_host_run("hmc", O.ROSENBROCK_ND, 3, [], init, 0.032, nc, nd, seed=42, n_leapfrog=10)
    assert np.array_equal(h.accept_counts, ref_acc)  # bit-exact accept counts for all 65 536 chains
    a, b = _chain_checksums(out.view(np.uint32)), _chain_checksums(ref.view(np.uint32))
    assert np.array_equal(a, b)
    assert np.bitwise_xor.reduce(a) == np.bitwise_xor.reduce(b)
    assert np.all(np.isfinite(out))


def test_config2_full_size_checksum(M, O):
    # BASELINE.json config 2: Gaussian2D MH, 65 536 chains, f32, 1000 + 100 (examples/minimal_mh.rs shape)
    C, nc, nd = 65536, 1000, 100
    init = M.core.init_with_seed(C, 2, 42, np.float32)
    tgt = M.dist.Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]])
    mh = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(1.0), init).seed(42)
    out = mh.run(nc, nd)
    ref, _, ref_acc = O.engine_host_run("mh", O.GAUSSIAN2D, 2, [0.0, 0.0, 1.0, 0.0, 0.0, 1.0], init, 1.0, nc, nd, seed=42)
    assert np.array_equal(mh.accept_counts, ref_acc)
    a, b = _chain_checksums(out.view(np.uint32)), _chain_checksums(ref.view(np.uint32))
    assert np.array_equal(a, b)
    # posterior moments within 1 % of the truth (north_star), far inside metropolis_hastings.rs:379-380
    flat = out.reshape(-1, 2).astype(np.float64)
    assert np.all(np.abs(flat.mean(axis=0)) < 0.01)
    assert np.all(np.abs(np.cov(flat.T) - np.eye(2)) < 0.01)


# ---------------------------------------------------------------- GPU vs reference-ordered restatement (statistical)


def test_mh_moments_vs_reference_restatement(M, O, kats):
    k = kats["mh_statistical"]
    C, nc, nd = 4096, 500, 500
    init = M.core.init_with_seed(C, 2, 42, np.float64)
    tgt = M.dist.Gaussian2D(k["mean"], k["cov"])
    g = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(1.0), init).seed(42).run(nc, nd)
    o = O.MetropolisHastings(O.gaussian2d(k["mean"], k["cov"]), 1.0, init, np.float64).use_engine_stream(42).run(nc, nd)
    # same noise, reference-ordered arithmetic (unfused, q-terms kept): f64 trajectories agree to rounding
    np.testing.assert_allclose(g, o, rtol=1e-9, atol=1e-9)
    fg = g.reshape(-1, 2)
    assert np.all(np.abs(fg.mean(axis=0) - k["mean"]) < 0.03)
    assert np.all(np.abs(np.cov(fg.T) - np.array(k["cov"])) < 0.06)
    # and against the reference's own stream (xoshiro + ziggurat, quirks Q1/Q2).  Q1 makes every chain of a sampler
    # share its proposal noise, so extra chains do not average the error away: use few chains and a long run, and
    # hold both sides to the reference's own tolerance around the truth (metropolis_hastings.rs:379-380)
    r = O.MetropolisHastings(O.gaussian2d(k["mean"], k["cov"]), 1.0, init[:8], np.float64, proposal_seed=42).seed(42).run(20000, 500)
    fr = r.reshape(-1, 2)
    assert np.all(np.abs(fr.mean(axis=0) - k["mean"]) < k["mean_atol"])
    assert np.all(np.abs(np.cov(fr.T) - np.array(k["cov"])) < k["cov_atol"])
    assert np.all(np.abs(fr.mean(axis=0) - fg.mean(axis=0)) < k["mean_atol"])
    assert np.all(np.abs(np.cov(fr.T) - np.cov(fg.T)) < k["cov_atol"])


def test_hmc_gaussian_ess_band_on_gpu(M, O, kats):
    # hmc.rs:764-786: mean ESS of 3 chains x 1000 (after 500) in [135,185] / [141,191]; many 3-chain groups at once
    k = kats["hmc_ess_bands"]
    groups = 200
    init = M.core.init_with_seed(3 * groups, 2, 11, np.float32)
    tgt = M.dist.DiffableGaussian2D(k["mean"], k["cov"])
    s = M.hmc.HMC(tgt, init, k["step_size"], k["n_leapfrog"]).set_seed(3).run(k["n_collect"], k["n_discard"])
    ess = np.array([O.split_rhat_mean_ess(s[3 * i:3 * i + 3])[1] for i in range(groups)])
    m = ess.mean(axis=0)
    assert k["ess1_band"][0] <= m[0] <= k["ess1_band"][1], m
    assert k["ess2_band"][0] <= m[1] <= k["ess2_band"][1], m
    flat = s.reshape(-1, 2).astype(np.float64)
    assert np.all(np.abs(flat.mean(axis=0) - k["mean"]) < 0.05)
    assert np.all(np.abs(np.cov(flat.T) - np.array(k["cov"])) < 0.1)


def test_mh_gaussian_ess_band_on_gpu(M, O, kats):
    # metropolis_hastings.rs:417-513 (the reference's ignored benchmark): 3 chains x 1000 (after 500), sigma_prop 1, mean
    # ESS of the first coordinate in [65, 125]; many 3-chain groups at once, started as the reference starts them (N(0, 1))
    k, band = kats["mh_statistical"], kats["mh_ess_bands"]["ess1_band"]
    groups = 200
    init = M.core.init_with_seed(3 * groups, 2, 17, np.float64)
    tgt = M.dist.Gaussian2D(k["mean"], k["cov"])
    s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(k["proposal_std"]), init).seed(5).run(1000, 500)
    ess = np.array([O.split_rhat_mean_ess(s[3 * i:3 * i + 3].astype(np.float32))[1] for i in range(groups)])
    assert band[0] <= ess[:, 0].mean() <= band[1], ess.mean(axis=0)


def test_mh_single_chain_from_a_far_start_and_against_a_wrong_target_on_gpu(M, kats):
    # tests/metrohast_2d_gaussian_test.rs:16-92: ONE chain from (10, 12), 10 000 + 2 500, mean and covariance within 0.5 of
    # the target's -- and the same run against the identity covariance must NOT match (max |cov difference| > 1.0)
    k = kats["mh_single_chain_statistical"]

    def moments(cov):
        tgt = M.dist.Gaussian2D(k["mean"], cov)
        init = np.array([k["init"]], dtype=np.float64)
        s = M.mh.MetropolisHastings(tgt, M.dist.IsotropicGaussian(k["proposal_std"]), init).seed(k["seed"])
        out = s.run(k["n_collect"], k["n_discard"])
        assert out.shape == (1, k["n_collect"], 2)
        flat = out.reshape(-1, 2)
        return flat.mean(axis=0), np.cov(flat.T)

    mean, cov = moments(k["cov"])
    assert np.all(np.abs(mean - k["mean"]) < k["mean_atol"]), mean
    assert np.abs(cov - np.array(k["cov"])).max() < k["cov_atol"], cov
    _, cov_false = moments(k["false_cov"])
    assert np.abs(cov_false - np.array(k["cov"])).max() > k["false_target_cov_diff_gt"], cov_false


# ---------------------------------------------------------------- diagnostics on the GPU (stats.rs:416-546)


def _ar1(rng, c, n, p, phi=0.6):
    x = np.zeros((c, n, p), dtype=np.float32)
    e = rng.standard_normal((c, n, p)).astype(np.float32)
    for t in range(1, n):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    return x


@pytest.mark.parametrize("c,n,p", [(4, 60, 2), (3, 200, 3), (7, 201, 1), (16, 400, 3), (2, 1000, 2), (5, 2, 2), (1, 50, 4),
                                   (64, 100, 32), (2, 6000, 3),  # the last: too long for the MFMA kernel's LDS layout -> direct kernel
                                   # one parameter per wave (mm_half_chain_tile1_kernel) with enough chains for every path:
                                   # six tile parts, eight column loads (m = 257, two parts), D = 1, ragged m, one slot
                                   (300, 1000, 3), (200, 514, 2), (100, 402, 1), (50, 331, 5), (700, 64, 3)])
def test_split_rhat_mean_ess_vs_oracle(M, O, c, n, p):
    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(c * 1000 + n)
    x = _ar1(rng, c, n, p)
    x[:, :, 0] += np.arange(c)[:, None] * 0.25
    r0, e0 = O.split_rhat_mean_ess(x)
    r1, e1 = S.split_rhat_mean_ess(x)
    ok = np.isfinite(r0)
    np.testing.assert_allclose(r1[ok], r0[ok], rtol=1e-4)
    ok = np.isfinite(e0)
    np.testing.assert_allclose(e1[ok], e0[ok], rtol=5e-3)  # n/2 > 100: the oracle (like the reference) uses an f32 FFT
    # f64 input is cast to f32 first, like RunStats::from (stats.rs:365)
    r2, e2 = S.split_rhat_mean_ess(x.astype(np.float64))
    assert np.array_equal(r1, r2) and np.array_equal(e1, e2)


@pytest.mark.parametrize("c,n,p", [(2000, 400, 3), (300, 1000, 3), (500, 200, 2), (200, 2048, 1), (64, 260, 5),
                                   (33, 203, 7), (10, 1500, 2)])
def test_stats_kernels_agree(M, O, c, n, p):
    """The lag-sum kernels -- power spectrum ("fft": the default above 100 draws per half-chain, like stats.rs:549), one
    parameter per wave ("tile1": the default below), all parameters in one wave ("tile"), matrix cores ("mfma"), plain
    sums ("direct") -- on one sample: R-hat / ESS equal to 1e-5 / 1e-4, each bitwise reproducible from call to call."""
    from mini_mcmc_amd import stats as S

    x = _ar1(np.random.default_rng(n + p), c, n, p)
    x[:, :, -1] += 3.0
    res = {}
    try:
        for k in ("auto", "fft", "tile1", "tile", "mfma", "direct"):
            S.set_kernel(k)
            a = S.split_rhat_mean_ess(x)
            b = S.split_rhat_mean_ess(x)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), k
            res[k] = a
    finally:
        S.set_kernel("auto")
    for k in ("fft", "tile1", "tile", "mfma", "direct"):
        np.testing.assert_allclose(res[k][0], res["auto"][0], rtol=1e-5, err_msg=k)
        np.testing.assert_allclose(res[k][1], res["auto"][1], rtol=1e-4, err_msg=k)
    r0, e0 = O.split_rhat_mean_ess(x)
    np.testing.assert_allclose(res["auto"][0], r0, rtol=1e-4)
    np.testing.assert_allclose(res["auto"][1], e0, rtol=5e-3)


def test_stats_partials_fft_vs_direct_lag_sums(M, O):
    """The power-spectrum kernel's lag sums themselves (not only the R-hat / ESS they feed): `stats_partials` of one sample
    under "fft" and under "direct" -- means and centred sums of squares per half-chain to float rounding, every lag sum to
    2e-6 of the lag-0 sum (the f32 transform's error is relative to the spectrum's largest bin)."""
    import torch

    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(5)
    # the last rows: half-chain lengths around the transform sizes 512 / 1024 / 2048 (m = 101 is the shortest, 1024 the longest)
    for c, n, p in [(257, 400, 3), (100, 1000, 3), (31, 333, 4), (12, 2047, 1), (50, 250, 6), (8, 202, 2), (9, 512, 3),
                    (9, 514, 1), (5, 1024, 2), (5, 1026, 3), (3, 2046, 1), (3, 2048, 2), (20, 400, 3)]:
        x = _ar1(rng, c, n, p)
        if (c, n) == (20, 400):
            x = x.astype(np.float64)  # f64 samples take the same kernel (converted on load)
        x[:, :, 0] += 100.0  # a mean far from zero: the centring happens before the transform
        t = torch.from_numpy(x).cuda()
        out = {}
        try:
            for k in ("fft", "direct"):
                S.set_kernel(k)
                out[k] = [v.cpu().numpy() for v in S.stats_partials(t)]
        finally:
            S.set_kernel("auto")
        np.testing.assert_allclose(out["fft"][0], out["direct"][0], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(out["fft"][1], out["direct"][1], rtol=2e-5)
        scale = out["direct"][2][0]  # lag 0, per parameter
        assert np.max(np.abs(out["fft"][2] - out["direct"][2]) / scale) < 2e-6, (c, n, p)
        # and against f64 sums on the host
        m = n // 2
        halves = np.concatenate([x[:, :m], x[:, n - m:]], axis=0).astype(np.float64)
        y = halves - halves.mean(axis=1, keepdims=True)
        ref = np.stack([(y[:, :m - k] * y[:, k:]).sum(axis=(0, 1)) for k in range(m)])
        assert np.max(np.abs(out["fft"][2] - ref) / ref[0]) < 2e-6, (c, n, p)


@pytest.mark.parametrize("n", [400, 1000])
def test_split_rhat_mean_ess_full_size_vs_oracle(M, O, n):
    """BASELINE's diagnostics shape [65 536, n, 3] against oracle/stats.c (stats.rs:416-546 with its FFT branch) -- not
    kernel against kernel: an HMC sample of config 3 from the engine itself, R-hat / ESS to 1e-4 / 2e-3."""
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import RosenbrockND
    from mini_mcmc_amd.hmc import HMC

    h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
    t = h.run(n, 50, to="torch", accept_counts=False)
    r1, e1 = S.split_rhat_mean_ess(t)
    x = t.cpu().numpy()
    r0, e0 = O.split_rhat_mean_ess(x)
    np.testing.assert_allclose(r1, r0, rtol=1e-4)
    np.testing.assert_allclose(e1, e0, rtol=2e-3)


def test_stats_kats_on_gpu(M, O, kats):
    from mini_mcmc_amd import stats as S

    # stats.rs:810-834 ess_1
    k = kats["ess_1"]
    r = O.SmallRng(k["seed"])
    data = np.array([[r.f32() for _ in range(k["n"])] for _ in range(k["chains"])], dtype=np.float32)[:, :, None]
    st = S.run_stats(data)
    assert st.ess.min > k["ess_min_gt"] and st.rhat.max < k["rhat_max_lt"]
    # autocovariance known answers (stats.rs:777-808) through the partial statistics: one chain of 8 = two halves of 4
    import torch

    for ka in kats["autocov"]:
        d = np.array(ka["data"], dtype=np.float32)
        x = np.concatenate([d, d], axis=0)[None]  # [1, 8, p]: both halves equal the 4-point series
        means, ssq, acov = S.stats_partials(torch.from_numpy(x).cuda())
        got = acov.cpu().numpy() / 4.0 / 2.0  # /n, mean over the two half-chains
        np.testing.assert_allclose(got, np.array(ka["expected"], dtype=np.float32), atol=ka["atol"])
        np.testing.assert_allclose(means.cpu().numpy()[0, 0], d.mean(axis=0), rtol=1e-6)


def test_distributed_diagnostics_over_rccl_one_rank(M, O):
    """stats.split_rhat_mean_ess_distributed through the real backend ("nccl" = RCCL) on a one-rank group in a child
    process: device statistics -> two all-reduces of CUDA tensors -> host finish, against the single-GPU entry point.
    (world_size 2 runs on CPU tensors with gloo in tests/test_distributed_stats.py; 8-GPU runs are the driver's.)"""
    import subprocess
    import sys

    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MM_ROOT"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29583")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mini_mcmc_amd import stats as S
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC
t = HMC(RosenbrockND(3), init_with_seed(1024, 3, 42, np.float32), 0.032, 10).set_seed(42).run(300, 50, to="torch")
r1, e1 = S.split_rhat_mean_ess(t)
r2, e2 = S.split_rhat_mean_ess_distributed(t)
np.testing.assert_allclose(r2, r1, rtol=2e-6); np.testing.assert_allclose(e2, e1, rtol=1e-4)
m, q, a = S.stats_partials(t)
g = S.gather_partials(m, q, a)
r3, e3 = S.stats_finish(*g)
np.testing.assert_allclose(r3, r1, rtol=2e-6); np.testing.assert_allclose(e3, e1, rtol=1e-4)
dist.destroy_process_group()
print("ok")
"""
    env = dict(os.environ, MM_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout.split(), out.stdout + out.stderr  # RCCL prints its banner too


def test_stats_on_device_sample_and_sharded_partials(M, O):
    import torch
    from mini_mcmc_amd import stats as S

    init = M.core.init_with_seed(512, 3, 42, np.float32)
    h = M.hmc.HMC(M.dist.RosenbrockND(3), init, 0.032, 10).set_seed(42)
    t = h.run(300, 50, to="torch")
    torch.cuda.synchronize()
    r_dev, e_dev = S.split_rhat_mean_ess(t)
    r_host, e_host = S.split_rhat_mean_ess(t.cpu().numpy())
    assert np.array_equal(r_dev, r_host) and np.array_equal(e_dev, e_host)
    r0, e0 = O.split_rhat_mean_ess(t.cpu().numpy())
    np.testing.assert_allclose(r_dev, r0, rtol=1e-4)
    np.testing.assert_allclose(e_dev, e0, rtol=5e-3)
    # sharding the chains over "ranks" and merging sufficient statistics gives the single-GPU answer
    parts = [S.stats_partials(t[i * 128:(i + 1) * 128]) for i in range(4)]
    means = torch.cat([p[0] for p in parts], dim=1).reshape(-1, 3).cpu().numpy()  # [2, C, D] -> splitcat order
    ssq = torch.cat([p[1] for p in parts], dim=1).reshape(-1, 3).cpu().numpy()
    acov = sum(p[2] for p in parts).cpu().numpy()
    r_sh, e_sh = S.stats_finish(means, ssq, acov)
    np.testing.assert_allclose(r_sh, r_dev, rtol=1e-6)
    np.testing.assert_allclose(e_sh, e_dev, rtol=1e-4)


# ---------------------------------------------------------------- NUTS on the GPU (nuts.rs)


def _nuts_pair(M, O, tgt, kind, params, init, mode, nc, nd, seed, progress=False, mat=None, max_depth=10, offset=0):
    from mini_mcmc_amd.nuts import NUTS

    s = NUTS(tgt, init, 0.8, mode=mode).set_seed(seed).set_max_depth(max_depth)
    if offset:
        s.set_chain_offset(offset)
    out = s._run(nc, nd, progress, "numpy")
    ref, pos, ad, nlf = O.engine_host_nuts_run(mode, kind, tgt.dim, params, init, 0.8, nc, nd, seed=seed, matrix=mat,
                                               progress=progress, max_depth=max_depth, chain_offset=offset)
    return s, out, ref, pos, ad, nlf


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_nuts_bit_exact_vs_host_build(M, O, mode):
    rng = np.random.default_rng(4)
    A = rng.standard_normal((8, 8))
    A = A @ A.T / 8 + np.eye(8)
    cases = [
        (M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.DIFFABLE_GAUSSIAN2D, GAUSS, None, 130, 21, 9, False),
        (M.dist.DiffableGaussian2D([1.0, 2.0], [[1.0, 2.0], [2.0, 5.0]]), O.DIFFABLE_GAUSSIAN2D, [1.0, 2.0, 1.0, 2.0, 2.0, 5.0], None, 64, 5, 5, False),
        (M.dist.Rosenbrock2D(1.0, 100.0), O.ROSENBROCK2D, [1.0, 100.0], None, 70, 12, 8, True),
        (M.dist.StandardNormal(3), O.STANDARD_NORMAL, [], None, 65, 10, 0, False),
        (M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], None, 64, 9, 6, True),
        (M.dist.GaussianND(A), O.GAUSSIAN_ND, [], A, 40, 8, 8, False),
    ]
    for tgt, kind, params, mat, C, nc, nd, progress in cases:
        init = M.core.init_with_seed(C, tgt.dim, 31) * 0.8
        s, out, ref, pos, ad, nlf = _nuts_pair(M, O, tgt, kind, params, init, mode, nc, nd, 77, progress, mat)
        name = f"{type(tgt).__name__} D={tgt.dim} mode={mode}"
        assert np.array_equal(out, ref), name
        assert np.array_equal(s.positions(), pos), name
        assert np.array_equal(s.leapfrog_counts(), nlf), name  # identical tree shapes
        a = s.adapt_state()
        assert np.array_equal(a["epsilon"], ad[:, 0]) and np.array_equal(a["epsilon_bar"], ad[:, 1]), name
        assert np.array_equal(a["h_bar"], ad[:, 2]) and np.array_equal(a["mu"], ad[:, 3]), name
        assert s.depth_histogram().sum() == C * (nc + nd - (0 if progress else 1))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_nuts_asynchronous_lanes_equal_synchronous_lanes(M, O, mode):
    """Kernel variants 4 (every lane advances through ITS transitions, one leaf per tick) and 5 (the same with the leaves
    in pairs, level 0 of the stack in registers, uniforms in an LDS ring; the default for dim <= 8)
    against variant 0 (the wave waits for its deepest tree) and against the host build: samples, positions, adaptation
    state, leapfrog counts and the depth histogram are identical, with ragged waves, chain offsets, run / run_progress,
    a continued handle and a depth cap that bites."""
    from mini_mcmc_amd.nuts import NUTS

    cases = [
        (M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], 200, 30, 25, True, 0, 10),
        (M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], 64, 1, 0, False, 0, 10),   # row 0 only: no transition at all
        (M.dist.RosenbrockND(3), O.ROSENBROCK_ND, [], 70, 2, 0, False, 0, 10),   # initial row + one transition
        (M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), O.DIFFABLE_GAUSSIAN2D, GAUSS, 129, 40, 11, False, 1 << 35, 10),
        (M.dist.StandardNormal(8), O.STANDARD_NORMAL, [], 65, 12, 7, True, 5, 3),  # max_depth 3 cuts trees short
    ]
    for tgt, kind, params, C, nc, nd, progress, off, max_depth in cases:
        init = M.core.init_with_seed(C, tgt.dim, 17) * 0.7
        res = {}
        for variant in (5, 4, 0):
            s = NUTS(tgt, init, 0.8, mode=mode).set_seed(23).set_max_depth(max_depth)
            assert s.kernel_variant == 5  # the default: asynchronous lanes, leaves in pairs
            s.set_kernel_variant(variant)
            if off:
                s.set_chain_offset(off)
            out = s._run(nc, nd, progress, "numpy")
            out2 = s._run(5, 0, True, "numpy")  # continue the handle
            res[variant] = (out, out2, s.positions(), s.leapfrog_counts(), s.depth_histogram(), s.adapt_state())
        name = f"{type(tgt).__name__} D={tgt.dim} mode={mode} C={C}"
        for v in (4, 5):
            for i in range(5):
                assert np.array_equal(res[v][i], res[0][i]), (name, v, i)
            for key in ("epsilon", "epsilon_bar", "h_bar", "mu"):
                assert np.array_equal(res[v][5][key], res[0][5][key]), (name, v, key)
        ref, pos, ad, nlf = O.engine_host_nuts_run(mode, kind, tgt.dim, params, init, 0.8, nc, nd, seed=23,
                                                   progress=progress, max_depth=max_depth, chain_offset=off)
        assert np.array_equal(res[4][0], ref), name
    with pytest.raises(Exception):  # no asynchronous-lane instance above dim 8
        NUTS(M.dist.GaussianND.ill_conditioned(16, 10.0, 1), M.core.init_with_seed(8, 16, 1), 0.8, mode=1).set_kernel_variant(4)


def test_hmc_lane_group_mfma_f32_bit_exact_vs_host_twin(M, O):
    """The f32 lane-group kernel (mm_hmc_lg.h, second half: v_mfma_f32_16x16x4_f32, lane (c, q) owning the coordinates
    16 (s >> 2) + 4 q + (s & 3)) against its host twin (blocked reduction order, columns of A x in k-step order), bit for
    bit: samples, states, accept counts, a continued handle, ragged chain counts, a chain offset beyond 2^32; and
    against the one-chain-per-lane kernel to rounding."""
    for dim, cond, C, nc, nd, L, eps, off in ((32, 100.0, 70, 40, 9, 10, 0.05, 0), (32, 1e3, 33, 17, 0, 7, 0.02, 1 << 33),
                                              (16, 50.0, 100, 25, 5, 10, 0.1, 0), (16, 10.0, 16, 3, 2, 1, 0.2, 5)):
        g = M.dist.GaussianND.ill_conditioned(dim, cond, 7)
        init = M.core.init_with_seed(C, dim, 3, np.float32) * np.float32(0.3)
        s = M.hmc.HMC(g, init, eps, L).set_seed(19)
        assert s.kernel_variant == 3  # the default where it exists
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        ref, st, acc = O.engine_host_hmc_grouped_run_f32(g.precision, init, eps, L, nc, nd, seed=19, chain_offset=off)
        name = f"f32 D={dim} cond={cond} C={C}"
        assert out.dtype == np.float32 and np.array_equal(out, ref) and np.array_equal(s.state(), st), name
        assert np.array_equal(s.accept_counts, acc), name
        out2 = s.run(6, 1)  # continues the chains and the stream
        ref2, _, acc2 = O.engine_host_hmc_grouped_run_f32(g.precision, st, eps, L, 6, 1, seed=19, chain_offset=off, iter0=nc + nd)
        assert np.array_equal(out2, ref2) and np.array_equal(s.accept_counts, acc2), name
        a3 = M.hmc.HMC(g, init, eps, L).set_seed(19).set_chain_offset(off).run(2, 0)
        a2 = M.hmc.HMC(g, init, eps, L).set_seed(19).set_chain_offset(off).set_kernel_variant(2).run(2, 0)
        np.testing.assert_allclose(a3, a2, rtol=2e-3, atol=2e-4)  # same transitions, other summation order, f32
    # posterior: whitened draws have unit covariance (32-D, cond 100, 4096 chains)
    g = M.dist.GaussianND.ill_conditioned(32, 100.0, 5)
    s = M.hmc.HMC(g, M.core.init_with_seed(4096, 32, 4, np.float32) * np.float32(0.1), 0.08, 10).set_seed(6)
    smp = s.run(50, 150).reshape(-1, 32).astype(np.float64)
    w = smp @ np.linalg.cholesky(g.precision)
    assert np.abs(w.mean(axis=0)).max() < 0.03 and np.abs(np.cov(w.T) - np.eye(32)).max() < 0.05


def test_nuts_asynchronous_lanes_at_full_size(M, O):
    """65 536 chains (1024 full waves, every lane on its own schedule): variant 4 and variant 0 agree on every sample,
    every chain's leapfrog count and the depth histogram; the histogram counts every transition once."""
    from mini_mcmc_amd.nuts import NUTS

    C, nc, nd = 65536, 12, 18
    init = M.core.init_with_seed(C, 3, 5) * 0.5
    res = {}
    for variant in (4, 0):
        s = NUTS(M.dist.RosenbrockND(3), init, 0.8, mode=1).set_seed(9).set_kernel_variant(variant)
        out = s._run(nc, nd, True, "numpy")
        res[variant] = (out, s.leapfrog_counts(), s.depth_histogram())
    assert np.array_equal(res[4][0], res[0][0])
    assert np.array_equal(res[4][1], res[0][1]) and np.array_equal(res[4][2], res[0][2])
    assert int(res[4][2].sum()) == C * (nc + nd)
    assert np.isfinite(res[4][0]).all()


def test_hmc_lane_group_mfma_bit_exact_vs_host_twin(M, O):
    """mm_hmc_lg.h (HMC on the dense f64 Gaussian: 16 chains per wave, gradient on v_mfma_f64_16x16x4) against the host
    build of mm_hmc_step_noise with the grouped reduction order; and against the one-chain-per-lane kernel to rounding."""
    for dim, cond, C, nc, nd, L, eps, off in ((32, 100.0, 70, 40, 9, 10, 0.05, 0), (32, 1e4, 33, 17, 0, 7, 0.01, 1 << 33),
                                              (16, 50.0, 100, 25, 5, 10, 0.1, 0), (16, 10.0, 16, 3, 2, 1, 0.2, 5)):
        g = M.dist.GaussianND.ill_conditioned(dim, cond, 7)
        init = M.core.init_with_seed(C, dim, 3, np.float64) * 0.3
        s = M.hmc.HMC(g, init, eps, L).set_seed(19)
        assert s.kernel_variant == 3  # the default where it exists
        if off:
            s.set_chain_offset(off)
        out = s.run(nc, nd)
        ref, st, acc = O.engine_host_hmc_grouped_run(g.precision, init, eps, L, nc, nd, seed=19, chain_offset=off)
        name = f"D={dim} cond={cond} C={C}"
        assert np.array_equal(out, ref) and np.array_equal(s.state(), st), name
        assert np.array_equal(s.accept_counts, acc), name
        out2 = s.run(6, 1)  # continues the chains and the stream
        ref2, _, acc2 = O.engine_host_hmc_grouped_run(g.precision, st, eps, L, 6, 1, seed=19, chain_offset=off, iter0=nc + nd)
        assert np.array_equal(out2, ref2) and np.array_equal(s.accept_counts, acc2), name
        # the one-chain-per-lane kernel computes the same transitions with a different summation order
        a3 = M.hmc.HMC(g, init, eps, L).set_seed(19).set_chain_offset(off).run(3, 0)
        a2 = M.hmc.HMC(g, init, eps, L).set_seed(19).set_chain_offset(off).set_kernel_variant(2).run(3, 0)
        np.testing.assert_allclose(a3, a2, rtol=1e-9, atol=1e-11)
    with pytest.raises(Exception):
        M.hmc.HMC(M.dist.StandardNormal(3), M.core.init_with_seed(4, 3, 1, np.float64), 0.1, 5).set_kernel_variant(3)
    # posterior: whitened draws have unit covariance (32-D, cond 100, 4096 chains)
    g = M.dist.GaussianND.ill_conditioned(32, 100.0, 5)
    s = M.hmc.HMC(g, M.core.init_with_seed(4096, 32, 4, np.float64) * 0.1, 0.08, 10).set_seed(6)
    smp = s.run(50, 150).reshape(-1, 32)
    w = smp @ np.linalg.cholesky(g.precision)
    assert np.abs(w.mean(axis=0)).max() < 0.03 and np.abs(np.cov(w.T) - np.eye(32)).max() < 0.05


def test_nuts_lane_group_mfma_bit_exact_vs_host_twin(M, O):
    """mm_nuts_lg.h (16 chains per wave, gradient on v_mfma_f64_16x16x4) against the host build of mm_nuts_step with
    the grouped reduction order (engine_host mode 3): samples, positions, adaptation state and tree shapes."""
    from mini_mcmc_amd.nuts import NUTS

    cases = [
        # dim, cond, chains, n_collect, n_discard, progress, max_depth, offset
        (32, 1e4, 70, 12, 25, False, 10, 0),
        (32, 50.0, 33, 9, 0, False, 10, 5),
        (32, 1e4, 48, 6, 10, True, 4, 0),   # depth cap hit
        (16, 1e3, 100, 10, 20, False, 10, 1 << 33),
        (16, 10.0, 16, 5, 5, True, 10, 0),
    ]
    for dim, cond, C, nc, nd, progress, cap, off in cases:
        g = M.dist.GaussianND.ill_conditioned(dim, cond, 7)
        init = M.core.init_with_seed(C, dim, 3) * 0.3
        ref, pos, ad, nlf = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, dim, [], init, 0.8, nc, nd, seed=19, matrix=g.precision,
                                                   progress=progress, max_depth=cap, chain_offset=off)
        # variants 2 / 3 = tree-depth compaction by launches / by the persistent scheduler (below 2048 chains 3 runs as
        # 1; the scheduler itself is exercised further down), variant 1 = one launch, no compaction
        for variant, start, groups in ((2, None, 0), (2, 0, 2), (2, 3, 3), (1, None, 0), (3, None, 0), (3, 2, 0)):
            s = NUTS(g, init, 0.8, mode=2).set_seed(19).set_max_depth(cap)
            assert s.kernel_variant == 3  # the default where it exists
            s.set_kernel_variant(variant)
            if start is not None:
                s.set_compaction(start, groups)
            if off:
                s.set_chain_offset(off)
            out = s._run(nc, nd, progress, "numpy")
            name = f"D={dim} cond={cond} C={C} variant={variant} start={start} groups={groups}"
            assert np.array_equal(s.leapfrog_counts(), nlf), name
            assert np.array_equal(out, ref), name
            assert np.array_equal(s.positions(), pos), name
            a = s.adapt_state()
            assert np.array_equal(a["epsilon"], ad[:, 0]) and np.array_equal(a["epsilon_bar"], ad[:, 1]), name
            assert np.array_equal(a["h_bar"], ad[:, 2]) and np.array_equal(a["mu"], ad[:, 3]), name
            assert s.depth_histogram().sum() == C * (nc + nd - (0 if progress else 1)), name
    # a second run() continues where the first stopped, in either variant
    g = M.dist.GaussianND.ill_conditioned(32, 100.0, 2)
    init = M.core.init_with_seed(37, 32, 6) * 0.3
    outs = []
    for variant in (1, 2, 3):
        s = NUTS(g, init, 0.8, mode=2).set_seed(4).set_kernel_variant(variant)
        outs.append(np.concatenate([s.run(5, 7), s.run(4, 0), s.run_progress(3, 2)[0]], axis=1))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # the persistent scheduler proper (>= 2048 chains): chains advance independently, through 8 queue shards, and still
    # every chain's samples, adaptation state and tree shapes are those of the single-launch kernel and of the host twin
    g = M.dist.GaussianND.ill_conditioned(32, 1e3, 11)
    init = M.core.init_with_seed(2500, 32, 12) * 0.3
    res = {}
    for variant, start in ((3, 5), (3, 2), (1, 5)):
        s = NUTS(g, init, 0.8, mode=2).set_seed(23).set_kernel_variant(variant).set_compaction(start, 0)
        out = s.run(6, 14)
        res[(variant, start)] = (out, s.positions(), s.leapfrog_counts(), s.adapt_state(), s.depth_histogram())
    for key in ((3, 2), (1, 5)):
        for x, y in zip(res[(3, 5)][:3], res[key][:3]):
            assert np.array_equal(x, y), key
        assert all(np.array_equal(res[(3, 5)][3][k], res[key][3][k]) for k in ("epsilon", "epsilon_bar", "h_bar", "mu")), key
        assert np.array_equal(res[(3, 5)][4], res[key][4]), key
    ref, pos, ad, nlf = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, 32, [], init[:96], 0.8, 6, 14, seed=23, matrix=g.precision)
    assert np.array_equal(res[(3, 5)][0][:96], ref) and np.array_equal(res[(3, 5)][2][:96], nlf)
    # scheduler corner cases: a depth cap below the first queue level, D = 16, continued runs, run_progress semantics
    g16 = M.dist.GaussianND.ill_conditioned(16, 50.0, 3)
    init16 = M.core.init_with_seed(2100, 16, 2) * 0.3
    outs = []
    for variant in (3, 1):
        s = NUTS(g16, init16, 0.8, mode=2).set_seed(9).set_kernel_variant(variant).set_max_depth(3)
        a = s.run(4, 6)
        s.set_max_depth(10)
        b = s.run(3, 0)
        c = s._run(5, 2, True, "numpy")
        outs.append((np.concatenate([a, b, c], axis=1), s.leapfrog_counts(), s.depth_histogram()))
    assert all(np.array_equal(x, y) for x, y in zip(outs[0], outs[1]))
    # the deepest trees the engine allows (12 doublings: every stack level, the HBM part of the first-leaf table and
    # the depth cap are exercised) on a cond-1e6 target, all mappings against the host twin
    g6 = M.dist.GaussianND.ill_conditioned(16, 1e6, 3)
    init6 = M.core.init_with_seed(20, 16, 5) * 0.3
    ref6, _, _, nlf6 = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, 16, [], init6, 0.8, 4, 70, seed=7, matrix=g6.precision,
                                              max_depth=12)
    for variant in (1, 2, 3):
        s = NUTS(g6, init6, 0.8, mode=2).set_seed(7).set_max_depth(12).set_kernel_variant(variant)
        assert np.array_equal(s.run(4, 70), ref6) and np.array_equal(s.leapfrog_counts(), nlf6), variant
        assert s.depth_histogram()[11:].sum() > 0  # trees of 2048+ leaves did occur
    # sharding: the wave a chain sits in (and its neighbours' tree depths) must not matter
    g = M.dist.GaussianND.ill_conditioned(32, 1e3, 1)
    init = M.core.init_with_seed(50, 32, 8) * 0.3
    whole = NUTS(g, init, 0.8, mode=2).set_seed(2).run(8, 12)
    lo = NUTS(g, init[:21], 0.8, mode=2).set_seed(2).run(8, 12)
    hi = NUTS(g, init[21:], 0.8, mode=2).set_seed(2).set_chain_offset(21).run(8, 12)
    assert np.array_equal(np.concatenate([lo, hi]), whole)
    # only where it exists
    with pytest.raises(Exception):
        NUTS(M.dist.StandardNormal(3), M.core.init_with_seed(4, 3, 1), 0.8, mode=2).set_kernel_variant(1)
    # posterior: whitened draws have unit covariance (32-D, cond 100, 1024 chains)
    g = M.dist.GaussianND.ill_conditioned(32, 100.0, 5)
    s = NUTS(g, M.core.init_with_seed(1024, 32, 4) * 0.1, 0.8, mode=2).set_seed(6)
    smp = s.run(60, 150).reshape(-1, 32)
    Lc = np.linalg.cholesky(g.precision)  # A = L L^T  ->  L^T x ~ N(0, I)
    w = smp @ Lc
    assert np.abs(w.mean(axis=0)).max() < 0.03
    assert np.abs(np.cov(w.T) - np.eye(32)).max() < 0.05


def test_nuts_reference_semantics_on_gpu(M, O, kats):
    from mini_mcmc_amd.nuts import NUTS

    # test_chain_1 (nuts.rs:1123-1136): run(1, 0) == the initial position, no transition taken
    k = kats["nuts_chains"][0]
    s = NUTS(M.dist.DiffableGaussian2D(k["mean"], k["cov"]), [k["init"]], k["target_accept"], mode=0).set_seed(42)
    out = s.run(1, 0)
    assert np.array_equal(out.ravel(), np.array(k["init"], dtype=np.float32)) and s.leapfrog_counts()[0] == 0
    # a second run() continues the chain: positions persist, step count persists, init_chain re-derives mu
    a = s.run(5, 3)
    b = s.run(4, 0)
    assert np.array_equal(b[:, 0], a[:, -1])  # with n_discard == 0 row 0 is the current (= last) position
    # sharding by chain offset reproduces the unsharded run
    tgt = M.dist.StandardNormal(3)
    init = M.core.init_with_seed(96, 3, 5)
    whole = NUTS(tgt, init, 0.8, mode=2).set_seed(3).run(12, 6)
    lo = NUTS(tgt, init[:40], 0.8, mode=2).set_seed(3).run(12, 6)
    hi = NUTS(tgt, init[40:], 0.8, mode=2).set_seed(3).set_chain_offset(40).run(12, 6)
    assert np.array_equal(np.concatenate([lo, hi]), whole)


def test_nuts_posterior_and_run_progress_stats(M, O):
    from mini_mcmc_amd.nuts import NUTS

    init = M.core.init_with_seed(2048, 2, 9)
    s = NUTS(M.dist.DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), init, 0.8, mode=0).set_seed(1)
    sample, stats = s.run_progress(300, 200)
    flat = sample.reshape(-1, 2).astype(np.float64)
    assert np.all(np.abs(flat.mean(axis=0) - [0.0, 1.0]) < 0.03)  # within 1 % of the scale (north_star)
    assert np.all(np.abs(np.cov(flat.T) - np.array([[4.0, 2.0], [2.0, 3.0]])) < 0.08)
    assert stats.ess.min > 0.2 * 2048 * 300 and 0.99 < stats.rhat.min <= stats.rhat.max < 1.01
    h = s.depth_histogram()
    assert h[:1].sum() == 0 and h.sum() == 2048 * 500
    # config-5 shaped target (32-D, condition number 1e4, f64) at a small chain count: runs, adapts, stays finite
    g = M.dist.GaussianND.ill_conditioned(32, 1e4, 7)
    s5 = NUTS(g, M.core.init_with_seed(64, 32, 42) * 0.1, 0.8, mode=2).set_seed(42).set_kernel_variant(0)
    out5 = s5.run(20, 30)
    assert np.all(np.isfinite(out5)) and s5.depth_histogram()[1:].sum() == 64 * 49
    ref5, _, _, nlf5 = O.engine_host_nuts_run(2, O.GAUSSIAN_ND, 32, [], M.core.init_with_seed(64, 32, 42) * 0.1, 0.8, 20, 30,
                                              seed=42, matrix=g.precision)
    assert np.array_equal(out5, ref5) and np.array_equal(s5.leapfrog_counts(), nlf5)


@pytest.mark.gpu
@pytest.mark.parametrize("c,n,p", [(40, 2600, 3), (6, 9000, 2), (3, 20001, 1), (130, 4096, 4), (2, 32768, 2), (5, 2050, 2),
                                   (2, 32770, 1), (2, 40000, 2), (1, 100001, 3), (12, 36000, 2), (64, 33000, 2), (400, 2100, 3), (9, 7000, 3), (70, 5001, 2), (33, 12000, 1),
                                   (64, 20000, 2), (3, 250000, 2), (1, 300000, 1), (16, 16384, 17), (700, 4100, 3), (4, 6144, 2), (2, 262144, 1), (24, 10240, 3)])
def test_split_rhat_mean_ess_long_chains_vs_oracle(M, O, c, n, p):
    """Half-chains beyond 1024 draws (the transform cut into N1 residues of 2048-point wave-level transforms: N1 = 2
    mm_chain_fft_long_kernel, N1 = 3 .. 128 mm_chain_fft_res_kernel -- round 6: one residue and parameter per wave, spectrum in
    registers, half-chains up to 131 072 draws; N1 = ceil(m / 1024), ANY count, not the next power of two: the shapes here take
    3, 4, 5, 6, 8, 10, 16, 17, 18, 20, 49, 123, 128, with m = 1024 N1 exactly at (4, 6144, 2) and (24, 10240, 3)) and beyond (mm_lag_sums_any_kernel: any length, straight from global memory;
    (1, 300000, 1)); more than 16 parameters (the means through the any-length moments kernel): R-hat / ESS against
    oracle/stats.c's FFT branch (stats.rs:576-620), odd n (the middle draw dropped), few chains, the first and the last length
    of each path, one chain of 10^5 draws; equal to the direct sums where those still exist; reproducible bit for bit;
    the partial statistics the multi-GPU path exchanges agree with the one-call path."""
    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(c + n)
    x = _ar1(rng, c, n, p)
    x[:, :, 0] += np.arange(c)[:, None] * 0.25
    r0, e0 = O.split_rhat_mean_ess(x)
    r1, e1 = S.split_rhat_mean_ess(x)
    np.testing.assert_allclose(r1, r0, rtol=1e-4)
    np.testing.assert_allclose(e1, e0, rtol=5e-3)
    r2, e2 = S.split_rhat_mean_ess(x)
    assert np.array_equal(r1, r2) and np.array_equal(e1, e2)
    if n <= 9000:
        try:
            S.set_kernel("direct")
            r3, e3 = S.split_rhat_mean_ess(x)
        finally:
            S.set_kernel("auto")
        np.testing.assert_allclose(r3, r1, rtol=1e-5)
        np.testing.assert_allclose(e3, e1, rtol=2e-4)
    import torch

    t = torch.as_tensor(x, device="cuda")
    means, ssq, acov = S.stats_partials(t)
    r4, e4 = S.stats_finish(means.cpu().numpy(), ssq.cpu().numpy(), acov.cpu().numpy())
    np.testing.assert_allclose(r4, r1, rtol=1e-6)
    # the two entry points cut the slabs' totals into different numbers of f32 partial sums: measured 2.4e-5 at (2, 32770, 1)
    # through the 32-residue power spectrum (1e-5 held while that shape went through the O(n^2) sums with f64 totals)
    np.testing.assert_allclose(e4, e1, rtol=1e-4)


@pytest.mark.gpu
def test_split_rhat_mean_ess_refuses_shapes_beyond_its_stated_limits(M):
    """The from-global-memory path (half-chains beyond 16 384 draws) states its limits (include/mmcmc.h, advisor r4): 32-bit
    element and grid counts -> MMCMC_ERR_SHAPE, more than 2^46 lag products (~ten seconds of device time) ->
    MMCMC_ERR_UNSUPPORTED -- decided from the shape alone, before anything reads the sample (a one-element buffer stands in)."""
    import ctypes as C

    import torch

    from mini_mcmc_amd import _lib as L

    lib = L.lib()
    buf = torch.zeros(64 * 33000 * 2, dtype=torch.float32, device="cuda")
    rhat, ess = (C.c_float * 8)(), (C.c_float * 8)()

    def call(c, n, d):
        return lib.mmcmc_split_rhat_mean_ess(C.c_void_p(buf.data_ptr()), 1, L.F32, c, n, d, rhat, ess, 0, None)

    # (round 6: [65536, 40000, 3], refused until then, is inside the power-spectrum path now -- half-chains up to 131 072 draws)
    assert call(4096, 300000, 3) == L.ERR_UNSUPPORTED      # n / 2 > 131072 -> O(n^2) path: 4096 * 3 * 150000^2 = 2.8e14 > 2^46
    try:  # the limit is a setter now (advisor r5): 64 * 2 * 150000^2 = 2.9e12 is inside 2^46 and outside 2^40
        assert lib.mmcmc_stats_set_direct_work_limit(1 << 40) == L.OK
        assert call(64, 300000, 2) == L.ERR_UNSUPPORTED
    finally:
        lib.mmcmc_stats_set_direct_work_limit(1 << 46)
    assert call(64, 33000, 2) == L.OK                       # inside the limits (N1 = 32 residues; values: the parity test above)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_lnu_filter_bound_holds_over_the_whole_lattice(tmp_path):
    """mm_ratio_exceeds_lnu_f32 (csrc/mm_rng.h, round 5): the MH split kernel decides `log_accept_ratio > ln u` from the
    hardware's base-2 logarithm and evaluates the table logarithm -- the DEFINITION of ln u in the f32 stream -- only inside a
    narrow band.  tests/hip/lnu_filter_lattice.hip, compiled here with hipcc against the engine's header, runs over ALL 2^24
    accept uniforms: the table-free value stays within half the band of the table's (the filter's safety margin), and the
    filtered decision equals the plain comparison for 16 probe ratios per uniform placed at, next to and around both values
    (and 0, +-inf, NaN).  The full-size check that every accept decision of config 2 is unchanged is
    test_config2_full_size_checksum."""
    import json
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "lnu_lattice")
    src = os.path.join(ROOT, "tests", "hip", "lnu_filter_lattice.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(ROOT, "mini_mcmc_amd", "csrc"),
                        src, "-o", exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["uniforms"] == 1 << 24 and j["mismatches"] == 0
    assert j["max_dev_over_band"] < 0.5, j
    assert j["probes_in_band"] >= 3 * (1 << 24)  # the probes really exercise the exact branch
    # the paired MH stream's filter (mm_ratio_exceeds_lnu_mhp): the accept uniform's low byte supplied on demand only -- same
    # decisions over all 2^16 x 2^8 uniforms and 20 probe ratios each; the byte is demanded for the probes placed inside the
    # interval [ln u_lo, ln u_hi] (at least seven of the twenty) and not for those two bands outside it
    assert j["paired_mismatches"] == 0
    assert 7 * (1 << 24) <= j["paired_low_byte_demanded"] <= 16 * (1 << 24), j
