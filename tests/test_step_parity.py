"""Level B parity (CPU): the engine's transition arithmetic (explicit-FMA, analytic gradients, cached log-density,
q-terms dropped; mini_mcmc_amd/csrc/mm_samplers.h compiled for the host) against the independent, reference-ordered
restatement (oracle/mh_hmc.c: metropolis_hastings.rs:303-315, hmc.rs:304-431 op for op, pinned by the reference's
known-answer tests) -- both driven by the SAME noise (the engine's counter-based stream, restated independently in
oracle/orng.c).  Agreement is to rounding, not bitwise: the tolerances below are the stated f32 / f64 bounds.
"""
import numpy as np
import pytest

GAUSS = [0.0, 1.0, 4.0, 2.0, 2.0, 3.0]


def _cases(O):
    rng = np.random.default_rng(11)
    A = rng.standard_normal((8, 8))
    A = A @ A.T / 8 + np.eye(8)
    return [
        ("mh", O.GAUSSIAN2D, 2, GAUSS, None, O.gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 1.0, 0),
        ("mh", O.ISOTROPIC_GAUSSIAN, 3, [1.3], None, O.isotropic_gaussian(1.3, 3), 0.8, 0),
        ("mh", O.ROSENBROCK_ND, 4, [], None, O.rosenbrock_nd(4), 0.05, 0),
        ("mh", O.GAUSSIAN_ND, 8, [], A, O.gaussian_nd(A), 0.3, 0),
        ("hmc", O.ROSENBROCK_ND, 3, [], None, O.rosenbrock_nd(3), 0.032, 10),
        ("hmc", O.ROSENBROCK2D, 2, [1.0, 100.0], None, O.rosenbrock2d(1.0, 100.0), 0.01, 20),
        ("hmc", O.DIFFABLE_GAUSSIAN2D, 2, GAUSS, None, None, 0.1, 10),
        ("hmc", O.STANDARD_NORMAL, 5, [], None, O.standard_normal(5), 0.3, 5),
        ("hmc", O.GAUSSIAN_ND, 8, [], A, O.gaussian_nd(A), 0.2, 6),
    ]


def _oracle_sampler(O, sampler, kind, otgt, init, scale, L, dtype):
    if kind == O.DIFFABLE_GAUSSIAN2D:
        otgt = O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=(dtype == np.float32))
    if sampler == "mh":
        return O.MetropolisHastings(otgt, scale, init, dtype)
    return O.HMC(otgt, init, scale, L, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_single_transition_agrees_to_rounding(O, dtype):
    # one transition from identical states with identical noise: every chain's new state within a few ulp-scale
    # units, and the accept decision identical except on a razor's edge (none expected among 512 chains)
    for sampler, kind, dim, params, mat, otgt, scale, L in _cases(O):
        init = O.init_with_seed(512, dim, 5, dtype) * 0.7
        _, st_e, acc_e = O.engine_host_run(sampler, kind, dim, params, init, scale, 1, 0, seed=9, n_leapfrog=L,
                                           matrix=mat, dtype=dtype, n_threads=1)
        s = _oracle_sampler(O, sampler, kind, otgt, init, scale, L, dtype).use_engine_stream(9)
        s.run(1, 0, n_threads=1)
        st_o, acc_o = s.state(), s.accept_counts
        name = f"{sampler} kind={kind} D={dim} {dtype.__name__}"
        assert np.array_equal(acc_e, acc_o), name
        tol = 3e-5 if dtype == np.float32 else 1e-12
        scale_x = np.maximum(1.0, np.abs(st_o.astype(np.float64)))
        assert np.max(np.abs(st_e.astype(np.float64) - st_o.astype(np.float64)) / scale_x) < tol, name
        assert 0.02 < acc_e.mean() <= 1.0, name  # the case exercises both branches


def test_short_runs_track_each_other_f64(O):
    # in f64 rounding differences stay far below the accept margins over a short run: same accept counts, samples
    # equal to ~1e-9 even for the stiff Rosenbrock dynamics
    for sampler, kind, dim, params, mat, otgt, scale, L in _cases(O):
        init = O.init_with_seed(64, dim, 6, np.float64) * 0.7
        out_e, _, acc_e = O.engine_host_run(sampler, kind, dim, params, init, scale, 40, 5, seed=3, n_leapfrog=L,
                                            matrix=mat, dtype=np.float64, n_threads=2)
        s = _oracle_sampler(O, sampler, kind, otgt, init, scale, L, np.float64).use_engine_stream(3)
        out_o = s.run(40, 5, n_threads=2)
        assert np.array_equal(acc_e, s.accept_counts), (sampler, kind, dim)
        np.testing.assert_allclose(out_e, out_o, rtol=1e-7, atol=1e-7, err_msg=f"{sampler} kind={kind} D={dim}")


def test_f32_accept_counts_agree(O):
    # in f32 individual trajectories of a chaotic target drift apart, but accept decisions only flip on a razor's
    # edge: over 128 chains x 60 transitions the accept counts of the two arithmetics must agree almost everywhere
    for sampler, kind, dim, params, mat, otgt, scale, L in _cases(O):
        init = O.init_with_seed(128, dim, 8, np.float32) * 0.7
        _, _, acc_e = O.engine_host_run(sampler, kind, dim, params, init, scale, 50, 10, seed=4, n_leapfrog=L,
                                        matrix=mat, dtype=np.float32, n_threads=2)
        s = _oracle_sampler(O, sampler, kind, otgt, init, scale, L, np.float32).use_engine_stream(4)
        s.run(50, 10, n_threads=2)
        agree = np.mean(acc_e == s.accept_counts)
        assert agree > 0.9, (sampler, kind, dim, agree)
        assert abs(acc_e.mean() - s.accept_counts.mean()) < 0.5


def test_engine_mh_moments_match_reference_bands(O, kats):
    # level D on the CPU: the engine arithmetic + engine stream reproduces the target's moments inside the
    # reference's own tolerances (metropolis_hastings.rs:379-380), with independent per-chain proposal noise
    k = kats["mh_statistical"]
    init = O.init_det(64, 2)
    out, _, acc = O.engine_host_run("mh", O.GAUSSIAN2D, 2, [k["mean"][0], k["mean"][1], 4.0, 2.0, 2.0, 3.0], init,
                                    k["proposal_std"], 2000, k["burnin"], seed=k["seed"], dtype=np.float64)
    flat = out.reshape(-1, 2)
    assert np.all(np.abs(flat.mean(axis=0) - k["mean"]) < k["mean_atol"] / 3)
    assert np.all(np.abs(np.cov(flat.T) - np.array(k["cov"])) < k["cov_atol"] / 3)
    rate = acc.mean() / (2000 + k["burnin"])
    assert 0.4 < rate < 0.9


def test_engine_hmc_ess_band(O, kats):
    # hmc.rs:764-786 band with the engine arithmetic on the host
    k = kats["hmc_ess_bands"]
    groups = 40
    init = O.init_with_seed(3 * groups, 2, 11, np.float32)
    out, _, _ = O.engine_host_run("hmc", O.DIFFABLE_GAUSSIAN2D, 2, GAUSS, init, k["step_size"], k["n_collect"],
                                  k["n_discard"], seed=3, n_leapfrog=k["n_leapfrog"], dtype=np.float32)
    ess = np.array([O.split_rhat_mean_ess(out[3 * i:3 * i + 3])[1] for i in range(groups)])
    m = ess.mean(axis=0)
    assert k["ess1_band"][0] <= m[0] <= k["ess1_band"][1], m
    assert k["ess2_band"][0] <= m[1] <= k["ess2_band"][1], m


def test_nuts_accept_statistic_branch_form_gives_the_bits_of_min_1_exp(O):
    """mm_accept_stat (csrc/mm_nuts.h, round 6) evaluates the leaf's acceptance statistic min(1, exp(d)) (nuts.rs:823-826) as
    `a = 1; if (d < 0) a = exp(d)` so that a wave with no d < 0 among its working lanes skips the exponential.  The claim is
    bit-identity for EVERY d: checked here on the host build of the same header over a million draws of both signs and
    scales, the neighbourhood of zero, signed zeros, infinities, NaN and the exponential's under- / overflow thresholds."""
    rng = np.random.default_rng(5)
    for dt in (np.float64, np.float32):
        d = np.concatenate([
            rng.standard_normal(400_000) * 3.0, rng.standard_normal(200_000) * 1e-6, rng.standard_normal(200_000) * 300.0,
            -np.abs(rng.standard_normal(200_000)) * 1e-12,
            np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -745.2, -745.0, 709.8, 710.0, -103.98, -103.9, 88.73, -1e-300, 1e-300,
                      -5e-324, 5e-324, np.nextafter(0.0, -1.0), np.nextafter(0.0, 1.0)]),
        ]).astype(dt)
        new, old = O.engine_host_accept_stat(d)
        assert np.array_equal(new.view(np.uint64 if dt == np.float64 else np.uint32), old.view(np.uint64 if dt == np.float64 else np.uint32))
        assert np.all(new[~np.isnan(d)] <= 1.0) and np.all(new[d >= 0] == 1.0) and new[np.isnan(d)].tolist() == [1.0]
