"""Pin the CPU oracle against every known-answer test the reference holds for the hot path.

CPU only.  If these fail, no parity claim made against the oracle means anything.
"""
import math

import numpy as np
import pytest


# ---------------------------------------------------------------- third-party RNG stack


def test_smallrng_seeding_and_stream(O):
    r = O.SmallRng(0)
    assert r.state[0] == 0xE220A8397B1DCDAF
    r = O.SmallRng(42)
    assert r.state == [0xBDD732262FEB6E95, 0x28EFE333B266F103, 0x47526757130F9F52, 0x581CE1FF0E4AE394]
    assert [r.next_u64() for _ in range(3)] == [0xD0764D4F4476689F, 0x519E4174576F3791, 0xFBE07CFB0C24ED8C]
    r = O.SmallRng(42)
    got = [float(r.f32()) for _ in range(3)]
    assert got == pytest.approx([0.81430513, 0.31882101, 0.98389411], abs=1e-8)


def test_ziggurat_tables(O):
    L = O.lib()
    nx, ex = L.rc_zig_norm_x(), L.rc_zig_exp_x()
    assert nx[0] == pytest.approx(3.91075795953709, rel=1e-14)
    assert nx[1] == 3.654152885361008796
    assert nx[2] == pytest.approx(3.4492782985609645, rel=1e-14)
    assert nx[255] == pytest.approx(0.2152418959132738, rel=1e-13)
    assert nx[256] == 0.0
    assert ex[0] == pytest.approx(8.697117470131053, rel=1e-14)
    assert ex[256] == 0.0
    nf = L.rc_zig_norm_f()
    assert nf[256] == 1.0


def test_init_det_matches_reference_stream(O):
    # core.rs:404-435 init_det(4, 2) == init_with_seed(4, 2, 42); values from the validated stream (App. A)
    got = O.init_det(4, 2, np.float32)
    exp = np.array(
        [[0.83439755, -0.51496293], [1.40772757, 0.46445486], [0.95366687, 0.27411556], [-1.37731726, 0.41445339]],
        dtype=np.float32,
    )
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-7)


def test_normal_and_exp_moments(O):
    r = O.SmallRng(7)
    z = np.array([r.normal() for _ in range(200000)])
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.02
    assert abs((z**4).mean() - 3) < 0.1  # kurtosis: tails come from the ziggurat tail branch
    e = np.array([r.exp1() for _ in range(200000)])
    assert abs(e.mean() - 1) < 0.01 and abs(e.var() - 1) < 0.03 and e.min() > 0


# ---------------------------------------------------------------- NUTS (nuts.rs:1050-1222)


def test_find_reasonable_epsilon(O, kats):
    k = kats["nuts_find_reasonable_epsilon"]
    eps = O.find_reasonable_epsilon(O.standard_normal(k["dim"]), k["position"], k["mom"], mode=0)
    assert eps == k["expected_epsilon"]


def test_build_tree(O, kats):
    k = kats["nuts_build_tree"]
    t = O.diffable_gaussian2d(k["target"]["mean"], k["target"]["cov"])
    r = O.build_tree(
        t, k["position"], k["mom"], k["grad"], k["logu"], k["v"], k["j"], k["epsilon"], k["joint_0"],
        rng_seed=k["rng_seed"], mode=0,
    )
    e = k["expected"]
    for name in ["position_minus", "mom_minus", "grad_minus", "position_plus", "mom_plus", "grad_plus",
                 "position_prime", "grad_prime"]:
        np.testing.assert_allclose(r[name], e[name], rtol=k["rtol"], atol=k["atol"], err_msg=name)
    assert r["n_prime"] == e["n_prime"]
    assert r["s_prime"] == e["s_prime"]
    assert r["n_alpha_prime"] == e["n_alpha_prime"]
    assert abs(r["logp_prime"] - e["logp_prime"]) < e["logp_atol"]
    assert abs(r["alpha_prime"] - e["alpha_prime"]) < e["alpha_atol"]


def test_seeded_chains(O, kats):
    for k in kats["nuts_chains"]:
        t = O.diffable_gaussian2d(k["mean"], k["cov"])
        s = O.NUTS(t, k["init"], k["target_accept"], mode=0)
        if "chain_seed" in k:
            s.set_chain_seed(0, k["chain_seed"])
        else:
            s.set_seed(k["sampler_seed"])
        out = s.run(k["n_collect"], k["n_discard"], n_threads=1)
        assert out.shape == (1, k["n_collect"], 2)
        np.testing.assert_allclose(out.ravel(), k["expected"], rtol=k["rtol"], atol=k["atol"], err_msg=k["source"])


def test_nuts_adaptation_trace(O, kats):
    # SURVEY.md App. A trace of test_chain_2: eps0 = 2.0 found by find_reasonable_epsilon, mu = ln 20
    k = kats["nuts_chains"][1]
    t = O.diffable_gaussian2d(k["mean"], k["cov"])
    s = O.NUTS(t, k["init"], k["target_accept"], mode=0).set_chain_seed(0, 42)
    s.run(1, 0, n_threads=1)  # init_chain only (N-1 = 0 steps)
    st = s.chain_state()
    assert st["epsilon"] == 2.0 and st["mu"] == pytest.approx(math.log(20.0))


# ---------------------------------------------------------------- diagnostics (stats.rs:704-834)


def test_multichain_tracker_rhat(O, kats):
    for k in kats["multichain_tracker_rhat"]:
        states = np.array([k["step0"], k["step1"]], dtype=np.float32)
        rhat, _ = O.multichain_tracker_rhat(states)
        np.testing.assert_allclose(rhat, np.array(k["expected"], dtype=np.float32), rtol=0, atol=k["atol"])


def test_autocov_bf_and_fft(O, kats):
    for k in kats["autocov"]:
        data = np.array(k["data"], dtype=np.float32)
        exp = np.array(k["expected"], dtype=np.float32)
        np.testing.assert_allclose(O.autocov_bf(data), exp, rtol=0, atol=k["atol"])
        np.testing.assert_allclose(O.autocov_fft(data), exp, rtol=0, atol=k["atol"])


def test_autocov_fft_equals_bf_on_long_series(O):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 3)).astype(np.float32).cumsum(axis=0) * 0.1
    np.testing.assert_allclose(O.autocov_fft(x), O.autocov_bf(x), rtol=2e-4, atol=2e-5)


def test_ess_1(O, kats):
    k = kats["ess_1"]
    r = O.SmallRng(k["seed"])
    data = np.array([[r.f32() for _ in range(k["n"])] for _ in range(k["chains"])], dtype=np.float32)
    st = O.run_stats(data[:, :, None])
    assert st["ess"]["min"] > k["ess_min_gt"]
    assert st["rhat"]["max"] < k["rhat_max_lt"]


def test_split_rhat_against_numpy_f64(O):
    rng = np.random.default_rng(1)
    c, n, p = 6, 240, 3
    x = rng.standard_normal((c, n, p)).astype(np.float32)
    x[:, :, 1] += np.arange(c)[:, None] * 0.5  # chains disagree in param 1
    rhat, ess = O.split_rhat_mean_ess(x)
    half = n // 2
    sp = np.concatenate([x[:, :half], x[:, n - half:]], axis=0).astype(np.float64)
    m = sp.mean(axis=1)
    b = ((m - m.mean(axis=0)) ** 2).sum(axis=0) * half / (2 * c - 1)
    w = ((sp - m[:, None, :]) ** 2).mean(axis=1).mean(axis=0)
    v = (half - 1) / half * w + b / half
    np.testing.assert_allclose(rhat, np.sqrt(w / v), rtol=2e-5)  # quirk Q7: inverse of Stan's R-hat
    assert rhat[1] < 0.9 and ess[1] < ess[0]
    assert np.all(ess > 0)


def test_basic_stats(O):
    st = O.basic_stats(np.array([3.0, 1.0, 2.0, 5.0], dtype=np.float32))
    assert (st["min"], st["max"], st["median"]) == (1.0, 5.0, 2.0)  # sorted desc -> [5,3,2,1][2] = 2
    assert st["mean"] == pytest.approx(2.75) and st["std"] == pytest.approx(np.std([3, 1, 2, 5], ddof=1), rel=1e-6)


def test_chain_tracker_rhat_runs(O):
    rng = np.random.default_rng(2)
    init = rng.standard_normal((4, 2)).astype(np.float32)
    st = rng.standard_normal((4, 50, 2)).astype(np.float32)
    rhat, pa = O.chain_trackers_rhat(init, st)
    assert rhat.shape == (2,) and np.all(np.isfinite(rhat)) and np.all((pa > 0) & (pa <= 1))


# ---------------------------------------------------------------- densities (distributions.rs tests)


def test_isotropic_gaussian_unnorm_logp(O, kats):
    for k in kats["isotropic_gaussian_unnorm_logp"]:
        d = len(k["x"])
        lp = O.isotropic_gaussian(k["std"], d).logp(k["x"])
        log_norm = -(d / 2.0) * (math.log(2.0) + math.log(math.pi) + 2.0 * math.log(k["std"]))
        assert abs(math.exp(lp + log_norm) - k["true_p"]) < k["tol"]


def test_gaussian2d_logp(O, kats):
    k = kats["gaussian2d_normalized_logp"]
    cov = np.array(k["cov"])
    lp = O.gaussian2d(k["mean"], cov).logp(k["x"])
    norm = -math.log(2 * math.pi) - 0.5 * math.log(abs(np.linalg.det(cov)))
    assert abs(lp + norm - k["expected_logp"]) < k["tol"]


def test_analytic_gradients_match_finite_differences(O):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((5, 5))
    A = A @ A.T + np.eye(5)
    targets = [
        O.gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]),
        O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]),
        O.isotropic_gaussian(1.5, 4),
        O.rosenbrock2d(1.0, 100.0),
        O.rosenbrock_nd(3),
        O.rosenbrock_nd(6),
        O.standard_normal(3),
        O.gaussian_nd(A),
    ]
    for t in targets:
        x = rng.standard_normal(t.dim) * 0.7
        lp, g = t.logp_grad(x)
        assert lp == pytest.approx(t.logp(x), rel=1e-12, abs=1e-12)
        num = np.zeros(t.dim)
        h = 1e-6
        for i in range(t.dim):
            xp, xm = x.copy(), x.copy()
            xp[i] += h
            xm[i] -= h
            num[i] = (t.logp(xp) - t.logp(xm)) / (2 * h)
        np.testing.assert_allclose(g, num, rtol=2e-5, atol=2e-5, err_msg=f"kind {t.kind} dim {t.dim}")
        lp32, g32 = t.logp_grad(x, np.float32)
        np.testing.assert_allclose(g32, g, rtol=2e-3, atol=2e-3)


def test_diffable_gaussian_gradient_kat(O):
    # SURVEY.md App. C check against nuts.rs:1103-1105: grad at position_plus of test_build_tree
    t = O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    _, g = t.logp_grad(np.array([-0.0198, 0.97025]), np.float32)
    np.testing.assert_allclose(g, [-1.250e-05, 9.925e-03], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- MH / HMC statistical bands


def _moments(sample):
    flat = sample.reshape(-1, sample.shape[-1]).astype(np.float64)
    return flat.mean(axis=0), np.cov(flat.T, ddof=1)


def test_mh_3_chains_statistical(O, kats):
    # metropolis_hastings.rs:338-391: reference seeds, reference quirks (Q1 shared proposal clone, Q2)
    k = kats["mh_statistical"]
    t = O.gaussian2d(k["mean"], k["cov"])
    init = O.init_det(k["n_chains"], 2)
    mh = O.MetropolisHastings(t, k["proposal_std"], init, np.float64, proposal_seed=k["seed"]).seed(k["seed"])
    sample = mh.run(k["sample_size"] // k["n_chains"], k["burnin"])
    assert sample.shape == (k["n_chains"], k["sample_size"] // k["n_chains"], 2)
    mean, cov = _moments(sample)
    assert np.all(np.abs(mean - k["mean"]) < k["mean_atol"])
    assert np.all(np.abs(cov - np.array(k["cov"])) < k["cov_atol"])


def test_mh_single_chain_accept_and_reject(O, kats):
    # tests/metrohast_2d_gaussian_test.rs:16-92
    k = kats["mh_single_chain_statistical"]

    def run(cov):
        t = O.gaussian2d(k["mean"], cov)
        mh = O.MetropolisHastings(t, k["proposal_std"], [k["init"]], np.float64, proposal_seed=k["seed"]).seed(k["seed"])
        return mh.run(k["n_collect"], k["n_discard"])

    mean, cov = _moments(run(k["cov"]))
    assert np.all(np.abs(mean - k["mean"]) < k["mean_atol"])
    assert np.abs(cov - np.array(k["cov"])).max() < k["cov_atol"]
    _, cov_false = _moments(run(k["false_cov"]))
    assert np.abs(cov_false - np.array(k["cov"])).max() > k["false_target_cov_diff_gt"]


def test_mh_q1_shared_proposal_noise(O):
    # Quirk Q1: every chain owns an identical clone of the proposal generator, so two chains started at the
    # same point and given the same accept seed would coincide; with different accept seeds they share noise.
    t = O.gaussian2d([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]])
    init = np.zeros((2, 2))
    mh = O.MetropolisHastings(t, 1.0, init, np.float64, proposal_seed=5).seed(9)
    s = mh.run(50, 0, n_threads=1)
    d0 = np.diff(np.concatenate([init[:1], s[0]]), axis=0)
    d1 = np.diff(np.concatenate([init[1:], s[1]]), axis=0)
    both = (np.abs(d0).sum(axis=1) > 0) & (np.abs(d1).sum(axis=1) > 0)
    assert both.sum() > 5
    np.testing.assert_allclose(d0[both], d1[both], rtol=0, atol=1e-12)


@pytest.mark.parametrize("n_runs", [100])
def test_hmc_ess_band(O, kats, n_runs):
    # hmc.rs:576-787 (ignored benchmark in the reference): 100 runs and the same bands, as there
    k = kats["hmc_ess_bands"]
    t = O.diffable_gaussian2d(k["mean"], k["cov"], t_is_f32=True)
    e1, e2, rh = [], [], []
    for run in range(n_runs):
        init = O.init_with_seed(k["n_chains"], 2, 1000 + run, np.float32)
        h = O.HMC(t, init, k["step_size"], k["n_leapfrog"], np.float32).seed_global(run)
        s = h.run(k["n_collect"], k["n_discard"], n_threads=1)
        rhat, ess = O.split_rhat_mean_ess(s)
        e1.append(ess[0])
        e2.append(ess[1])
        rh.append(rhat)
    lo1, hi1 = k["ess1_band"]
    lo2, hi2 = k["ess2_band"]
    assert lo1 <= np.mean(e1) <= hi1, np.mean(e1)
    assert lo2 <= np.mean(e2) <= hi2, np.mean(e2)
    m = np.mean(rh, axis=0)
    assert np.all((m >= k["rhat_band"][0]) & (m <= k["rhat_band"][1]))


def test_mh_ess_band(O, kats):
    # metropolis_hastings.rs:417-513 (ignored benchmark): mean ESS(x1) in [65, 125]
    k = kats["mh_statistical"]
    band = kats["mh_ess_bands"]["ess1_band"]
    t = O.gaussian2d(k["mean"], k["cov"])
    e1 = []
    for run in range(30):
        init = O.init_with_seed(3, 2, 2000 + run)
        mh = O.MetropolisHastings(t, 1.0, init, np.float64, proposal_seed=run).seed(run)
        s = mh.run(1000, 500, n_threads=1)
        _, ess = O.split_rhat_mean_ess(s)
        e1.append(ess[0])
    assert band[0] <= np.mean(e1) <= band[1], np.mean(e1)
