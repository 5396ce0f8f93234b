"""Running diagnostics on the GPU: MultiChainTracker (stats.rs:189-306) as a device-side tracker fed
blocks of states, against the oracle's restatement (pinned to stats.rs:704-752 in tests/test_oracle_pins.py)."""
import numpy as np
import pytest


def test_tracker_abi_symbols():
    import mini_mcmc_amd

    lib = mini_mcmc_amd.lib()
    for sym in ("create", "steps", "stats", "init_last", "chain_stats", "n", "destroy"):
        assert hasattr(lib, "mmcmc_tracker_" + sym)


@pytest.mark.gpu
def test_tracker_matches_reference_semantics(O):
    import torch

    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(8)
    # parameters 2 ... 8: one wave per parameter (tracker_step_dims_kernel); 1: all in one wave; f64 input up to 4
    for chains, steps, params in ((4, 30, 2), (300, 57, 3), (20000, 12, 3), (70, 40, 5), (129, 33, 8), (64, 32, 1), (100, 49, 4)):
        # states with repeats (rejections): a chain keeps its previous state with probability 0.35
        x = rng.standard_normal((steps, chains, params)).astype(np.float32)
        keep = rng.random((steps, chains)) < 0.35
        for t in range(1, steps):
            x[t][keep[t]] = x[t - 1][keep[t]]
        rhat_o, p_o = O.multichain_tracker_rhat(x)  # [steps, chains, params]
        cm = np.ascontiguousarray(x.transpose(1, 0, 2))  # [chains, steps, params]
        # (a) one call with the whole block, from the host; (b) step by step / in uneven blocks, from HBM; (c) f64 input
        ta = S.MultiChainTracker(chains, params).step(cm)
        tb = S.MultiChainTracker(chains, params)
        tdev = torch.as_tensor(cm, device="cuda")
        tb.step(tdev[:, 0, :].contiguous())
        t = 1
        for k in (1, 5, 3, steps):
            k = min(k, steps - t)
            if k > 0:
                tb.step(tdev, t0=t, k=k)
                t += k
        tc = S.MultiChainTracker(chains, params).step(cm.astype(np.float64))
        for tr in (ta, tb, tc):
            assert tr.n == steps
            np.testing.assert_allclose(tr.rhat(), rhat_o, rtol=2e-5)
            assert abs(float(tr.max_rhat()) - float(rhat_o.max())) <= 2e-5 * float(rhat_o.max())
            # exact also where the history is longer than the kernel replays: the replay of the last 4096 flags from p = 0
            # and from p = 1 meets in one number, and a step is a monotone map of p (tracker_paccept_kernel's certificate)
            assert tr.p_accept == p_o
        assert ta.p_accept == tb.p_accept
        # rows in tiles of 16 through the coalescing kernel (a) or lane by lane in short blocks (b): the same numbers
        assert np.array_equal(ta.rhat(), tb.rhat())
        ra, mxa, pa = ta.chain_stats()
        rb, mxb, pb = tb.chain_stats()
        assert np.array_equal(ra, rb) and mxa == mxb and pa == pb


@pytest.mark.gpu
def test_tracker_p_accept_when_the_certificate_fails(O):
    """Every flag 1 (no state repeats): from p = 0 the rounded recurrence stalls a few ulps below 1, from p = 1 it stays at
    1, so the two-sided replay does not meet and the kernel falls back to the sequential replay of the last 16 384 flags
    from 0.5 -- still the oracle's number to 1e-6 (and in fact the same stalled value)."""
    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(3)
    chains, steps, params = 5000, 9, 2
    x = rng.standard_normal((steps, chains, params)).astype(np.float32)
    _, p_o = O.multichain_tracker_rhat(x)
    tr = S.MultiChainTracker(chains, params).step(np.ascontiguousarray(x.transpose(1, 0, 2)))
    assert abs(float(tr.p_accept) - float(p_o)) < 1e-6 and float(p_o) > 0.999


@pytest.mark.gpu
def test_per_chain_trackers_match_reference_semantics(O):
    """ChainTracker + collect_rhat (stats.rs:26-178), what the generic run_progress shows (core.rs:90-140, 255-300)."""
    import torch

    from mini_mcmc_amd import stats as S

    rng = np.random.default_rng(9)
    for chains, steps, params in ((4, 40, 2), (257, 33, 3), (5000, 20, 1)):
        init = rng.standard_normal((chains, params)).astype(np.float32)
        x = rng.standard_normal((chains, steps, params)).astype(np.float32)
        keep = rng.random((chains, steps)) < 0.4
        keep[:, 0] &= rng.random(chains) < 0.5
        x[:, 0][keep[:, 0]] = init[keep[:, 0]]          # a rejected first step repeats the initial state
        for t in range(1, steps):
            x[:, t][keep[:, t]] = x[:, t - 1][keep[:, t]]
        if params > 1:                                   # first coordinate unchanged but another one moved: Q12
            x[0, 0, 0] = init[0, 0]
            x[0, 0, 1] = init[0, 1] + 1.0
        rhat_o, p_o = O.chain_trackers_rhat(init, x)
        tr = S.MultiChainTracker(chains, params).init_last(init)
        tdev = torch.as_tensor(x, device="cuda")
        tr.step(tdev, t0=0, k=7).step(tdev, t0=7, k=steps - 7)
        rhat, mx, p = tr.chain_stats()
        np.testing.assert_allclose(rhat, rhat_o, rtol=2e-5)
        assert abs(float(mx) - float(np.nanmax(rhat_o))) <= 2e-5 * float(np.nanmax(rhat_o))
        assert abs(float(p) - float(p_o.astype(np.float64).mean())) < 1e-6  # per-chain EMAs exact, their mean in f64


@pytest.mark.gpu
def test_run_progress_with_tracker_on_gpu(O):
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import DiffableGaussian2D, Gaussian2D, IsotropicGaussian
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    seen = []
    s = HMC(DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), init_with_seed(4096, 2, 3, np.float32), 0.1, 10).set_seed(5)
    sample, stats = s.run_progress(200, 50, every=40, callback=lambda d, p, r: seen.append((d, p, r)))
    assert sample.shape == (4096, 200, 2) and [d for d, _, _ in seen] == [40, 80, 120, 160, 200]
    assert all(0.8 < p < 1.001 for _, p, _ in seen), seen       # HMC at eps 0.1 accepts nearly always (f32 EMA: 1 + 1 ulp possible)
    assert seen[-1][2] < seen[0][2] and 1.0 <= seen[-1][2] < 1.1  # running R-hat falls towards 1
    assert s.tracker.n == 201                                   # the start state + every collected state
    # the tracker fed by run_progress equals the oracle's tracker fed the same states
    start = None
    s2 = HMC(DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), init_with_seed(256, 2, 3, np.float32), 0.1, 10).set_seed(5)
    s2.run(0, 20, collect=False)
    start = s2.state()
    smp, _ = s2.run_progress(30, 0, every=7)
    states = np.concatenate([start[None], smp.transpose(1, 0, 2)], axis=0)
    rhat_o, p_o = O.multichain_tracker_rhat(states)
    np.testing.assert_allclose(s2.tracker.rhat(), rhat_o, rtol=2e-5)
    assert s2.tracker.p_accept == p_o
    # MH: acceptance near the textbook rate for a unit Gaussian with a unit proposal (~0.5-0.6 in 2-D)
    mh = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(8192, 2, 42, np.float32)).seed(42)
    _, st = mh.run_progress(100, 100)
    assert 0.4 < float(mh.tracker.p_accept) < 0.75 and st.rhat.max < 1.05


@pytest.mark.gpu
def test_run_progress_c_abi_feeding_orders_vs_oracle(O):
    """mmcmc_{mh,hmc,nuts}_run_progress (csrc/mm_progress.hip): the tracker each of them hands back equals the oracle's
    tracker fed in the reference's order -- HMC: positions after the burn-in, then the collected states (hmc.rs:229-262);
    MH and NUTS: per-chain ChainTrackers on the initial state, then EVERY state, burn-in included (core.rs:103-112,
    nuts.rs:486-506) -- and RunStats is RunStats::from(sample)."""
    import ctypes as C

    from mini_mcmc_amd import _lib as L
    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import DiffableGaussian2D, Gaussian2D, IsotropicGaussian, RosenbrockND
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
    from mini_mcmc_amd.nuts import NUTS

    # MH: a twin with the same seed gives the states of the burn-in the runner does not return
    n_chains, nc, nd = 300, 37, 23
    init = init_with_seed(n_chains, 2, 7, np.float32)
    tgt = Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    mh = MetropolisHastings(tgt, IsotropicGaussian(1.0), init).seed(11)
    seen = []
    sample, stats = mh.run_progress(nc, nd, every=10, callback=lambda d, p, r: seen.append((d, p, r)))
    twin = MetropolisHastings(tgt, IsotropicGaussian(1.0), init).seed(11)
    every_state = twin.run(nc + nd, 0)
    assert np.array_equal(sample, every_state[:, nd:, :])          # segmented launches = one run
    assert [d for d, _, _ in seen] == [10, 20, 23, 33, 43, 53, 60]   # burn-in chunks, then the collected ones
    rhat_o, p_o = O.chain_trackers_rhat(init, every_state)
    rhat, mx, p = mh.tracker.chain_stats()
    assert mh.tracker.n == nc + nd
    np.testing.assert_allclose(rhat, rhat_o, rtol=2e-5)
    assert abs(float(p) - float(p_o.astype(np.float64).mean())) < 1e-6
    assert abs(seen[-1][1] - float(p)) < 1e-6 and abs(seen[-1][2] - float(mx)) <= 2e-5 * float(mx)
    ref = S.run_stats(sample)
    assert stats.ess.mean == ref.ess.mean and stats.rhat.max == ref.rhat.max
    # ess_from_chainstats (stats.rs:668-671) with the tracker the runner hands back
    ess = S.ess_from_chainstats(sample, mh.tracker)
    ess_o = O.ess_from_chainstats(sample, init, every_state)
    np.testing.assert_allclose(ess, ess_o, rtol=2e-3)
    w, v = mh.tracker.within_var()
    assert np.all(w > 0) and np.all(v > 0)
    # odd chain count and a long sample (the power-spectrum branch above 100 draws, stats.rs:549)
    mh2 = MetropolisHastings(tgt, IsotropicGaussian(1.0), init[:33]).seed(3)
    s2, _ = mh2.run_progress(256, 0)
    np.testing.assert_allclose(S.ess_from_chainstats(s2, mh2.tracker), O.ess_from_chainstats(s2, init[:33], s2), rtol=2e-3)

    # NUTS: every state through the C ABI (progress = 2) on a twin
    for mode, tgt_n, dim in ((0, DiffableGaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 2), (2, RosenbrockND(3), 3)):
        init_n = init_with_seed(200, dim, 5) * 0.5
        s = NUTS(tgt_n, init_n, 0.8, mode=mode).set_seed(9)
        smp, st = s.run_progress(20, 15)
        t = NUTS(tgt_n, init_n, 0.8, mode=mode).set_seed(9)
        allr = np.empty((200, 35, dim), dtype=t.dtype)
        L.check(L.lib().mmcmc_nuts_run(t._h, 20, 15, allr.ctypes.data, 0, 2, None), "mmcmc_nuts_run")
        L.check(L.lib().mmcmc_nuts_sync(t._h), "sync")
        assert np.array_equal(smp, allr[:, 15:, :])
        plain = NUTS(tgt_n, init_n, 0.8, mode=mode).set_seed(9)._run(20, 15, True, "numpy")
        assert np.array_equal(smp, plain)                            # recording the burn-in changes no state
        rhat_o, p_o = O.chain_trackers_rhat(init_n.astype(np.float32), allr.astype(np.float32))
        rhat, mx, p = s.tracker.chain_stats()
        assert s.tracker.n == 35
        np.testing.assert_allclose(rhat, rhat_o, rtol=3e-5)
        assert abs(float(p) - float(p_o.astype(np.float64).mean())) < 1e-6
        ref = S.run_stats(smp)
        assert st.ess.min == ref.ess.min and st.rhat.mean == ref.rhat.mean


def test_progress_abi_symbols():
    import mini_mcmc_amd

    lib = mini_mcmc_amd.lib()
    for sym in ("mh_run_progress", "hmc_run_progress", "nuts_run_progress", "ess_from_chainstats", "tracker_within_var",
                "tracker_shape", "mh_run_rows", "hmc_run_rows", "mh_shape", "hmc_shape", "nuts_shape"):
        assert hasattr(lib, "mmcmc_" + sym)


@pytest.mark.gpu
def test_run_progress_at_baseline_sizes(O):
    """mmcmc_*_run_progress at BASELINE.json's chain counts: config 3 (65 536 x run_progress(400, 50)), config 2
    (65 536 x (1000, 100)) and config 5 (NUTS, 32-D f64, 65 536 x (100 + 200): the every-state buffer is 5 GB).  Size-independent
    properties: the sample equals the plain run's bit for bit, the trackers saw the right number of states, the callback's
    last report equals the tracker's final statistics, RunStats is RunStats::from(sample)."""
    import torch

    from mini_mcmc_amd import stats as S
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, GaussianND, IsotropicGaussian, RosenbrockND
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
    from mini_mcmc_amd.nuts import NUTS

    C = 65536
    seen = []
    h = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42)
    smp, st = h.run_progress(400, 50, to="torch", callback=lambda d, p, r: seen.append((d, p, r)))
    ref = HMC(RosenbrockND(3), init_with_seed(C, 3, 42, np.float32), 0.032, 10).set_seed(42).run(400, 50, to="torch")
    assert torch.equal(smp, ref) and h.tracker.n == 401 and len(seen) == 10 and seen[-1][0] == 400
    assert abs(seen[-1][2] - float(h.tracker.max_rhat())) <= 1e-6 * seen[-1][2]
    r0 = S.run_stats(smp)
    assert st.ess.min == r0.ess.min and st.rhat.max == r0.rhat.max
    del smp, ref

    m = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float32)).seed(42)
    smp, st = m.run_progress(1000, 100, to="torch")
    ref = MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_with_seed(C, 2, 42, np.float32)).seed(42).run(1000, 100, to="torch")
    assert torch.equal(smp, ref) and m.tracker.n == 1100
    _, mx, p = m.tracker.chain_stats()
    assert 0.4 < float(p) < 0.7 and 1.0 <= float(mx) < 1.02 and st.rhat.min > 0.98
    ess_cs = S.ess_from_chainstats(smp, m.tracker)
    assert np.all(ess_cs > 1e6)
    del smp, ref

    g = GaussianND.ill_conditioned(32, 1e4, 7)
    init = init_with_seed(C, 32, 42) * 0.1
    n = NUTS(g, init, 0.8, mode=2).set_seed(42).set_max_depth(10)
    smp, st = n.run_progress(100, 200, to="torch")
    ref = NUTS(g, init, 0.8, mode=2).set_seed(42).set_max_depth(10)._run(100, 200, True, "torch")
    assert torch.equal(smp, ref) and n.tracker.n == 300 and np.isfinite(st.ess.min)
