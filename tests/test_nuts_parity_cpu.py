"""NUTS, level B (CPU): the engine's ITERATIVE transition (mini_mcmc_amd/csrc/mm_nuts.h: explicit pending-subtree stack,
engine arithmetic, compiled for the host) against the RECURSIVE restatement of nuts.rs (oracle/nuts_impl.inc) that is
pinned to the reference's seeded known-answer vectors (tests/test_oracle_pins.py) -- both on the engine's stream.
Equal leapfrog counts mean equal tree shapes, i.e. the same sequence of doublings, merges and early exits."""
import numpy as np
import pytest

GAUSS = [0.0, 1.0, 4.0, 2.0, 2.0, 3.0]


def _cases(O):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((4, 4))
    A = A @ A.T / 4 + np.eye(4)
    return [
        (O.DIFFABLE_GAUSSIAN2D, 2, GAUSS, None, lambda f32: O.diffable_gaussian2d([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]], t_is_f32=f32)),
        (O.ROSENBROCK2D, 2, [1.0, 100.0], None, lambda f32: O.rosenbrock2d(1.0, 100.0)),
        (O.STANDARD_NORMAL, 3, [], None, lambda f32: O.standard_normal(3)),
        (O.ROSENBROCK_ND, 3, [], None, lambda f32: O.rosenbrock_nd(3)),
        (O.GAUSSIAN_ND, 4, [], A, lambda f32: O.gaussian_nd(A)),
    ]


def test_iterative_tree_equals_recursive_tree_f64(O):
    for kind, dim, params, mat, mk in _cases(O):
        init = O.init_with_seed(6, dim, 17) * 0.8
        for progress in (False, True):
            out_e, pos_e, ad_e, nlf = O.engine_host_nuts_run(2, kind, dim, params, init, 0.8, 25, 15, seed=5, matrix=mat,
                                                             progress=progress, n_threads=1)
            s = O.NUTS(mk(False), init, 0.8, mode=2).use_engine_stream(5).set_max_depth(10)
            out_o = s.run(25, 15, progress=progress, n_threads=1)
            lf_o = np.array([s.chain_state(i)["n_leapfrog_total"] for i in range(6)])
            assert np.array_equal(nlf.astype(np.int64), lf_o.astype(np.int64)), (kind, dim, progress)
            # stiff targets (Rosenbrock) amplify the 1e-16 rounding differences of fused vs unfused arithmetic
            np.testing.assert_allclose(out_e, out_o, rtol=1e-6, atol=1e-6, err_msg=f"kind {kind} D {dim}")
            eps_o = np.array([s.chain_state(i)["epsilon"] for i in range(6)])
            np.testing.assert_allclose(ad_e[:, 0], eps_o, rtol=1e-6)


def test_grouped_reduction_twin_keeps_the_tree_shapes(O):
    """engine_host mode 3 = mm_nuts_step with the dot products summed as four interleaved partial sums (the order the
    lane-group / MFMA kernel mm_nuts_lg.h produces; it is that kernel's bit-exact host twin).  Against the pinned
    recursive restatement on the config-5 shaped target (32-D / 16-D dense Gaussian, f64): same tree shapes, samples
    equal up to the rounding drift a stiff trajectory amplifies."""
    for dim, cond in ((32, 1e4), (16, 100.0)):
        rng = np.random.default_rng(7)
        q, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
        A = (q * np.logspace(0.0, np.log10(cond), dim)) @ q.T
        A = (A + A.T) / 2.0
        init = O.init_with_seed(6, dim, 42) * 0.1
        out_g, _, ad_g, nlf_g = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, dim, [], init, 0.8, 10, 20, seed=42, matrix=A, n_threads=2)
        out_s, _, _, nlf_s = O.engine_host_nuts_run(2, O.GAUSSIAN_ND, dim, [], init, 0.8, 10, 20, seed=42, matrix=A, n_threads=2)
        s = O.NUTS(O.gaussian_nd(A), init, 0.8, mode=2).use_engine_stream(42).set_max_depth(10)
        out_o = s.run(10, 20, n_threads=2)
        lf_o = np.array([s.chain_state(i)["n_leapfrog_total"] for i in range(6)])
        assert np.array_equal(nlf_g, nlf_s) and np.array_equal(nlf_g.astype(np.int64), lf_o.astype(np.int64))
        assert not np.array_equal(out_g, out_s)  # a different summation order, really
        # ~6000 leapfrog steps of a cond-1e4 system amplify the 1e-16 differences; the shapes above are the check
        np.testing.assert_allclose(out_g, out_s, atol=2e-2)
        np.testing.assert_allclose(out_g, out_o, atol=2e-2)
        # early in the run (the first transitions from the initial point) the three agree tightly
        e_g, _, _, l_g = O.engine_host_nuts_run(3, O.GAUSSIAN_ND, dim, [], init, 0.8, 4, 0, seed=42, matrix=A, n_threads=2)
        e_s, _, _, l_s = O.engine_host_nuts_run(2, O.GAUSSIAN_ND, dim, [], init, 0.8, 4, 0, seed=42, matrix=A, n_threads=2)
        assert np.array_equal(l_g, l_s)
        np.testing.assert_allclose(e_g, e_s, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("mode", [0, 1])
def test_first_transitions_agree_in_f32_modes(O, mode):
    # f32 tensors: compare the first few transitions (before rounding drift can flip a tree decision)
    for kind, dim, params, mat, mk in _cases(O):
        init = O.init_with_seed(8, dim, 23) * 0.8
        out_e, _, ad_e, nlf = O.engine_host_nuts_run(mode, kind, dim, params, init, 0.8, 4, 0, seed=9, matrix=mat, n_threads=1)
        s = O.NUTS(mk(mode == 1), init, 0.8, mode=mode).use_engine_stream(9).set_max_depth(10)
        out_o = s.run(4, 0, n_threads=1)
        lf_o = np.array([s.chain_state(i)["n_leapfrog_total"] for i in range(8)])
        assert np.mean(nlf.astype(np.int64) == lf_o.astype(np.int64)) >= 0.75, (kind, dim)
        same = nlf.astype(np.int64) == lf_o.astype(np.int64)
        np.testing.assert_allclose(out_e[same], out_o[same], rtol=5e-4, atol=5e-4)
        assert np.array_equal(out_e[:, 0].astype(np.float64), init.astype(np.float32).astype(np.float64))  # row 0 = initial point


def test_reference_off_by_one_and_epsilon_search(O, kats):
    # test_chain_1 (nuts.rs:1123-1136): run(1, 0) takes no step and returns the initial position; the epsilon search
    # (test_find_reasonable_epsilon, nuts.rs:1050-1055) is RNG-free, so the engine reproduces it exactly
    out, pos, ad, nlf = O.engine_host_nuts_run(0, O.DIFFABLE_GAUSSIAN2D, 2, GAUSS, [[0.0, 1.0]], 0.8, 1, 0, seed=42)
    assert np.array_equal(out.ravel(), [0.0, 1.0]) and nlf[0] == 0
    # the epsilon found depends on the momentum drawn in init_chain (engine stream here), always a power of two
    assert ad[0, 0] == 2.0 ** round(np.log2(ad[0, 0])) and ad[0, 3] == pytest.approx(np.log(10 * ad[0, 0]))


def test_depth_cap_is_honoured_and_matches_oracle_cap(O):
    init = O.init_with_seed(4, 3, 1) * 0.5
    for cap in (1, 3):
        out_e, _, _, nlf = O.engine_host_nuts_run(2, O.STANDARD_NORMAL, 3, [], init, 0.8, 10, 10, seed=2, max_depth=cap, n_threads=1)
        s = O.NUTS(O.standard_normal(3), init, 0.8, mode=2).use_engine_stream(2).set_max_depth(cap)
        out_o = s.run(10, 10, n_threads=1)
        np.testing.assert_allclose(out_e, out_o, rtol=1e-9, atol=1e-9)
        assert nlf.max() <= 19 * (2**cap - 1)


def test_engine_nuts_posterior_moments(O, kats):
    # level D on the CPU: Gaussian2D of the reference's NUTS tests, moments within the MH tolerances
    init = O.init_with_seed(64, 2, 3)
    out, _, ad, _ = O.engine_host_nuts_run(0, O.DIFFABLE_GAUSSIAN2D, 2, GAUSS, init, 0.8, 400, 200, seed=11, progress=True)
    flat = out.reshape(-1, 2).astype(np.float64)
    assert np.all(np.abs(flat.mean(axis=0) - [0.0, 1.0]) < 0.1)
    assert np.all(np.abs(np.cov(flat.T) - np.array([[4.0, 2.0], [2.0, 3.0]])) < 0.25)
    assert np.all(ad[:, 0] == ad[:, 1])  # after warm-up epsilon is frozen at epsilon_bar (nuts.rs:688-690)


def test_leaf_pairs_walk_the_same_trees_bit_for_bit(O):
    """mm_nuts_step_pairs (leaves in pairs, level 0 of the pending-subtree stack in registers, pushes folded into the step
    that precedes them: the order in which mm_nuts_pair_kernel walks a tree) against mm_nuts_step, host build, every type
    mode: samples, positions, adaptation state and leapfrog counts equal bit for bit -- including depth-capped trees and
    a stiff target whose trees are cut short by divergences."""
    E = O.engine_host_lib()
    for kind, dim, params, mat, _ in _cases(O):
        for mode in (0, 1, 2):
            for max_depth, scale in ((10, 0.8), (3, 0.8), (10, 6.0)):
                init = O.init_with_seed(24, dim, 23) * scale
                res = []
                try:
                    for pairs in (0, 1):
                        E.eh_nuts_set_pairs(pairs)
                        res.append(O.engine_host_nuts_run(mode, kind, dim, params, init, 0.8, 12, 10, seed=9, matrix=mat,
                                                          max_depth=max_depth, n_threads=2))
                finally:
                    E.eh_nuts_set_pairs(0)
                for a, b in zip(res[0], res[1]):
                    assert np.array_equal(a, b), (kind, dim, mode, max_depth, scale)
                assert res[0][3].sum() > 0
