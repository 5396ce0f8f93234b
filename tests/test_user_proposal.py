"""User-defined proposals (csrc/mm_rtc.hip, mmcmc_proposal_register_source): the GPU analogue of implementing the
reference's open `Proposal` trait (distributions.rs:92-101).  With a proposal of the caller's own the MH transition keeps
BOTH q-terms of the acceptance ratio, in the reference's order (metropolis_hastings.rs:303-315):
    log_accept_ratio = (lp(x') + log q(x | x')) - (lp(x) + log q(x' | x)).
Checked here: a user restatement of the built-in IsotropicGaussian random walk (symmetric: the q-terms cancel up to the
rounding of a zero difference) against the built-in kernels; an ASYMMETRIC proposal (multiplicative log-normal random
walk) against oracle/mh_hmc.c, which restates MHMarkovChain::step op for op with the same custom proposal; its
posterior; the error paths."""
import ctypes as C

import numpy as np
import pytest

# IsotropicGaussian::sample / ::logp (distributions.rs:364-386) as a user proposal
ISOTROPIC = r"""
template <class T> struct mmcmc_user_proposal {
    MM_HD static void sample(T sigma, const T *x, const T *z, T *out) {
        for (int i = 0; i < MM_USER_DIM; ++i)
            out[i] = mm_fma(sigma, z[i], x[i]);
    }
    MM_HD static T logp(T sigma, const T *from, const T *to) {
        const T var = sigma * sigma;
        T acc = 0;
        for (int i = 0; i < MM_USER_DIM; ++i) {
            const T d = to[i] - from[i];
            acc += -(d * d) / (T(2) * var);
        }
        return acc;
    }
};
"""

# the asymmetric custom proposal oracle/mh_hmc.c restates (prop_kind 1), operation for operation
LOGNORMAL = r"""
template <class T> struct mmcmc_user_proposal {
    MM_HD static void sample(T sigma, const T *x, const T *z, T *out) {
        for (int i = 0; i < MM_USER_DIM; ++i)
            out[i] = x[i] * mm_expT(sigma * z[i]);
    }
    MM_HD static T logp(T sigma, const T *from, const T *to) {
        const T two = T(2), var = sigma * sigma;
        T acc = 0;
        for (int i = 0; i < MM_USER_DIM; ++i) {
            const T lt = mm_logT(to[i]), lf = mm_logT(from[i]);
            const T dd = lt - lf;
            acc += -lt - (dd * dd) / (two * var);
        }
        return acc - T(MM_USER_DIM) * (mm_logT(sigma) + T(0.91893853320467274178));
    }
};
"""

GAMMA2 = r"""
template <class T> struct mmcmc_user_target { /* independent Gamma(k_i, 1): logp = sum (k_i - 1) ln x_i - x_i, x > 0 */
    static constexpr int dim = 2;
    MM_HD static T logp(const mm_tparams<T> &P, const T *x) {
        T acc = 0;
        for (int i = 0; i < dim; ++i)
            acc += (P.p[i] - T(1)) * mm_logT(x[i]) - x[i];
        return acc;
    }
    MM_HD static T logp_grad(const mm_tparams<T> &P, const T *x, T *g) {
        for (int i = 0; i < dim; ++i)
            g[i] = (P.p[i] - T(1)) / x[i] - T(1);
        return logp(P, x);
    }
};
"""


def test_register_proposal_without_device_reports_no_device():
    import torch

    import mini_mcmc_amd
    from mini_mcmc_amd import _lib as L

    lib = mini_mcmc_amd.lib()
    if not torch.cuda.is_available():
        kind = C.c_int()
        assert lib.mmcmc_proposal_register_source(b"iso", L.GAUSSIAN2D, 2, ISOTROPIC.encode(), C.byref(kind), None, 0) == L.ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_symmetric_user_proposal_follows_the_builtin_random_walk(O, dtype):
    """The isotropic random walk as a user proposal: same proposals from the same noise as the built-in kernels; the accept
    test now carries q_b - q_f = -(d^2)/(2 var) + (d^2)/(2 var), an exact zero (the two squares are of opposite numbers),
    inside (lp' + q_b) - (lp + q_f): decisions and states are the built-in kernels' except where that regrouping rounds a
    ratio across ln u -- never over these runs."""
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND, UserProposal
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    for tgt, std in ((Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]]), 1.0), (RosenbrockND(3), 0.1)):
        prop = UserProposal("iso_rw", tgt, ISOTROPIC, std)
        assert prop.kind >= 1000
        init = init_with_seed(777, tgt.dim, 42, dtype)
        a = MetropolisHastings(tgt, prop, init).seed(5)
        b = MetropolisHastings(tgt, IsotropicGaussian(std), init).seed(5)
        out_a, out_b = a.run(60, 7), b.run(60, 7)
        same = (out_a == out_b).all(axis=(1, 2))
        assert same.mean() > 0.999, (type(tgt).__name__, dtype, same.mean())
        assert np.abs(a.accept_counts.astype(np.int64) - b.accept_counts.astype(np.int64)).max() <= (0 if same.all() else 67)
        # a continued handle and a second run partition give the same chain (the stream is keyed by chain and iteration)
        c = MetropolisHastings(tgt, prop, init).seed(5)
        parts = np.concatenate([c.run(30, 7), c.run(30, 0)], axis=1)
        assert np.array_equal(parts, out_a)


@pytest.mark.gpu
def test_asymmetric_user_proposal_vs_reference_order_oracle(O):
    """The multiplicative log-normal random walk (asymmetric: q(x'|x) != q(x|x')) through hipRTC against oracle/mh_hmc.c,
    which runs MHMarkovChain::step (metropolis_hastings.rs:303-315) with the same custom proposal on the engine's stream:
    f64 -- identical accept decisions, states to rounding (the target's Sigma^-1 is precomputed on the GPU and re-derived
    per call in the oracle, Q5); f32 -- first transition to f32 rounding."""
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, RosenbrockND, UserProposal
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    cases = [(Gaussian2D([2.0, 3.0], [[4.0, 2.0], [2.0, 3.0]]), O.gaussian2d([2.0, 3.0], [[4.0, 2.0], [2.0, 3.0]]), 0.4),
             (RosenbrockND(3), O.rosenbrock_nd(3), 0.05)]
    for tgt, otgt, std in cases:
        prop = UserProposal("lognormal_rw", tgt, LOGNORMAL, std)
        init = np.abs(init_with_seed(256, tgt.dim, 6, np.float64)) + 0.5  # the walk lives on the positive orthant
        s = MetropolisHastings(tgt, prop, init).seed(3)
        o = O.MetropolisHastings(otgt, std, init, np.float64).set_proposal(1).use_engine_stream(3)
        g, ref = s.run(40, 5), o.run(40, 5, n_threads=2)
        name = type(tgt).__name__
        assert np.array_equal(s.accept_counts, o.accept_counts), name
        assert 0.05 < s.accept_counts.mean() / 45 < 0.95, name
        np.testing.assert_allclose(g, ref, rtol=1e-9, atol=1e-9, err_msg=name)
        assert (g > 0).all()
        # f32: one transition from the same start
        init32 = init.astype(np.float32)
        s32 = MetropolisHastings(tgt, prop, init32).seed(3)
        o32 = O.MetropolisHastings(otgt, std, init32, np.float32).set_proposal(1).use_engine_stream(3)
        g32, r32 = s32.run(1, 0), o32.run(1, 0, n_threads=2)
        agree = s32.accept_counts == o32.accept_counts
        assert agree.mean() > 0.98, name
        np.testing.assert_allclose(g32[agree], r32[agree], rtol=2e-5, atol=2e-5, err_msg=name)


@pytest.mark.gpu
def test_asymmetric_proposal_on_a_user_target_samples_its_posterior():
    """User proposal over a user target: independent Gamma(k, 1) on the positive quadrant under the log-normal walk.  A
    sampler that dropped the q-terms would converge to the density times x (the walk's Jacobian), i.e. Gamma(k + 1, 1):
    means off by one.  65 536 chains x 200 draws after 300: mean and variance = k to 2 %."""
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import UserProposal, UserTarget
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    k = [2.0, 5.0]
    tgt = UserTarget("gamma2", 2, GAMMA2, params=k)
    prop = UserProposal("lognormal_rw_gamma", tgt, LOGNORMAL, 0.8)
    init = np.abs(init_with_seed(65536, 2, 11, np.float32)) + 0.5
    s = MetropolisHastings(tgt, prop, init).seed(21)
    out = s.run(200, 300).reshape(-1, 2).astype(np.float64)
    np.testing.assert_allclose(out.mean(axis=0), k, rtol=2e-2)
    np.testing.assert_allclose(out.var(axis=0), k, rtol=4e-2)


@pytest.mark.gpu
def test_user_proposal_error_paths():
    from mini_mcmc_amd import _lib as L
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, RosenbrockND, UserProposal
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    g2 = Gaussian2D([0.0, 1.0], [[4.0, 2.0], [2.0, 3.0]])
    with pytest.raises(L.MmcmcError) as e:  # does not compile: the compiler's log comes back
        UserProposal("broken", g2, "template <class T> struct mmcmc_user_proposal { oops };")
    assert "error" in str(e.value).lower()
    prop = UserProposal("iso_rw_g2", g2, ISOTROPIC, 1.0)
    init3 = init_with_seed(64, 3, 1, np.float32)
    with pytest.raises(Exception):  # compiled for Gaussian2D: another target kind is refused
        MetropolisHastings(RosenbrockND(3), prop, init3)
    # a model kind is not a target: it cannot be handed to the other samplers
    class Fake(RosenbrockND):
        pass
    fake = Fake(2)
    fake.kind = prop.kind
    with pytest.raises(Exception):
        HMC(fake, init_with_seed(64, 2, 1, np.float32), 0.1, 3)


# ---- integer states: the reference's own custom Target + Proposal pairs as user source -------------------------------
# tests/metrohast_poisson_test.rs:18-85: PoissonDist + PoissonRandomWalk
POISSON_REFLECT = r"""
struct mmcmc_user_discrete {
    MM_HD static double logp(const double *p, int k) {
        if (k < 0) return MM_NEG_INF;
        const double kf = (double)k;
        return kf * mm_log(p[0]) - p[0] - mm_ln_factorial(k);      /* :31 */
    }
    MM_HD static int sample(const double *, int current, int step) {
        const int n = current + step;
        return n < 0 ? 0 : n;                                       /* "reflect instead of going negative" :70-76 */
    }
    MM_HD static double logq(const double *, int, int) { return mm_log(0.5); }   /* :79-82 */
};
"""
# tests/metrohast_poisson_test.rs:150-212: BinomialDist + BinomialRandomWalk (p[0] = n, p[1] = p)
BINOMIAL_CLAMP = r"""
struct mmcmc_user_discrete {
    MM_HD static double logp(const double *p, int k) {
        const int n = (int)p[0];
        if (k < 0 || k > n) return MM_NEG_INF;
        const double kf = (double)k, nf = (double)n;
        const double bc = mm_ln_factorial(n) - mm_ln_factorial(k) - mm_ln_factorial(n - k);
        return bc + kf * mm_log(p[1]) + (nf - kf) * mm_log(1.0 - p[1]);
    }
    MM_HD static int sample(const double *p, int current, int step) {
        const int n = (int)p[0], v = current + step;
        return v < 0 ? 0 : (v > n ? n : v);
    }
    MM_HD static double logq(const double *, int, int) { return mm_log(0.5); }
};
"""
# examples/poisson_mh.rs:8-76: PoissonTarget + NonnegativeProposal (asymmetric at 0)
POISSON_NONNEG = r"""
struct mmcmc_user_discrete {
    MM_HD static double logp(const double *p, int k) {
        if (k < 0) return MM_NEG_INF;
        const double kf = (double)k;
        return -p[0] + kf * mm_log(p[0]) - mm_ln_factorial(k);      /* poisson_mh.rs:25 */
    }
    MM_HD static int sample(const double *, int current, int step) { return current == 0 ? 1 : current + step; }
    MM_HD static double logq(const double *, int x, int y) {
        if (x == 0) return y == 1 ? 0.0 : MM_NEG_INF;
        return (y == x + 1 || y + 1 == x) ? mm_log(0.5) : MM_NEG_INF;
    }
};
"""
# a model the library does not have: geometric pmf (1 - r) r^k on k >= 0 under a walk that reflects at 0
GEOMETRIC = r"""
struct mmcmc_user_discrete {
    MM_HD static double logp(const double *p, int k) { return k < 0 ? MM_NEG_INF : (double)k * mm_log(p[0]); }
    MM_HD static int sample(const double *, int current, int step) { const int n = current + step; return n < 0 ? 0 : n; }
    MM_HD static double logq(const double *, int, int) { return mm_log(0.5); }
};
"""


@pytest.mark.gpu
def test_user_restatements_of_the_discrete_models_are_bit_identical():
    """The reference's three hand-written Target + Proposal pairs as user source against the built-in integer-state models
    (whose log-densities are tables filled on the host in the reference's operation order): samples, final states and
    accept counts equal bit for bit -- ragged chain counts, a chain offset, a continued handle."""
    from mini_mcmc_amd.discrete import (BinomialClamp, DiscreteMetropolisHastings, PoissonNonneg, PoissonReflect,
                                        UserDiscreteModel)

    cases = [(PoissonReflect(4.0), UserDiscreteModel("poisson_reflect", POISSON_REFLECT, [4.0])),
             (BinomialClamp(10, 0.3), UserDiscreteModel("binomial_clamp", BINOMIAL_CLAMP, [10.0, 0.3])),
             (PoissonNonneg(4.0), UserDiscreteModel("poisson_nonneg", POISSON_NONNEG, [4.0]))]
    rng = np.random.default_rng(4)
    for builtin, user in cases:
        assert user.kind >= 1000
        for C_, off in ((5000, 0), (130, 1 << 33)):  # four-waves-per-SIMD kernel / one-wave kernel of the built-in model
            init = rng.integers(0, 9, size=C_).astype(np.int32)
            a = DiscreteMetropolisHastings(builtin, init).seed(11).set_chain_offset(off)
            b = DiscreteMetropolisHastings(user, init).seed(11).set_chain_offset(off)
            for nc, nd in ((70, 13), (25, 0)):
                assert np.array_equal(a.run(nc, nd), b.run(nc, nd)), (user.name, C_, nc)
            assert np.array_equal(a.state(), b.state()) and np.array_equal(a.accept_counts(), b.accept_counts()), user.name


@pytest.mark.gpu
def test_new_discrete_model_samples_its_pmf():
    from mini_mcmc_amd.discrete import DiscreteMetropolisHastings, UserDiscreteModel

    r = 0.6
    m = UserDiscreteModel("geometric", GEOMETRIC, [r])
    s = DiscreteMetropolisHastings(m, np.zeros(65536, dtype=np.int32)).seed(3)
    out = s.run(100, 400).reshape(-1)
    k = np.arange(12)
    emp = np.array([(out == i).mean() for i in k])
    np.testing.assert_allclose(emp, (1 - r) * r**k, atol=2e-3)
    with pytest.raises(Exception):  # an integer-state model is not a continuous target
        from mini_mcmc_amd.distributions import IsotropicGaussian, RosenbrockND
        from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

        fake = RosenbrockND(1)
        fake.kind = m.kind
        MetropolisHastings(fake, IsotropicGaussian(1.0), np.zeros((8, 1), dtype=np.float32))
