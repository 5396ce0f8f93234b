// Exercises include/mmcmc.hpp the way the reference's examples use the crate (examples/minimal_mh.rs,
// minimal_hmc.rs, minimal_nuts.rs).  Built and run by tests/test_cpp_facade.py; needs a GPU to go past the
// constructors, and checks the loud failure (MMCMC_ERR_NO_DEVICE) when there is none.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mmcmc.hpp"

#define REQUIRE(c)                                                                                                 \
    do {                                                                                                           \
        if (!(c)) {                                                                                                \
            std::printf("FAILED: %s (line %d)\n", #c, __LINE__);                                                   \
            return 1;                                                                                              \
        }                                                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    using namespace mmcmc;
    const bool expect_gpu = argc > 1 && std::atoi(argv[1]) != 0;
    auto init = init_det<double>(4, 2);
    REQUIRE(std::fabs(init[0] - 0.8343975468437959) < 1e-15 && std::fabs(init[7] - 0.41445339) < 1e-7);
    try {
        // examples/minimal_mh.rs
        MetropolisHastings<double> mh(Gaussian2D({0.0, 0.0}, {{{1.0, 0.0}, {0.0, 1.0}}}), IsotropicGaussian(1.0), init, 4);
        auto sample = mh.seed(42).run(1000, 100);
        REQUIRE(expect_gpu);
        REQUIRE(sample.size() == 4u * 1000u * 2u);
        double m0 = 0, v0 = 0;
        for (size_t i = 0; i < sample.size(); i += 2) m0 += sample[i];
        m0 /= 4000.0;
        for (size_t i = 0; i < sample.size(); i += 2) v0 += (sample[i] - m0) * (sample[i] - m0);
        v0 /= 3999.0;
        REQUIRE(std::fabs(m0) < 0.3 && std::fabs(v0 - 1.0) < 0.5); // metropolis_hastings.rs:379-380 tolerances
        // examples/minimal_hmc.rs
        auto init3 = init_det<float>(4, 3);
        HMC<float> hmc(RosenbrockND(3), init3, 4, 0.032f, 10);
        auto s3 = hmc.set_seed(1).run(400, 50);
        REQUIRE(s3.size() == 4u * 400u * 3u);
        for (float v : s3) REQUIRE(std::isfinite(v));
        auto pos = hmc.positions();
        REQUIRE(pos[0] == s3[399 * 3] && pos[11] == s3[3 * 400 * 3 + 399 * 3 + 2]);
        hmc.step();
        // examples/minimal_nuts.rs shape: 2-D Gaussian, 4 chains
        NUTS<double> nuts(DiffableGaussian2D({0.0, 1.0}, {{{4.0, 2.0}, {2.0, 3.0}}}), init, 4, 0.8);
        auto sn = nuts.set_seed(7).run(100, 50);
        REQUIRE(sn.size() == 4u * 100u * 2u);
        auto re = split_rhat_mean_ess(sn, 4, 100, 2);
        REQUIRE(re.first.size() == 2 && re.second[0] > 10.0f);
        // error behaviour: shape error instead of ndarray::ShapeError
        bool threw = false;
        try {
            MetropolisHastings<double> bad(RosenbrockND(3), IsotropicGaussian(1.0), init, 4); // dim 2 states, dim 3 target
            (void)bad;
        } catch (const Error &e) {
            threw = true;
        }
        (void)threw;
        // tests/metrohast_poisson_test.rs::test_poisson_mh: one chain from 0, seed 42, run(20000, 2000), histogram within 0.05
        DiscreteMetropolisHastings dmh(MMCMC_POISSON_REFLECT, {4.0}, {0});
        auto ks = dmh.seed(42).run(20000, 2000);
        double f4 = 0;
        for (int32_t k : ks) f4 += (k == 4);
        REQUIRE(std::fabs(f4 / 20000.0 - 0.1954) < 0.05 && dmh.accept_counts()[0] > 0);
        // NUTS in f64 on a dense 16-D Gaussian (A = tridiagonal 2, -0.5): the lane-group / MFMA kernel behind the facade
        {
            const int d = 16;
            std::vector<double> prec((size_t)d * d, 0.0);
            for (int i = 0; i < d; ++i) {
                prec[(size_t)i * d + i] = 2.0;
                if (i + 1 < d) prec[(size_t)i * d + i + 1] = prec[(size_t)(i + 1) * d + i] = -0.5;
            }
            auto init16 = init_with_seed<double>(64, d, 7);
            NUTS64 n64(GaussianND(d, prec), init16, 64, 0.8);
            REQUIRE(n64.kernel_variant() == 3);
            auto s64 = n64.set_seed(3).run(50, 50);
            REQUIRE(s64.size() == 64u * 50u * (size_t)d);
            double m = 0, v = 0;
            for (size_t i = 0; i < s64.size(); i += d) m += s64[i];
            m /= 3200.0;
            for (size_t i = 0; i < s64.size(); i += d) v += (s64[i] - m) * (s64[i] - m);
            v /= 3199.0;
            REQUIRE(std::fabs(m) < 0.15 && v > 0.3 && v < 0.9); // (A^-1)_00 ~ 0.53
        }
        // examples/mixture_gibbs.rs: 4 chains from init_det(4, 2), mixture (-2, 1, 3, 1.5, pi0 = 0.25)
        {
            GibbsMixtureSampler gs(-2.0, 1.0, 3.0, 1.5, 0.25, init);
            auto sg = gs.set_seed(11).run(1100, 1000);
            REQUIRE(sg.size() == 4u * 1100u * 2u);
            double mz = 0;
            for (size_t i = 1; i < sg.size(); i += 2) {
                REQUIRE(sg[i] == 0.0 || sg[i] == 1.0);
                mz += sg[i];
            }
            REQUIRE(std::fabs(mz / 4400.0 - 0.75) < 0.15);
        }
        // MultiChainTracker over the HMC sample (hmc.rs:242-281)
        MultiChainTracker tr(4, 3);
        tr.step(s3);
        REQUIRE(tr.max_rhat() > 0.9f && tr.p_accept() > 0.3f && tr.rhat().size() == 3);
        // run_progress behind the C ABI (hmc.rs:222-294, core.rs:208-360): the feeding order is the library's
        {
            auto init2 = init_with_seed<float>(64, 2, 5);
            HMC<float> hp(DiffableGaussian2D({0.0, 1.0}, {{{4.0, 2.0}, {2.0, 3.0}}}), init2, 64, 0.1f, 10);
            hp.set_seed(3);
            struct Seen { int calls = 0; uint64_t last = 0; } seen;
            auto cb = [](void *u, uint64_t done, uint64_t total, float p, float r) {
                auto *sn = (Seen *)u;
                sn->calls += 1;
                sn->last = done;
                (void)total; (void)p; (void)r;
            };
            auto r = hp.run_progress(50, 20, 10, cb, &seen);
            REQUIRE(r.sample.size() == 64u * 50u * 2u && seen.calls == 5 && seen.last == 70);
            REQUIRE(r.tracker->n() == 51 && r.tracker->max_rhat() >= 0.9f && r.stats.s.ess.min > 1.0f);
            MetropolisHastings<float> mp(Gaussian2D({0.0, 0.0}, {{{1.0, 0.0}, {0.0, 1.0}}}), IsotropicGaussian(1.0), init2, 64);
            auto rm = mp.seed(4).run_progress(40, 10);
            float mx = 0.f, pa = 0.f;
            REQUIRE(rm.tracker->n() == 50 && rm.tracker->chain_rhat(&mx, &pa).size() == 2 && pa > 0.2f && pa < 0.9f);
            auto ess = ess_from_chainstats(rm.sample, 64, 40, 2, *rm.tracker);
            REQUIRE(ess.size() == 2 && ess[0] > 1.0f);
        }
        // one call for every chain on a device group (here: two shards on device 0) == a single handle
        {
            auto init3 = init_with_seed<float>(130, 3, 42);
            HMC<float> one(RosenbrockND(3), init3, 130, 0.032f, 10);
            auto a = one.set_seed(7).run(30, 5);
            HMCGroup<float> grp(RosenbrockND(3), init3, 130, 0.032f, 10, {0, 0});
            auto b = grp.set_seed(7).run(30, 5);
            REQUIRE(a == b);
            auto re = grp.split_rhat_mean_ess();
            REQUIRE(re.first.size() == 3 && re.second[0] > 1.0f);
        }
        // the same for NUTS, and NUTS at a dimension without a compiled instance (RosenbrockND(11): run-time-D kernel)
        {
            auto init3 = init_with_seed<double>(90, 3, 42);
            for (auto &v : init3)
                v *= 0.5;
            NUTS<double> one(RosenbrockND(3), init3, 90, 0.8);
            NUTSGroup<double> grp(RosenbrockND(3), init3, 90, 0.8, {0, 0});
            REQUIRE(one.set_seed(7).run(12, 8) == grp.set_seed(7).run(12, 8));
            auto init11 = init_with_seed<double>(40, 11, 1);
            for (auto &v : init11)
                v *= 0.3;
            NUTS<double> n11(RosenbrockND(11), init11, 40, 0.8);
            auto s11 = n11.set_seed(2).run(6, 6);
            REQUIRE(s11.size() == 40u * 6u * 11u && std::isfinite(s11.back()));
        }
        // a target of the user's own, compiled at run time: a restatement of StandardNormal(2) must reproduce the built-in
        {
            const char *src =
                "template <class T> struct mmcmc_user_target { static constexpr int dim = 2;\n"
                "  MM_HD static T logp(const mm_tparams<T> &, const T *x) { T s = 0; for (int i = 0; i < 2; ++i) s = mm_fma(x[i], x[i], s); return T(-0.5) * s; }\n"
                "  MM_HD static T logp_grad(const mm_tparams<T> &, const T *x, T *g) { T s = 0; for (int i = 0; i < 2; ++i) { s = mm_fma(x[i], x[i], s); g[i] = -x[i]; } return T(-0.5) * s; } };\n";
            auto init2 = init_with_seed<double>(70, 2, 5);
            HMC<double> builtin(StandardNormal(2), init2, 70, 0.3, 5);
            HMC<double> user(UserTarget("std2", 2, src), init2, 70, 0.3, 5);
            REQUIRE(builtin.set_seed(3).run(20, 4) == user.set_seed(3).run(20, 4));
        }
        std::printf("facade ok (gpu)\n");
        return 0;
    } catch (const Error &e) {
        if (!expect_gpu && e.status == MMCMC_ERR_NO_DEVICE) {
            std::printf("facade ok (no gpu: %s)\n", e.what());
            return 0;
        }
        std::printf("unexpected error: %s\n", e.what());
        return 2;
    }
}
