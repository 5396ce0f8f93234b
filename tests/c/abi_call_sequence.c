/* The call sequence the Rust wrappers (rust/mini-mcmc-hip/src/lib.rs) make, in plain C99 against include/mmcmc.h:
 *   init_det -> GpuMetropolisHastings::new(..).seed(42).run(1000, 100)        (examples/minimal_mh.rs)
 *            -> GpuHmc::new(RosenbrockND, init, 0.032, 10).set_seed(42).run(400, 50), step(), positions()
 *            -> split_rhat_mean_ess(sample)
 *            -> GpuNuts::new(DiffableGaussian2D, [[0, 1]], 0.8).set_seed(42).run(1, 0) == the initial point (nuts.rs:1123-1136)
 *            -> a one-device group run
 * Exit code 0 and "abi sequence ok" on a GPU box; 77 ("no device") where there is none -- never a CPU fallback.
 *   gcc -std=c99 -Iinclude tests/c/abi_call_sequence.c -Lmini_mcmc_amd -lmmcmc -Wl,-rpath,$PWD/mini_mcmc_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mmcmc.h"

#define CHECK(expr)                                                                                               \
    do {                                                                                                          \
        int st_ = (expr);                                                                                         \
        if (st_ != MMCMC_OK) {                                                                                    \
            fprintf(stderr, "%s -> %d (%s)\n", #expr, st_, mmcmc_status_string(st_));                             \
            return st_ == MMCMC_ERR_NO_DEVICE ? 77 : 1;                                                           \
        }                                                                                                         \
    } while (0)

int main(void)
{
    int n_dev = 0;
    if (mmcmc_device_count(&n_dev) != MMCMC_OK || n_dev < 1) {
        printf("no device (status %s): the engine has no CPU fallback\n", mmcmc_status_string(MMCMC_ERR_NO_DEVICE));
        return 77;
    }
    /* init_det(4, 2): the reference's values (core.rs:404-435) */
    double init_d[8];
    CHECK(mmcmc_init_with_seed(4, 2, 42, init_d));
    if (fabs(init_d[0] - 0.8343975468437959) > 1e-15 || fabs(init_d[7] - 0.41445339) > 1e-7)
        return 2;

    /* minimal_mh.rs: Gaussian2D, IsotropicGaussian(1.0), 4 chains, run(1000, 100), f64 */
    mmcmc_target_desc g2 = {MMCMC_GAUSSIAN2D, 2, {0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0, 0}, NULL};
    mmcmc_proposal_desc prop = {MMCMC_PROPOSAL_ISOTROPIC_GAUSSIAN, 0, 1.0};
    mmcmc_mh *mh = NULL;
    CHECK(mmcmc_mh_create(&mh, &g2, &prop, init_d, 4, MMCMC_F64, 0));
    CHECK(mmcmc_mh_seed(mh, 42));
    double *mh_out = (double *)malloc(sizeof(double) * 4 * 1000 * 2);
    uint64_t acc[4];
    CHECK(mmcmc_mh_run(mh, 1000, 100, mh_out, 0, acc, NULL));
    double mean = 0;
    for (int i = 0; i < 4 * 1000 * 2; ++i)
        mean += mh_out[i];
    mean /= 4 * 1000 * 2;
    if (!(fabs(mean) < 0.3) || acc[0] == 0 || acc[0] > 1100)
        return 3;
    CHECK(mmcmc_mh_destroy(mh));
    free(mh_out);

    /* minimal_hmc.rs: RosenbrockND(3), eps 0.032, L 10, 400 + 50, f32, 256 chains */
    enum { C = 256, NC = 400, D = 3 };
    double *init3 = (double *)malloc(sizeof(double) * C * D);
    float *init3f = (float *)malloc(sizeof(float) * C * D);
    CHECK(mmcmc_init_with_seed(C, D, 42, init3));
    for (int i = 0; i < C * D; ++i)
        init3f[i] = (float)init3[i];
    mmcmc_target_desc ros = {MMCMC_ROSENBROCK_ND, D, {0}, NULL};
    mmcmc_hmc *hmc = NULL;
    CHECK(mmcmc_hmc_create(&hmc, &ros, init3f, C, 0.032, 10, MMCMC_F32, 0));
    CHECK(mmcmc_hmc_seed(hmc, 42));
    float *sample = (float *)malloc(sizeof(float) * C * NC * D);
    uint64_t *hacc = (uint64_t *)malloc(sizeof(uint64_t) * C);
    CHECK(mmcmc_hmc_run(hmc, NC, 50, sample, 0, hacc, NULL));
    double rate = 0;
    for (int c = 0; c < C; ++c)
        rate += (double)hacc[c];
    rate /= (double)C * (NC + 50);
    if (!(rate > 0.3 && rate <= 1.0))
        return 9;
    float pos[C * D];
    CHECK(mmcmc_hmc_state(hmc, pos));
    if (memcmp(pos, sample + ((size_t)0 * NC + NC - 1) * D, sizeof(float) * D) != 0) /* current state == last row */
        return 4;
    CHECK(mmcmc_hmc_step(hmc, NULL));
    CHECK(mmcmc_hmc_sync(hmc));
    float rhat[D], ess[D];
    CHECK(mmcmc_split_rhat_mean_ess(sample, 0, MMCMC_F32, C, NC, D, rhat, ess, 0, NULL));
    for (int d = 0; d < D; ++d)
        if (!(rhat[d] > 0.f && rhat[d] < 1.5f && ess[d] > 1.f))
            return 5;
    CHECK(mmcmc_hmc_destroy(hmc));

    /* the same chains through a one-device group: identical sample */
    mmcmc_hmc_group *grp = NULL;
    int dev0 = 0, used = -1;
    CHECK(mmcmc_hmc_group_create(&grp, &ros, init3f, C, 0.032, 10, MMCMC_F32, &dev0, 1));
    CHECK(mmcmc_hmc_group_seed(grp, 42));
    float *sample_g = (float *)malloc(sizeof(float) * C * NC * D);
    CHECK(mmcmc_hmc_group_run(grp, NC, 50, sample_g, NULL));
    if (memcmp(sample, sample_g, sizeof(float) * C * NC * D) != 0)
        return 6;
    float rhat_g[D], ess_g[D];
    CHECK(mmcmc_hmc_group_split_rhat_mean_ess(grp, rhat_g, ess_g, &used));
    for (int d = 0; d < D; ++d)
        if (fabsf(rhat_g[d] - rhat[d]) > 1e-5f * rhat[d] || fabsf(ess_g[d] - ess[d]) > 1e-3f * ess[d])
            return 7;
    CHECK(mmcmc_hmc_group_destroy(grp));

    /* test_chain_1 (nuts.rs:1123-1136): run(1, 0) returns the initial position */
    mmcmc_target_desc dg = {MMCMC_DIFFABLE_GAUSSIAN2D, 2, {0.0, 1.0, 4.0, 2.0, 2.0, 3.0, 0, 0}, NULL};
    double nuts_init[2] = {0.0, 1.0};
    mmcmc_nuts *nuts = NULL;
    CHECK(mmcmc_nuts_create(&nuts, &dg, nuts_init, 1, 0.8, 0, 0));
    CHECK(mmcmc_nuts_seed(nuts, 42));
    float row[2];
    CHECK(mmcmc_nuts_run(nuts, 1, 0, row, 0, 0, NULL));
    CHECK(mmcmc_nuts_sync(nuts));
    if (row[0] != 0.0f || row[1] != 1.0f)
        return 8;
    /* NUTS::run_progress (nuts.rs:172-345) through the C ABI: sample + RunStats + the per-chain trackers, then
     * stats::ess_from_chainstats (stats.rs:668-671) with them */
    {
        enum { CN = 64, NN = 40 };
        double ninit[CN * 2];
        CHECK(mmcmc_init_with_seed(CN, 2, 3, ninit));
        mmcmc_nuts *np = NULL;
        CHECK(mmcmc_nuts_create(&np, &dg, ninit, CN, 0.8, 0, 0));
        CHECK(mmcmc_nuts_seed(np, 1));
        float *ns = (float *)malloc(sizeof(float) * CN * NN * 2);
        mmcmc_run_stats rs;
        mmcmc_tracker *tr = NULL;
        CHECK(mmcmc_nuts_run_progress(np, NN, 25, 0, NULL, NULL, ns, 0, &rs, &tr, NULL));
        uint64_t tn = 0;
        CHECK(mmcmc_tracker_n(tr, &tn));
        float ess_cs[2], within[2], var[2];
        CHECK(mmcmc_tracker_within_var(tr, within, var, NULL));
        CHECK(mmcmc_ess_from_chainstats(ns, 0, MMCMC_F32, CN, NN, 2, tr, ess_cs, 0, NULL));
        if (tn != NN + 25 || !(rs.ess.min > 1.f) || !(ess_cs[0] > 1.f && ess_cs[1] > 1.f) || !(within[0] > 0.f && var[0] > 0.f))
            return 10;
        CHECK(mmcmc_tracker_destroy(tr));
        CHECK(mmcmc_nuts_destroy(np));
        free(ns);
    }
    CHECK(mmcmc_nuts_destroy(nuts));
    free(init3);
    free(init3f);
    free(sample);
    free(sample_g);
    free(hacc);
    printf("abi sequence ok (HMC accept rate %.3f, used_rccl %d)\n", rate, used);
    return 0;
}
