// Event counters of the leaf-pair NUTS kernel (mm_nuts_pair_kernel, -DMM_NUTS_PROFILE) on RosenbrockND(3), f32 tensors /
// f64 scalars, 65 536 chains, 100 + 100 transitions: ticks, lanes served per section, Philox blocks, free-step trips.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_NUTS_PROFILE tools/nuts_pair_probe.hip -o /tmp/npp && /tmp/npp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../mini_mcmc_amd/csrc/mm_nuts_kernels.h"

int main()
{
    using TT = float;
    using ST = double;
    using Tgt = mm_target<TT, MM_ROSENBROCK_ND, 3>;
    const unsigned long long C = 65536;
    std::vector<TT> x(C * 3);
    std::mt19937_64 g(42);
    std::normal_distribution<double> nd(0.0, 0.5);
    for (auto &v : x)
        v = (TT)nd(g);
    std::vector<mm_nuts_adapt<ST>> ad(C);
    for (auto &q : ad) {
        q.epsilon = -1;
        q.epsilon_bar = 1;
        q.h_bar = 0;
        q.mu = std::log(10.0);
    }
    TT *d_x;
    mm_nuts_adapt<ST> *d_ad;
    unsigned long long *d_nlf;
    unsigned int *d_hist;
    (void)hipMalloc(&d_x, x.size() * sizeof(TT));
    (void)hipMalloc(&d_ad, C * sizeof(mm_nuts_adapt<ST>));
    (void)hipMalloc(&d_nlf, C * 8);
    (void)hipMalloc(&d_hist, 64 * 4);
    (void)hipMemcpy(d_x, x.data(), x.size() * sizeof(TT), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_ad, ad.data(), C * sizeof(mm_nuts_adapt<ST>), hipMemcpyHostToDevice);
    (void)hipMemset(d_nlf, 0, C * 8);
    (void)hipMemset(d_hist, 0, 64 * 4);
    mm_tparams<TT> P = {};
    (void)mm_launch_nuts_init<TT, ST, Tgt>(P, d_x, d_ad, C, 1, 0, 0);
    mm_nuts_args<TT, ST> a = {};
    a.P = P;
    a.state = d_x;
    a.adapt = d_ad;
    a.out = nullptr;
    a.n_leapfrog = d_nlf;
    a.depth_hist = d_hist;
    a.n_chains = C;
    a.seed = 1;
    a.n_total = 100;
    a.m0 = 0;
    a.n_pre = 100;
    a.n_rec = 100;
    a.n_discard = 100;
    a.max_depth = 10;
    a.target_accept_p = 0.8;
    a.stack_in_lds = 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    (void)mm_launch_nuts_run_pair<TT, ST, Tgt>(a, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(C);
    (void)hipMemcpy(h.data(), d_nlf, C * 8, hipMemcpyDeviceToHost);
    const char *names[16] = {"ticks", "lane-ticks waiting between transitions", "lane-ticks alive", "transition sections",
                             "lanes served by them", "Philox blocks for the ring", "lanes that took one", "doublings begun",
                             "first leaves", "second leaves", "merges at level >= 1", "free-step trips", "free steps",
                             "doublings ended", "", ""};
    double tot[16] = {};
    for (unsigned long long w = 0; w < C / 64; ++w)
        for (int i = 0; i < 16; ++i)
            tot[i] += (double)h[w * 64 + i];
    const double waves = (double)(C / 64);
    printf("kernel %.2f ms\n", ms);
    for (int i = 0; i < 14; ++i)
        printf("%-44s per wave %12.0f   per tick %8.3f\n", names[i], tot[i] / waves, tot[i] / tot[0]);
    printf("leaves per wave-lane %.0f, ticks per leaf %.3f\n", (tot[8] + tot[9]) / waves / 64.0, tot[0] / ((tot[8] + tot[9]) / 64.0));
    return 0;
}
