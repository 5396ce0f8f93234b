// Stand-alone driver of the headline kernel (mm_run_kernel<float, RosenbrockND<3>, HMC, PIPE=2, L=10>, config 3) for
// quick A/B experiments on mm_kernels.h: compiles in seconds, no library rebuild.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 [-DMM_PROBE_SKIP_STORE | -DMM_PROBE_SKIP_FLUSH] tools/hmc_kernel_probe.hip -o /tmp/probe
// (-DMM_PROBE_SKIP_STORE: the flush reads the tile but does not store; -DMM_PROBE_SKIP_FLUSH: no flush at all; the
//  difference between the three builds is the cost of the stores / of the LDS reads / of the staging writes)
#include "../mini_mcmc_amd/csrc/mm_kernels.h"
#include <cstdio>
#include <vector>
int main()
{
    using Tgt = mm_target<float, MM_ROSENBROCK_ND, 3>;
    const unsigned long long C = 65536, NC = 400, ND = 50;
    std::vector<float> h(C * 3);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)(i % 97) - 0.3f;
    float *state, *out;
    (void)hipMalloc(&state, C * 3 * 4);
    (void)hipMalloc(&out, C * NC * 3 * 4);
    (void)hipMemcpy(state, h.data(), C * 3 * 4, hipMemcpyHostToDevice);
    mm_run_args<float> a{};
    a.scale = 0.032f; a.n_leapfrog = 10; a.state = state; a.out = out; a.n_chains = C; a.seed = 42;
    a.n_discard = ND; a.n_collect = NC; a.n_total = NC;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        a.out = mode ? nullptr : out;
        float best = 1e9;
        for (int r = 0; r < 8; ++r) {
            (void)hipEventRecord(e0);
            (void)mm_launch_run<float, Tgt, MM_SAMPLER_HMC, 2, 10>(a, 1024, 64, 0);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s: %.4f ms\n", mode ? "no output " : "with output", best);
    }
    double s = 0; std::vector<float> ho(1 << 16);
    (void)hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    for (float v : ho) s += v;
    printf("checksum %.6f\n", s);
    return 0;
}
