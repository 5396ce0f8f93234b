// Stand-alone driver of the split-role kernel (mm_split_kernels.h) next to mm_run_kernel for quick A/B experiments:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 [-DMM_SPLIT_PRIO_Q=3]
//         [-DMM_SPLIT_MH_QP=0|1|2] tools/split_probe.hip -o /tmp/split_probe
// config 3 (HMC RosenbrockND(3), 65 536 chains, run(400, 50)) and config 2 (MH Gaussian2D, run(1000, 100)).
#define MM_NOISE_SCALAR_CUBICS 1 /* as mm_inst.inc */
#include "../mini_mcmc_amd/csrc/mm_split_kernels.h"
#include "../mini_mcmc_amd/csrc/mm_params.h"
#include <cstdio>
#include <vector>

/* role layout of the split kernel under test: pairs the transition wave draws itself, noise waves per pair, ring half
 * (0 = the plan's): -DMM_PROBE_MH_NN=3 -DMM_PROBE_MH_RB=6 -DMM_PROBE_MH_QP=0 ... */
#ifndef MM_PROBE_MH_QP
#define MM_PROBE_MH_QP (mm_split_mh_qp<float, 2>::value)
#endif
#ifndef MM_PROBE_MH_NN
#define MM_PROBE_MH_NN 1
#endif
#ifndef MM_PROBE_MH_RB
#define MM_PROBE_MH_RB 0
#endif
#ifndef MM_PROBE_HMC_QP
#define MM_PROBE_HMC_QP 0
#endif
#ifndef MM_PROBE_HMC_NN
#define MM_PROBE_HMC_NN 1
#endif
#ifndef MM_PROBE_HMC_RB
#define MM_PROBE_HMC_RB 0
#endif

template <class Tgt, int SAMPLER, int LCT> static void bench(const char *name, int D, unsigned NC, unsigned ND, float scale, mm_tparams<float> P)
{
    const unsigned long long C = 65536;
    std::vector<float> h(C * D);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = 0.01f * (float)(i % 97) - 0.3f;
    float *state, *out;
    (void)hipMalloc(&state, C * D * 4);
    (void)hipMalloc(&out, C * NC * D * 4);
    mm_run_args<float> a{};
    a.P = P;
    a.scale = scale; a.n_leapfrog = 10; a.state = state; a.out = out; a.n_chains = C; a.seed = 42;
    a.n_discard = ND; a.n_collect = NC; a.n_total = NC;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int kern = 0; kern < 2; ++kern) {
        for (int mode = 0; mode < 2; ++mode) {
            a.out = mode ? nullptr : out;
            float best = 1e9;
            for (int r = 0; r < 8; ++r) {
                (void)hipMemcpy(state, h.data(), C * D * 4, hipMemcpyHostToDevice);
                (void)hipEventRecord(e0);
                if (kern == 0)
                    (void)mm_launch_run<float, Tgt, SAMPLER, 2, LCT>(a, 1024, 64, 0);
#ifdef MM_PROBE_PW /* round 6: workgroup = PW (1 or 2) pairs of 64 chains, 4 / PW workgroups per CU; -DMM_PROBE_PW_HMC: HMC too */
#ifndef MM_PROBE_PW_HMC
#define MM_PROBE_PW_HMC 0
#endif
                else if (SAMPLER == MM_SAMPLER_MH || MM_PROBE_PW_HMC)
                    (void)mm_launch_run_splitw<float, Tgt, SAMPLER, MM_PROBE_PW, LCT, (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_QP : MM_PROBE_HMC_QP),
                                               (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_NN : MM_PROBE_HMC_NN),
                                               (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_RB : MM_PROBE_HMC_RB)>(a, 0);
#endif
                else
                    (void)mm_launch_run_split<float, Tgt, SAMPLER, LCT, (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_QP : MM_PROBE_HMC_QP),
                                              (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_NN : MM_PROBE_HMC_NN),
                                              (SAMPLER == MM_SAMPLER_MH ? MM_PROBE_MH_RB : MM_PROBE_HMC_RB)>(a, 0);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best)
                    best = ms;
            }
            double s = 0;
            std::vector<float> ho(C * D), hs(1 << 18);
            (void)hipMemcpy(ho.data(), state, C * D * 4, hipMemcpyDeviceToHost);
            for (float v : ho)
                s += v;
            if (!mode) {
                (void)hipMemcpy(hs.data(), out + (size_t)12345 * NC * D, hs.size() * 4, hipMemcpyDeviceToHost);
                for (float v : hs)
                    s += v;
            }
            printf("%-10s %-14s %s: %.4f ms   checksum %.6f\n", name, kern ? "split" : "mm_run_kernel", mode ? "no output " : "with output", best, s);
#ifdef MM_SPLIT_PROFILE
            if (kern == 1) {
                unsigned long long hp[2][3], zero[2][3] = {{0, 0, 0}, {0, 0, 0}};
                (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(mm_split_prof), sizeof(hp));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(mm_split_prof), zero, sizeof(zero));
                printf("    noise waves: %.1f %% of their time at the barrier   transition waves: %.1f %%, %.1f %% in their own draws   (ticks per wave and launch: %.0f)\n",
                       100.0 * (double)hp[0][0] / (double)hp[0][1], 100.0 * (double)hp[1][0] / (double)hp[1][1], 100.0 * (double)hp[1][2] / (double)hp[1][1], (double)hp[1][1] / 1024.0 / 8.0);
            }
#endif
        }
    }
    (void)hipFree(state);
    (void)hipFree(out);
}

int main()
{
    mm_tparams<float> P{};
    bench<mm_target<float, MM_ROSENBROCK_ND, 3>, MM_SAMPLER_HMC, 10>("hmc cfg3", 3, 400, 50, 0.032f, P);
    const double g[6] = {0.0, 0.0, 1.0, 0.0, 0.0, 1.0};
    mm_fill_params<float>(MM_GAUSSIAN2D, g, &P);
    bench<mm_target<float, MM_GAUSSIAN2D, 2>, MM_SAMPLER_MH, 0>("mh cfg2", 2, 1000, 100, 1.0f, P);
#ifdef MM_PROBE_MH_ALIGNED /* round 6: does the alignment of a chain's output row matter?  1008 x 8 B = 63 whole 128-byte lines per chain */
    bench<mm_target<float, MM_GAUSSIAN2D, 2>, MM_SAMPLER_MH, 0>("mh 1008", 2, 1008, 92, 1.0f, P);
    bench<mm_target<float, MM_GAUSSIAN2D, 2>, MM_SAMPLER_MH, 0>("mh 1024", 2, 1024, 76, 1.0f, P);
#endif
    return 0;
}
