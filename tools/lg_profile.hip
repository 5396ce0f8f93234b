// Section timing of the lane-group NUTS kernel (mm_nuts_lg.h built with MM_LG_PROFILE): where a wave's cycles go.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -shared -fPIC -I mini_mcmc_amd/csrc tools/lg_profile.hip -o tools/liblgprof.so
// Driven by tools/lg_profile.py (device pointers from torch tensors).
#define MM_LG_PROFILE 1
#include "mm_nuts_lg.h"

extern "C" int lgprof_run(const double *mat, double *state, void *adapt, unsigned long long n_chains,
                          unsigned long long seed, unsigned int m0, unsigned int n_steps, unsigned int n_discard,
                          int max_depth, double *scratch, double *rec, unsigned long long *prof)
{
    mm_nuts_lg_args a;
    a.mat = mat;
    a.state = state;
    a.adapt = (mm_nuts_adapt<double> *)adapt;
    a.out = nullptr;
    a.n_leapfrog = nullptr;
    a.depth_hist = nullptr;
    a.n_chains = n_chains;
    a.seed = seed;
    a.chain_offset = 0;
    a.n_total = 0;
    a.m0 = m0;
    a.n_pre = n_steps;
    a.n_rec = 0;
    a.write_initial = 0;
    a.out_t0 = 0;
    a.n_discard = n_discard;
    a.max_depth = max_depth;
    a.target_accept_p = 0.8;
    a.scratch = scratch;
    a.rec = rec;
    a.c_pad = (n_chains + 15) / 16 * 16;
    a.lists = nullptr;
    a.counts = nullptr;
    a.j0 = 0;
    a.j = 0;
    a.m = 0;
    a.row = 0xffffffffu;
    a.ctrl = nullptr;
    a.patience = 64u;
    a.min_unit = 16u;
    a.slots = nullptr;
    a.prof = prof;
    hipError_t e = mm_launch_nuts_lg<32>(a, nullptr);
    if (e != hipSuccess)
        return (int)e;
    return (int)hipDeviceSynchronize();
}

extern "C" unsigned long long lgprof_scratch_doubles_per_wave() { return mm_lg_cfg<32>::scratch_doubles_per_wave; }
extern "C" unsigned long long lgprof_rec_doubles_per_chain() { return mm_lg_cfg<32>::rec_doubles(1); }
