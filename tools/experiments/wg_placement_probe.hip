// Where do the waves of small workgroups land?  For the split kernel with ONE pair per workgroup (round 6) the role
// rotation must put the four transition waves of a CU's four workgroups on four different SIMDs: records HW_ID / XCC_ID of every
// wave of a grid of 256-thread workgroups with 36 KB of LDS each (four per CU) and prints, per CU, which (workgroup, wave) sit
// on which SIMD.    hipcc --offload-arch=gfx950 -O2 tools/experiments/wg_placement_probe.hip -o /tmp/wgp && /tmp/wgp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned int *out, int spin)
{
    extern __shared__ unsigned int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident long enough that the whole grid is co-resident (1024 workgroups = 4 per CU)
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin)
        ;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + lds[threadIdx.x] * 0;
    }
}
int main()
{
    const int G = 1024;
    unsigned int *d;
    (void)hipMalloc(&d, G * 4 * 2 * 4);
    std::vector<unsigned int> h(G * 4 * 2);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 36 * 1024, 0, d, 2000000);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // key = (xcc, se, sh, cu); value: list of (block, wave, simd, slot)
    std::map<unsigned int, std::vector<std::array<int, 4>>> cus;
    for (int b = 0; b < G; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned int hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15u;
            const int slot = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({b, w, simd, slot});
        }
    printf("%zu distinct CUs hold the grid\n", cus.size());
    int shown = 0, n4 = 0, same_simd_w0 = 0;
    std::map<int, int> hist_rot_b3, hist_rot_b8, hist_rot_b256;
    for (auto &kv : cus) {
        std::map<int, int> blocks; // block -> simd of its wave 0
        std::map<int, std::map<int, int>> wave_simd;
        for (auto &e : kv.second) {
            wave_simd[e[0]][e[1]] = e[2];
            if (e[1] == 0)
                blocks[e[0]] = e[2];
        }
        if (blocks.size() == 4)
            ++n4;
        // how many distinct SIMDs would the transition waves land on under each rotation rule?
        auto distinct = [&](auto rot) {
            std::map<int, int> s;
            for (auto &bw : wave_simd)
                s[bw.second[rot(bw.first) & 3]]++;
            return (int)s.size();
        };
        hist_rot_b3[distinct([](int b) { return b; })]++;
        hist_rot_b8[distinct([](int b) { return b >> 3; })]++;
        hist_rot_b256[distinct([](int b) { return b >> 8; })]++;
        if (shown < 6) {
            printf("xcc %u se %u sh %u cu %u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (auto &bw : wave_simd) {
                printf("  wg %d -> simd", bw.first);
                for (auto &ws : bw.second)
                    printf(" %d", ws.second);
            }
            printf("\n");
            ++shown;
        }
    }
    printf("CUs with exactly 4 workgroups: %d\n", n4);
    for (auto &kv : hist_rot_b3) printf("rotation b & 3: %d CUs have their transition waves on %d distinct SIMDs\n", kv.second, kv.first);
    for (auto &kv : hist_rot_b8) printf("rotation (b >> 3) & 3: %d CUs on %d distinct SIMDs\n", kv.second, kv.first);
    for (auto &kv : hist_rot_b256) printf("rotation (b >> 8) & 3: %d CUs on %d distinct SIMDs\n", kv.second, kv.first);
    return 0;
}
