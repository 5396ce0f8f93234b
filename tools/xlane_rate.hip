// Issue rates of the cross-lane vector instructions on gfx950 (cycles per wave-instruction, per wave and per SIMD), the
// companion of tools/issue_rate.hip:  hipcc --offload-arch=gfx950 -O3 tools/xlane_rate.hip -o /tmp/xlane_rate && /tmp/xlane_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE> __global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink, int iters, float a)
{
    extern __shared__ float lds[];
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // v_permlane32_swap, four independent register pairs
            asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 1) { // v_permlane16_swap
            asm volatile(REP16("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 2) { // v_mov_b32_dpp quad_perm, independent
            asm volatile(REP16("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (MODE == 3) { // v_add_f32_dpp row_shr (the wave sums' step)
            asm volatile(REP16("v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (MODE == 4) { // v_cndmask_b32 with a mask in an SGPR pair
            asm volatile("s_mov_b64 s[20:21], 0xaaaaaaaa\n" REP16("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %5, s[20:21]\n v_cndmask_b32 %2, %2, %6, s[20:21]\n v_cndmask_b32 %3, %3, %7, s[20:21]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7) : "s20", "s21");
        } else if (MODE == 5) { // v_add_f32 e32 independent (reference)
            asm volatile(REP16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (MODE == 6) { // ds_bpermute_b32, independent
            unsigned addr = ((threadIdx.x ^ 9u) & 63u) * 4u;
            asm volatile(REP16("ds_bpermute_b32 %0, %8, %4\n ds_bpermute_b32 %1, %8, %5\n ds_bpermute_b32 %2, %8, %6\n ds_bpermute_b32 %3, %8, %7\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(addr));
        } else if (MODE == 7) { // ds_swizzle_b32 (xor 4 within groups of 32), independent
            asm volatile(REP16("ds_swizzle_b32 %0, %4 offset:0x101f\n ds_swizzle_b32 %1, %5 offset:0x101f\n ds_swizzle_b32 %2, %6 offset:0x101f\n ds_swizzle_b32 %3, %7 offset:0x101f\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (MODE == 8) { // v_mov_b32_dpp row_shr:4 with a bank mask (the xor-4 step of a transposition)
            asm volatile(REP16("v_mov_b32_dpp %0, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %1, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n"
                               "v_mov_b32_dpp %2, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %3, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        cyc[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + lds[0];
}

template <int MODE> void run(const char *name, int waves_per_simd)
{
    const int iters = 2000, block = 256 * waves_per_simd, grid = 256, inst_per_iter = 64;
    unsigned long long *cyc;
    float *sink;
    (void)hipMalloc(&cyc, (size_t)grid * (block / 64) * 8);
    (void)hipMalloc(&sink, (size_t)grid * block * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<MODE><<<grid, block, 100 * 1024>>>(cyc, sink, 50, 1.0001f);
    (void)hipDeviceSynchronize();
    k<MODE><<<grid, block, 100 * 1024>>>(cyc, sink, iters, 1.0001f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * (block / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double inst = (double)iters * inst_per_iter;
    printf("%-44s waves/SIMD=%d  per wave %.2f  per SIMD %.2f cycles/instr\n", name, waves_per_simd, (double)h[h.size() / 2] / inst,
           (double)h.back() / inst / waves_per_simd);
    (void)hipFree(cyc);
    (void)hipFree(sink);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<5>("v_add_f32 independent (reference)", w);
        run<0>("v_permlane32_swap", w);
        run<1>("v_permlane16_swap", w);
        run<2>("v_mov_b32_dpp quad_perm", w);
        run<8>("v_mov_b32_dpp row_shr/shl:4 bank-masked", w);
        run<3>("v_add_f32_dpp quad_perm", w);
        run<4>("v_cndmask_b32 (SGPR mask)", w);
        run<6>("ds_bpermute_b32", w);
        run<7>("ds_swizzle_b32", w);
    }
    return 0;
}
