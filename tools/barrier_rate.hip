// What does one workgroup barrier cost on gfx950?  One workgroup per CU of W waves; every wave runs `work` dependent
// FMAs (wave 0: `work0`), then s_barrier; cycles per round trip measured with s_memtime in wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_rate tools/barrier_rate.hip && /tmp/barrier_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float *out, long long *cyc, int rounds, int work, int work0, int lds_sync)
{
    __shared__ float buf[1024];
    float x = threadIdx.x * 1e-3f;
    const int w = (threadIdx.x >> 6) == 0 ? work0 : work;
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < w; ++i)
            x = fmaf(x, 1.0001f, 0.5f);
        if (lds_sync)
            buf[threadIdx.x] = x;
        __syncthreads();
        if (lds_sync)
            x += buf[(threadIdx.x + 64) & 1023];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main()
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int rounds = 2000;
    for (int waves : {4, 8, 16})
        for (int lds : {0, 1})
            for (auto wk : std::vector<std::pair<int, int>>{{0, 0}, {100, 100}, {100, 400}, {400, 100}}) {
                k<<<256, waves * 64>>>(out, cyc, rounds, wk.first, wk.second, lds);
                hipDeviceSynchronize();
                k<<<256, waves * 64>>>(out, cyc, rounds, wk.first, wk.second, lds);
                hipDeviceSynchronize();
                long long h[256];
                hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
                double m = 0;
                for (int i = 0; i < 256; ++i)
                    m += (double)h[i];
                printf("waves %2d lds %d work %3d (wave 0: %3d FMAs): %.0f cycles per round\n", waves, lds, wk.first, wk.second, m / 256 / rounds);
            }
    return 0;
}
