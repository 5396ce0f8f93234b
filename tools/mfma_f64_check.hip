// Checks, on gfx950, the two facts the lane-group NUTS kernel relies on:
//  (1) with X distributed as lane (c = l & 15, q = l >> 4) owning dims d = 4 s + q (s = 0..7) of chain c, the product
//      G^T = A . X^T computed by v_mfma_f64_16x16x4_f64 returns G in the SAME distribution (no lane movement);
//  (2) the accumulation over k is the in-order chain y = fma(A[i][k], x[k], y), k = 0..31, bit for bit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *X, double *G)
{
    const int l = threadIdx.x, c = l & 15, q = l >> 4;
    double xs[8];
    for (int s = 0; s < 8; ++s) xs[s] = X[c * 32 + 4 * s + q];
    for (int tile = 0; tile < 2; ++tile) {
        d4 acc = {0, 0, 0, 0};
        for (int s = 0; s < 8; ++s) {
            // A operand: lane (i = l & 15, k = l >> 4) holds A[16 tile + i][4 s + k]
            const double a = A[(16 * tile + (l & 15)) * 32 + 4 * s + (l >> 4)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xs[s], acc, 0, 0, 0);
        }
        // D: lane (col = c, q) holds rows q + 4 r, r = 0..3 -> dims 16 tile + 4 r + q
        for (int r = 0; r < 4; ++r) G[c * 32 + 16 * tile + 4 * r + q] = acc[r];
    }
}
int main()
{
    std::vector<double> A(32 * 32), X(16 * 32), G(16 * 32), R(16 * 32);
    srand(1);
    for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
    for (auto &v : X) v = (rand() / (double)RAND_MAX - 0.5) * 3;
    double *dA, *dX, *dG;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dX, X.size() * 8); hipMalloc(&dG, G.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dX, dG);
    hipMemcpy(G.data(), dG, G.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0; double maxd = 0;
    for (int c = 0; c < 16; ++c)
        for (int i = 0; i < 32; ++i) {
            double y = 0;
            for (int j = 0; j < 32; ++j) y = std::fma(A[i * 32 + j], X[c * 32 + j], y);
            R[c * 32 + i] = y;
            bad += (y != G[c * 32 + i]);
            maxd = std::fmax(maxd, std::fabs(y - G[c * 32 + i]));
        }
    printf("mismatching elements (bitwise): %d of 512, max abs diff %.3e\n", bad, maxd);
    return 0;
}
