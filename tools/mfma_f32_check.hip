// The f32 counterpart of mfma_f64_check.hip, for a lane-group kernel on v_mfma_f32_16x16x4_f32:
//  (1) layout: lane (c = l & 15, q = l >> 4) owns coordinates coord(q, s) = 16 (s >> 2) + 4 q + (s & 3) of chain c; with the
//      A operand's columns taken in that order, G^T = A . X^T comes back in the same distribution;
//  (2) which f32 arithmetic reproduces the accumulation over k bit for bit?  Candidates: the in-order chain of fmaf over
//      the k-steps (4 products per MFMA, k ascending); per-MFMA "sum the four exact products, round once" (f64 emulation).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int coord(int q, int s) { return 16 * (s >> 2) + 4 * q + (s & 3); }
__global__ void k(const float *A, const float *X, float *G)
{
    const int l = threadIdx.x, c = l & 15, q = l >> 4;
    float xs[8];
    for (int s = 0; s < 8; ++s) xs[s] = X[c * 32 + coord(q, s)];
    for (int tile = 0; tile < 2; ++tile) {
        f4 acc = {0, 0, 0, 0};
        for (int s = 0; s < 8; ++s) {
            const float a = A[(16 * tile + (l & 15)) * 32 + coord(l >> 4, s)]; // A operand: lane (row, k)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[s], acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) G[c * 32 + 16 * tile + 4 * q + r] = acc[r]; // D: lane (col, q) holds rows 4 q + r
    }
}
int main()
{
    std::vector<float> A(32 * 32), X(16 * 32), G(16 * 32);
    srand(1);
    for (auto &v : A) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : X) v = (rand() / (float)RAND_MAX - 0.5f) * 3;
    float *dA, *dX, *dG;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dG, G.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dX, dG);
    (void)hipMemcpy(G.data(), dG, G.size() * 4, hipMemcpyDeviceToHost);
    int bad_chain = 0, bad_chain_rev = 0, bad_sum4 = 0, bad_sum4_f32acc = 0; double maxd = 0;
    for (int c = 0; c < 16; ++c)
        for (int i = 0; i < 32; ++i) {
            float y1 = 0, y2 = 0, y3 = 0; 
            for (int s = 0; s < 8; ++s) {
                for (int kq = 0; kq < 4; ++kq) y1 = std::fmaf(A[i * 32 + coord(kq, s)], X[c * 32 + coord(kq, s)], y1);
                for (int kq = 3; kq >= 0; --kq) y2 = std::fmaf(A[i * 32 + coord(kq, s)], X[c * 32 + coord(kq, s)], y2);
                double t = (double)y3;
                for (int kq = 0; kq < 4; ++kq) t += (double)A[i * 32 + coord(kq, s)] * (double)X[c * 32 + coord(kq, s)];
                y3 = (float)t;
            }
            const float g = G[c * 32 + i];
            bad_chain += (y1 != g); bad_chain_rev += (y2 != g); bad_sum4 += (y3 != g);
            maxd = std::fmax(maxd, std::fabs((double)y1 - g));
        }
    printf("bitwise mismatches of 512: fmaf chain k ascending %d, k descending %d, exact sum of 4 products rounded once %d; max |chain - mfma| %.3e\n",
           bad_chain, bad_chain_rev, bad_sum4, maxd);
    return 0;
}
