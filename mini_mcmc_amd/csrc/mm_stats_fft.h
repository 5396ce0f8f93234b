/*
 * mm_stats_fft.h -- the lag sums of split-R-hat / ESS through the power spectrum (stats.rs:576-620 `autocov_fft`).
 *
 * The reference switches from the direct autocovariance to an FFT once a half-chain is longer than 100 draws
 * (stats.rs:549), i.e. at every BASELINE configuration: per chain and parameter one forward and one inverse complex
 * FFT of length next_pow2(2 m - 1) (rustfft).  `ess` (stats.rs:496-546) then only uses the MEAN over the half-chains of
 * every lag, and the transform is linear, so here
 *
 *   * per chain and parameter ONE forward complex FFT of length N = 64 R1 >= 2 m: the chain's two half-chains
 *     (centred on their own means, zero padded) travel as real and imaginary part, z = a + i b.  With A, B their
 *     spectra |Z(f)|^2 = |A|^2 + |B|^2 + 2 Im(A conj(B)), and the cross term is odd in f: it vanishes from every
 *     cosine sum over the full circle, so nothing has to be unpacked;
 *   * S(f) = sum over the wave's chains of |Z(f)|^2 is accumulated in registers (never the lag sums themselves);
 *   * one inverse per parameter for ALL chains: c_k = (1 / N) sum_f S(f) cos(2 pi f k / N) = the sum over every half-chain
 *     of sum_t y[t] y[t + k]  (mm_fft_finish_kernel, f64).
 *
 * The FFT of one (chain, parameter) is the work of one wave: time index t = 64 a + l sits in lane l, register a.
 *   pass 1   radix R1 over a in registers (decimation in frequency; the upper half of the inputs is the zero padding, so
 *            it is two DFTs of R1 / 2 points: even outputs from z, odd outputs from z[a] w_R1^a), twiddle w_N^(l b);
 *            what is left are R1 FFTs of 64 points over the lanes, one per residue b = f mod R1;
 *   pass 2   l = 8 c + e: exchange through LDS so that lane (beta, e) holds u_b[8 c + e], c = 0..7, for the rows
 *            b = beta + 8 j; radix 8 over c, twiddle w_64^(e g);
 *   pass 3   exchange inside the groups of 8 lanes (LDS again: 16 LDS instructions where a register transpose is 100
 *            vector instructions), radix 8 over e.  Lane (beta, g), register (j, h) ends with bin
 *            f = R1 (g + 8 h) + beta + 8 j; bins are never reordered: S is indexed by (lane, register) until the
 *            workgroup folds its waves' S into natural order once, at the end of the launch.
 * Vector instructions per FFT of 512 points: about 240 (two DFT-4, 7 + 7 twiddles, two DFT-8, |Z|^2), against 935 per
 * chain and parameter in the register-tile kernel's direct sums at m = 200 -- and O(m log m) instead of O(m^2) beyond.
 *
 * LDS per wave: R1 rows of 72 complex numbers (a row = the 64 points of one residue; pitch 72 = 64 + 8 makes both
 * exchanges conflict-free: pass 2 reads unit 72 b + 8 c + e, pass 3's data sits at 72 b + 9 g + e).
 *
 * Everything below the kernel is written per lane (MM_FD) so that a host build can run the 64 lanes of a wave one
 * after another, phase by phase, against a plain DFT (oracle/engine_host.cpp, tests/test_stats_fft_host.py).
 */
#ifndef MM_STATS_FFT_H
#define MM_STATS_FFT_H

#include <utility>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define MM_FD __device__ __forceinline__
#else
#include <cmath>
#define MM_FD static inline
#endif

struct mm_cx {
    float re, im;
};

MM_FD mm_cx mm_cx_add(mm_cx a, mm_cx b) { return mm_cx{a.re + b.re, a.im + b.im}; }
MM_FD mm_cx mm_cx_sub(mm_cx a, mm_cx b) { return mm_cx{a.re - b.re, a.im - b.im}; }
/* a * (wr + i wi): two multiplies, two fused multiply-adds */
MM_FD mm_cx mm_cx_mul(mm_cx a, float wr, float wi)
{
    return mm_cx{fmaf(-a.im, wi, a.re * wr), fmaf(a.re, wi, a.im * wr)};
}

/* Eight complex numbers from LDS at p + STRIDE k, k = 0..7.  The compiler pairs neighbouring reads into ds_read2_b64,
 * which the LDS serves at 128 bytes per clock -- half the rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table) -- and the
 * transform is LDS-bound (tools/fft_probe.hip: 672 of its 860 cycles per SIMD are LDS time).  -DMM_FFT_ASM_LDS_LOADS
 * issues eight ds_read_b64 from a volatile asm instead: in the probe the transform drops to 763-789 cycles, in the kernel
 * the wait inside the asm (the compiler cannot count these loads) costs what the faster reads gain -- [65536, 400, 3]
 * 0.154 ms either way, [65536, 1000, 3] 0.32 against 0.29 ms -- so the plain form is the default. */
#if defined(__HIPCC__) && defined(MM_FFT_ASM_LDS_LOADS)
template <int STRIDE, class P>
MM_FD void mm_lds_load8(P p, mm_cx (&out)[8])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const unsigned int a = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) const void *)(const void *)p;
    f2 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64 %0, %8 offset:%9\n\t"
                 "ds_read_b64 %1, %8 offset:%10\n\t"
                 "ds_read_b64 %2, %8 offset:%11\n\t"
                 "ds_read_b64 %3, %8 offset:%12\n\t"
                 "ds_read_b64 %4, %8 offset:%13\n\t"
                 "ds_read_b64 %5, %8 offset:%14\n\t"
                 "ds_read_b64 %6, %8 offset:%15\n\t"
                 "ds_read_b64 %7, %8 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(a), "n"(0), "n"(8 * STRIDE), "n"(16 * STRIDE), "n"(24 * STRIDE), "n"(32 * STRIDE), "n"(40 * STRIDE),
                   "n"(48 * STRIDE), "n"(56 * STRIDE)
                 : "memory");
    out[0] = mm_cx{r0[0], r0[1]};
    out[1] = mm_cx{r1[0], r1[1]};
    out[2] = mm_cx{r2[0], r2[1]};
    out[3] = mm_cx{r3[0], r3[1]};
    out[4] = mm_cx{r4[0], r4[1]};
    out[5] = mm_cx{r5[0], r5[1]};
    out[6] = mm_cx{r6[0], r6[1]};
    out[7] = mm_cx{r7[0], r7[1]};
}
#else
template <int STRIDE, class P>
MM_FD void mm_lds_load8(P p, mm_cx (&out)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k)
        out[k] = p[STRIDE * k];
}
#endif

/* cos(2 pi k / 32) for any k from its first octant */
constexpr float mm_cos32(int k)
{
    constexpr float t[9] = {1.0f,
                            0.98078528040323044913f,
                            0.92387953251128675613f,
                            0.83146961230254523708f,
                            0.70710678118654752440f,
                            0.55557023301960222474f,
                            0.38268343236508977173f,
                            0.19509032201612826785f,
                            0.0f};
    k = ((k % 32) + 32) % 32;
    if (k > 16)
        k = 32 - k;
    return k > 8 ? -t[16 - k] : t[k];
}
constexpr float mm_sin32(int k) { return mm_cos32(k - 8); }

/* a * exp(-2 pi i J / M) with J, M compile-time (M divides 32): the trivial rotations cost nothing, the odd eighths two
 * additions and two multiplies, the rest a full complex multiply with literal constants */
template <int M, int J>
MM_FD mm_cx mm_cx_rot(mm_cx a)
{
    constexpr int j = ((J % M) + M) % M;
    constexpr float s = 0.70710678118654752440f;
    if constexpr (j == 0)
        return a;
    else if constexpr (2 * j == M)
        return mm_cx{-a.re, -a.im};
    else if constexpr (4 * j == M)
        return mm_cx{a.im, -a.re};
    else if constexpr (4 * j == 3 * M)
        return mm_cx{-a.im, a.re};
    else if constexpr (8 * j == M)
        return mm_cx{(a.re + a.im) * s, (a.im - a.re) * s};
    else if constexpr (8 * j == 3 * M)
        return mm_cx{(a.im - a.re) * s, -((a.re + a.im) * s)};
    else {
        static_assert(32 % M == 0, "rotation angles are multiples of 2 pi / 32");
        constexpr int k = j * (32 / M);
        return mm_cx_mul(a, mm_cos32(k), -mm_sin32(k));
    }
}

constexpr int mm_bitrev(int p, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i)
        r |= ((p >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int mm_log2i(int v) { return v <= 1 ? 0 : 1 + mm_log2i(v / 2); }

/* in-register DFT of H points, decimation in frequency, radix 2 with the rotations above: position p ends with
 * X[mm_bitrev(p, log2 H)].  Butterfly I of the stage with span S: block I / S, offset J = I % S. */
template <int H, int S, int I>
MM_FD void mm_dif_bfly(mm_cx (&v)[H])
{
    constexpr int k = (I / S) * 2 * S + (I % S);
    const mm_cx a = v[k], b = v[k + S];
    v[k] = mm_cx_add(a, b);
    v[k + S] = mm_cx_rot<2 * S, I % S>(mm_cx_sub(a, b));
}
template <int H, int S, int... I>
MM_FD void mm_dif_stage(mm_cx (&v)[H], std::integer_sequence<int, I...>)
{
    (mm_dif_bfly<H, S, I>(v), ...);
}
template <int H, int S = H / 2>
MM_FD void mm_dft_dif(mm_cx (&v)[H])
{
    if constexpr (S >= 1) {
        mm_dif_stage<H, S>(v, std::make_integer_sequence<int, H / 2>{});
        mm_dft_dif<H, S / 2>(v);
    }
}

template <int R1>
struct mm_fft_plan {
    static_assert(R1 == 8 || R1 == 16 || R1 == 32, "N = 512, 1024 or 2048");
    static constexpr int H = R1 / 2;      /* non-zero inputs per lane */
    static constexpr int N = 64 * R1;     /* transform length */
    static constexpr int J = R1 / 8;      /* rows (residues b) per lane in passes 2 and 3 */
    static constexpr int PITCH = 72;      /* complex numbers per LDS row */
    static constexpr int LDS_CX = R1 * PITCH; /* complex numbers of LDS per wave */
    static constexpr int HB = mm_log2i(H);
};

/* bin held by lane (beta = lane >> 3, g = lane & 7), register (j, h) after pass 3 */
template <int R1>
constexpr int mm_fft_bin(int lane, int j, int h)
{
    return R1 * ((lane & 7) + 8 * h) + (lane >> 3) + 8 * j;
}

template <int R1, int... A>
MM_FD void mm_fft_turn(const mm_cx (&z)[R1 / 2], mm_cx (&od)[R1 / 2], std::integer_sequence<int, A...>)
{
    ((od[A] = mm_cx_rot<R1, A>(z[A])), ...);
}

/* pass 1: z[a] = point 64 a + lane (a < R1 / 2, the rest is padding); tw1(b) = w_N^(lane b) (a callable: the kernel
 * keeps the factors in registers or reads them from LDS).  Row b of the wave's LDS
 * block receives u_b[lane]. */
template <int R1, class P, class TW>
MM_FD void mm_fft_pass1(const mm_cx (&z)[R1 / 2], TW tw1, P lds, int lane)
{
    using pl = mm_fft_plan<R1>;
    mm_cx ev[pl::H], od[pl::H];
    /* odd residues: the inputs turned by w_R1^a first */
    mm_fft_turn<R1>(z, od, std::make_integer_sequence<int, pl::H>{});
#pragma unroll
    for (int a = 0; a < pl::H; ++a)
        ev[a] = z[a];
    mm_dft_dif<pl::H>(ev);
    mm_dft_dif<pl::H>(od);
#pragma unroll
    for (int p = 0; p < pl::H; ++p) {
        const int b0 = 2 * mm_bitrev(p, pl::HB);
        mm_cx t0 = ev[p];
        if (b0 != 0) {
            const mm_cx w = tw1(b0);
            t0 = mm_cx_mul(ev[p], w.re, w.im);
        }
        const mm_cx w1 = tw1(b0 + 1);
        lds[b0 * pl::PITCH + lane] = t0;
        lds[(b0 + 1) * pl::PITCH + lane] = mm_cx_mul(od[p], w1.re, w1.im);
    }
}

/* pass 1 without the zero half: all R1 points of a lane are data (the inner transforms of the long-chain kernel, whose
 * padding sits in the OUTER index).  v[a] = point 64 a + lane; row b receives u_b[lane] as above. */
template <int R1, class P, class TW>
MM_FD void mm_fft_pass1_full(mm_cx (&v)[R1], TW tw1, P lds, int lane)
{
    using pl = mm_fft_plan<R1>;
    constexpr int RB = mm_log2i(R1);
    mm_dft_dif<R1>(v); /* position p holds X[mm_bitrev(p, RB)] */
#pragma unroll
    for (int p = 0; p < R1; ++p) {
        const int b = mm_bitrev(p, RB);
        mm_cx t = v[p];
        if (b != 0) {
            const mm_cx w = tw1(b);
            t = mm_cx_mul(v[p], w.re, w.im);
        }
        lds[b * pl::PITCH + lane] = t;
    }
}

/* pass 2: lane (beta, e) takes u_b[8 c + e] of its rows b = beta + 8 j, radix 8 over c, twiddle tw2[g] = w_64^(e g),
 * and leaves V_b[g][e] at unit 72 b + 9 g + e of the same row (the row's other readers are the lanes of this group, and
 * a wave's LDS instructions execute in order: every lane's loads are issued before any lane's stores -- two functions so
 * that a host build can run them as two phases). */
template <int R1, class P>
MM_FD void mm_fft_pass2_load(mm_cx (&v)[R1 / 8][8], P lds, int lane)
{
    using pl = mm_fft_plan<R1>;
    const int beta = lane >> 3, e = lane & 7;
#pragma unroll
    for (int j = 0; j < pl::J; ++j)
        mm_lds_load8<8>(lds + (beta + 8 * j) * pl::PITCH + e, v[j]);
}
template <int R1, class P, class TW>
MM_FD void mm_fft_pass2_store(mm_cx (&v)[R1 / 8][8], TW tw2, P lds, int lane)
{
    using pl = mm_fft_plan<R1>;
    const int beta = lane >> 3, e = lane & 7;
#pragma unroll
    for (int j = 0; j < pl::J; ++j) {
        mm_dft_dif<8>(v[j]);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int g = mm_bitrev(p, 3);
            mm_cx t = v[j][p];
            if (g != 0) {
                const mm_cx w = tw2(g);
                t = mm_cx_mul(v[j][p], w.re, w.im);
            }
            lds[(beta + 8 * j) * pl::PITCH + 9 * g + e] = t;
        }
    }
}

/* pass 3: lane (beta, g) takes V_b[g][e], e = 0..7, radix 8 over e, and adds |X|^2 of its R1 bins to S[j][h] */
template <int R1, class P>
MM_FD void mm_fft_pass3(P lds, int lane, float (&S)[R1 / 8][8])
{
    using pl = mm_fft_plan<R1>;
    const int beta = lane >> 3, g = lane & 7;
    mm_cx w[pl::J][8];
#pragma unroll
    for (int j = 0; j < pl::J; ++j)
        mm_lds_load8<1>(lds + (beta + 8 * j) * pl::PITCH + 9 * g, w[j]);
#pragma unroll
    for (int j = 0; j < pl::J; ++j) {
        mm_dft_dif<8>(w[j]);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int h = mm_bitrev(p, 3);
            S[j][h] = fmaf(w[j][p].im, w[j][p].im, fmaf(w[j][p].re, w[j][p].re, S[j][h]));
        }
    }
}

#endif /* MM_STATS_FFT_H */
