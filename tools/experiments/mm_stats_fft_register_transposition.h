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
 *   pass 2   l = 8 c + e: a transposition through LDS (eight lanes x eight registers, over c) so that lane (beta, e) holds
 *            u_b[8 c + e], c = 0..7, for the rows b = beta + 8 j; radix 8 over c, twiddle w_64^(e g);
 *   pass 3   the same transposition over e, in registers: e sits in the lane bits v_permlane32_swap / v_permlane16_swap /
 *            a quad permute exchange (48 vector instructions); radix 8 over e.  Lane (beta, g), register (j, h) ends with
 *            bin f = R1 (g + 8 h) + beta + 8 j; bins are never reordered: S is indexed by (lane, register) until the
 *            workgroup folds its waves' S into natural order once, at the end of the launch.
 * Vector instructions per FFT of 512 points: about 290 (two DFT-4, 7 + 7 twiddles, two DFT-8, the transposition,
 * |Z|^2), against 935 per chain and parameter in the register-tile kernel's direct sums at m = 200 -- and O(m log m)
 * instead of O(m^2) beyond.
 *
 * Why only ONE of the two transpositions goes through LDS (tools/fft_probe.hip): with both there, a transform is 860
 * cycles per SIMD at any occupancy, of which the LDS traffic alone is 710 (16 KB per transform through a write path of
 * ~90 bytes per clock per CU) and the arithmetic alone 500: the kernel was LDS-bandwidth bound.
 *
 * LDS per wave: R1 rows of 66 complex numbers (a row = the 64 points of one residue, stored at the writer's lane;
 * pitch 66 = 2 mod 32 makes pass 2's gather conflict-free).
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
#define MM_HDC __host__ __device__ constexpr
#else
#define MM_HDC constexpr
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
    static constexpr int PITCH = 66;      /* complex numbers per LDS row */
    static constexpr int LDS_CX = R1 * PITCH; /* complex numbers of LDS per wave */
    static constexpr int HB = mm_log2i(H);
};

/* The two 3-bit fields of a lane.  Position l = 8 c + e inside a block of 64 points: c sits in lane bits 3..1, e in lane
 * bits 5, 4 and 0 -- the bits v_permlane32_swap, v_permlane16_swap and a quad permute exchange, which is what lets the
 * second transposition (over e) stay in registers.  After the first transposition the field of c holds beta (the residue
 * b mod 8 of the lane's rows), after the second the field of e holds g. */
MM_HDC int mm_fft_lane_c(int lane) { return (lane >> 1) & 7; }
MM_HDC int mm_fft_lane_e(int lane) { return ((lane >> 5) & 1) << 2 | ((lane >> 4) & 1) << 1 | (lane & 1); }
MM_HDC int mm_fft_lane_pos(int lane) { return 8 * mm_fft_lane_c(lane) + mm_fft_lane_e(lane); }

/* bin held by lane (beta, g), register (j, h) after pass 3 */
template <int R1>
MM_HDC int mm_fft_bin(int lane, int j, int h)
{
    return R1 * (mm_fft_lane_e(lane) + 8 * h) + mm_fft_lane_c(lane) + 8 * j;
}

template <int R1, int... A>
MM_FD void mm_fft_turn(const mm_cx (&z)[R1 / 2], mm_cx (&od)[R1 / 2], std::integer_sequence<int, A...>)
{
    ((od[A] = mm_cx_rot<R1, A>(z[A])), ...);
}

/* pass 1: z[a] = point 64 a + mm_fft_lane_pos(lane) (a < R1 / 2, the rest is padding); tw1(b) = w_N^(pos b) (a callable:
 * the kernel keeps the factors in registers or reads them from LDS).  Row b of the wave's LDS block receives u_b[pos] at
 * the writer's LANE (unit b PITCH + lane: a wave's store is 64 contiguous units). */
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

/* pass 2: lane (beta, e) takes u_b[8 c + e], c = 0..7, of its rows b = beta + 8 j -- written by the lanes (c, e), i.e. at
 * units (lane & 0x31) + 2 c of the row; with PITCH = 66 = 2 mod 32 the 32 lanes of a half-wave meet 32 different banks
 * (2 beta + e0 + 16 e1) --, radix 8 over c, twiddle tw2(g) = w_64^(e g); V[j][g] in natural order of g. */
template <int R1, class P, class TW>
MM_FD void mm_fft_pass2(mm_cx (&V)[R1 / 8][8], TW tw2, P lds, int lane)
{
    using pl = mm_fft_plan<R1>;
    const int beta = mm_fft_lane_c(lane), eb = lane & 0x31;
    mm_cx v[pl::J][8];
#pragma unroll
    for (int j = 0; j < pl::J; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v[j][c] = lds[(beta + 8 * j) * pl::PITCH + eb + 2 * c];
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
            V[j][g] = t;
        }
    }
}

/* The second transposition, in registers: inside every group of eight lanes that share beta, lane e register g
 * becomes lane g register e.  Three steps, one per bit: bit 2 of (e, g) by v_permlane32_swap (lane bit 5), bit 1 by
 * v_permlane16_swap (lane bit 4) -- one instruction exchanges the upper lanes of one register with the lower lanes of
 * another, i.e. a whole step for a register pair --, bit 0 (lane bit 0) by a quad permute and three selects. */
#ifdef __HIPCC__
MM_FD void mm_fft_swap_hi_lo32(float &x0, float &x1)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
    x0 = __uint_as_float(r[0]);
    x1 = __uint_as_float(r[1]);
}
MM_FD void mm_fft_swap_hi_lo16(float &x0, float &x1)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
    x0 = __uint_as_float(r[0]);
    x1 = __uint_as_float(r[1]);
}
MM_FD void mm_fft_swap_odd_even(float &x0, float &x1, bool odd)
{
    const float send = odd ? x0 : x1;
    const float got = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
    x0 = odd ? got : x0;
    x1 = odd ? x1 : got;
}
template <int JJ>
MM_FD void mm_fft_xpose(mm_cx (&V)[JJ][8], int lane)
{
    const bool odd = lane & 1;
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { /* bit 2 */
            mm_fft_swap_hi_lo32(V[j][g].re, V[j][g | 4].re);
            mm_fft_swap_hi_lo32(V[j][g].im, V[j][g | 4].im);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { /* bit 1 */
            const int g = (k & 1) | (k & 2) << 1;
            mm_fft_swap_hi_lo16(V[j][g].re, V[j][g | 2].re);
            mm_fft_swap_hi_lo16(V[j][g].im, V[j][g | 2].im);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { /* bit 0 */
            const int g = 2 * k;
            mm_fft_swap_odd_even(V[j][g].re, V[j][g | 1].re, odd);
            mm_fft_swap_odd_even(V[j][g].im, V[j][g | 1].im, odd);
        }
    }
}
#else
/* host twin of the above on the 64 lanes of a wave at once (test harness): element (lane field e, register g) <-> (g, e) */
template <int JJ>
static inline void mm_fft_xpose_wave(mm_cx (*V)[JJ][8])
{
    for (int j = 0; j < JJ; ++j)
        for (int lane = 0; lane < 64; ++lane)
            for (int g = 0; g < 8; ++g) {
                const int e = mm_fft_lane_e(lane);
                if (e < g) { /* each pair once */
                    const int partner = (lane & ~0x31) | ((g >> 2) & 1) << 5 | ((g >> 1) & 1) << 4 | (g & 1);
                    const mm_cx t = V[lane][j][g];
                    V[lane][j][g] = V[partner][j][e];
                    V[partner][j][e] = t;
                }
            }
}
#endif

/* pass 3: lane (beta, g) holds W[j][e] = V_b[g][e], e = 0..7: radix 8 over e, |X|^2 of its R1 bins added to S[j][h] */
template <int R1>
MM_FD void mm_fft_pass3(mm_cx (&W)[R1 / 8][8], float (&S)[R1 / 8][8])
{
    using pl = mm_fft_plan<R1>;
#pragma unroll
    for (int j = 0; j < pl::J; ++j) {
        mm_dft_dif<8>(W[j]);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int h = mm_bitrev(p, 3);
            S[j][h] = fmaf(W[j][p].im, W[j][p].im, fmaf(W[j][p].re, W[j][p].re, S[j][h]));
        }
    }
}

#endif /* MM_STATS_FFT_H */
