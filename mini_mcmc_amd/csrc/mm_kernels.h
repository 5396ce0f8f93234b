/*
 * mm_kernels.h -- the many-chain sampling kernel for gfx950 (MH and HMC share one skeleton).
 *
 * Replaces the reference's chain drivers:
 *   run_chain core.rs:55-73 + ChainRunner::run core.rs:176-186 (rayon task per chain, then ndarray::stack)
 *   HMC::run hmc.rs:137-158 (one batched burn tensor op per arithmetic step, slice_assign per iteration)
 *
 * Mapping onto CDNA4
 *   - one chain per lane; a wave owns 64 consecutive chains; position (and for HMC gradient, momentum) live in
 *     VGPRs for the whole launch; the target's parameters are kernel arguments (SGPRs / scalar cache)
 *   - the whole run (burn-in + collection) is ONE launch by default: state is read once and written once
 *   - the output contract is chain-major [C, n_collect, D]; a lane-per-chain store would put every lane in its
 *     own cache line, so each wave stages TILE_T iterations in LDS (row per chain, odd stride => conflict free
 *     on the lane-strided writes AND on the chain-linear reads) and then writes, for each of its 64 chains,
 *     one contiguous run of TILE_T*D elements with whole-wave coalesced stores
 *   - no inter-wave communication at all: a wave only synchronises with itself (wave-scope fences), so blocks
 *     are one wave (64 threads) and 65 536 chains give 1024 independent workgroups = 4 per CU, one per SIMD
 *   - accept decisions are tallied per lane (one add under the accept mask) and summed over the wave once, at the end
 */
#ifndef MM_KERNELS_H
#define MM_KERNELS_H

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "mm_samplers.h"


#define MM_SAMPLER_MH 0
#define MM_SAMPLER_HMC 1

template <class T> struct mm_run_args {
    mm_tparams<T> P;
    T scale;        /* MH: proposal std ; HMC: step size */
    int n_leapfrog; /* HMC */
    T *state;       /* [n_chains, D] current positions, row-major (the reference's layout) */
    T *out;         /* [n_chains, n_total, D] or NULL */
    unsigned long long *accept;      /* [n_chains] running per-chain accept counts, or NULL */
    unsigned long long *accept_total; /* single counter fed by the wave ballots, or NULL */
    unsigned long long n_chains;
    unsigned long long seed, chain_offset;
    unsigned int iter0;     /* iteration index of the first transition of this launch */
    unsigned int n_discard; /* transitions without output */
    unsigned int n_collect; /* transitions with output, written to rows out_t0 .. out_t0 + n_collect */
    unsigned int out_t0;
    unsigned long long n_total; /* row count of `out` per chain (n_collect of the whole run) */
};

constexpr int mm_gcd_c(int a, int b) { return b == 0 ? a : mm_gcd_c(b, a % b); }

/* TILE_T iterations of 64 chains staged per flush */
template <class T, int D, int TILE_T> struct mm_tile_t {
    static constexpr int tile_t = TILE_T;
    static constexpr int run = tile_t * D;
    /* row pitch of the staged tile.  When a chain's run is a whole number of 16-byte pieces the pitch keeps every
     * piece 16-byte aligned (one ds_read_b128 per piece in the flush; the lane-strided staging writes then meet a
     * 4-way bank conflict, which costs a few LDS cycles per transition); otherwise odd => conflict-free both ways */
    static constexpr int epl = 16 / (int)sizeof(T);
    static constexpr int stride = (run % epl) == 0 ? run + epl : (run | 1);
    static constexpr size_t lds_bytes_per_wave = (size_t)64 * stride * sizeof(T);
    /* f32 kernels keep the tables of the normal draw and of ln u (mm_rng.h) in LDS, once per block, before the tiles */
    static constexpr size_t lds_bytes_table = (sizeof(T) == 4) ? (size_t)MM_NOISE_TABLE_BYTES : 0;
};
/* about 96 f32 (48 f64) elements per chain => 25.6 KB of LDS per wave; even whenever possible: the kernels advance
 * two transitions at a time */
template <class T, int D> struct mm_tile_default {
    static constexpr int target = (sizeof(T) == 4) ? 96 : 48;
    static constexpr int tile_t = (target / D) >= 2 ? ((target / D) & ~1) : 1;
};
template <class T, int D> struct mm_tile : mm_tile_t<T, D, mm_tile_default<T, D>::tile_t> {};

/* Write the staged tile of one wave to `out`: for each of the wave's 64 chains one contiguous run of nt*D elements
 * starting at row `row0` of that chain.  Whole-wave stores of 64 consecutive elements of the chain-linear image. */
template <class T, int D, class Tile = mm_tile<T, D>>
__device__ __forceinline__ void mm_flush_tile_raw(T *out, unsigned long long n_total, unsigned long long n_chains,
                                                  const T *tile, int lane, unsigned long long wave_c0,
                                                  unsigned long long row0, unsigned int nt);

template <class T, int D>
__device__ __forceinline__ void mm_flush_tile(const mm_run_args<T> &a, const T *tile, int lane,
                                              unsigned long long wave_c0, unsigned long long row0, unsigned int nt)
{
    mm_flush_tile_raw<T, D>(a.out, a.n_total, a.n_chains, tile, lane, wave_c0, row0, nt);
}

template <class T, int D, class Tile>
__device__ __forceinline__ void mm_flush_tile_raw(T *out, unsigned long long n_total, unsigned long long n_chains,
                                                  const T *tile, int lane, unsigned long long wave_c0,
                                                  unsigned long long row0, unsigned int nt)
{
    constexpr int TILE_T = Tile::tile_t, RUN = Tile::run, STRIDE = Tile::stride;
#ifdef MM_PROBE_SKIP_FLUSH /* measurement aid of tools/hmc_kernel_probe.hip; never defined in the library build */
    return;
#endif
    /* LDS operations of one wave execute in order, so the reads below see the rows staged by all 64 lanes and the
     * next tile's writes cannot overtake them: only the compiler has to keep program order (wave_barrier). */
    __builtin_amdgcn_wave_barrier();
    /* wave-uniform base of this tile in `out`; a chain's rows are chain_stride elements apart
     * (host guarantees n_total * D < 2^32, so j * chain_stride is one v_mad_u64_u32) */
    const unsigned int chain_stride = (unsigned int)(n_total * D);
    T *const wbase = out + (wave_c0 * n_total + row0) * D;
    const unsigned int n_valid = (unsigned int)min(64ull, n_chains > wave_c0 ? n_chains - wave_c0 : 0ull);
    constexpr int EPL = 16 / (int)sizeof(T); /* elements per 16-byte store: 4 (f32) or 2 (f64) */
    /* piece p = k*64 + lane of the chain-linear image (see below) belongs to chain p / V; because 64 k advances by a
     * whole number of chains every K0 = V / gcd(64, V) steps (V = 24: K0 = 3 steps = 8 chains), a lane meets only K0
     * different (chain-in-group, element) pairs: its K0 LDS / global offsets are loop invariants and step a of the
     * outer loop adds compile-time (LDS) and wave-uniform (global) constants */
    constexpr int VP = (RUN % EPL) == 0 ? RUN / EPL : 1;
    constexpr int GC = mm_gcd_c(64, VP), K0 = VP / GC, J0 = 64 / GC;
    if (nt == (unsigned int)TILE_T && (RUN % EPL) == 0 && K0 <= 4 && n_valid == 64u &&
        (unsigned long long)chain_stride * 64ull * sizeof(T) < (1ull << 32)) {
        /* Full tile of a full wave (the common case): all reads of the tile are issued before the first store has
         * to wait for them, no per-piece address arithmetic, no exec-mask branches. */
        typedef T mm_vec16 __attribute__((ext_vector_type(EPL), aligned(sizeof(T))));
        unsigned int src_off[K0], dst_off[K0];
        MM_UNROLL
        for (int m = 0; m < K0; ++m) {
            const int p = m * 64 + lane;
            const int j = p / VP;
            const int e = (p - j * VP) * EPL;
            src_off[m] = (unsigned int)(j * STRIDE + e);
            dst_off[m] = ((unsigned int)j * chain_stride + (unsigned int)e) * (unsigned int)sizeof(T); /* bytes */
        }
        /* every read of the tile is in flight before the first store waits for its data */
        typedef T mm_vec16a __attribute__((ext_vector_type(EPL))); /* 16-byte aligned: STRIDE % EPL == 0 */
        mm_vec16a v[GC][K0];
        MM_UNROLL
        for (int a = 0; a < GC; ++a) {
            MM_UNROLL
            for (int m = 0; m < K0; ++m)
                v[a][m] = *reinterpret_cast<const mm_vec16a *>(tile + a * J0 * STRIDE + src_off[m]);
        }
        MM_UNROLL
        for (int a = 0; a < GC; ++a) {
            char *const gbase = reinterpret_cast<char *>(wbase + (unsigned long long)(a * J0) * chain_stride);
            MM_UNROLL
            for (int m = 0; m < K0; ++m) {
#ifdef MM_PROBE_SKIP_STORE /* measurement aid (tools/hmc_kernel_probe.hip): the tile is read, not written out */
                asm volatile("" ::"v"(v[a][m]), "v"(gbase + dst_off[m]));
#else
                *reinterpret_cast<mm_vec16 *>(gbase + dst_off[m]) = v[a][m];
#endif
            }
        }
    } else if (nt == (unsigned int)TILE_T && (RUN % EPL) == 0) {
        /* Full tile.  A wave's store instructions are issue-bound (~100 cycles each, whatever their width), so the
         * tile leaves as 16-byte stores: the chain-linear image is cut into 16-byte pieces, piece p = k*64 + lane
         * belongs to chain p / V at element EPL * (p % V) (V = RUN / EPL pieces per chain, compile-time divisor).
         * 64 chains x RUN elements = V store instructions instead of RUN. */
        typedef T mm_vec16 __attribute__((ext_vector_type(EPL), aligned(sizeof(T))));
        constexpr int V = RUN / EPL;
        const bool full = n_valid == 64u;
#pragma unroll 4
        for (int k = 0; k < V; ++k) {
            const int piece = k * 64 + lane;
            const int j = piece / V;
            const int e = (piece - j * V) * EPL;
            const T *src = tile + j * STRIDE + e;
            mm_vec16 v;
            MM_UNROLL
            for (int i = 0; i < EPL; ++i)
                v[i] = src[i];
            if (full || (unsigned int)j < n_valid)
                *reinterpret_cast<mm_vec16 *>(wbase + (unsigned long long)j * chain_stride + e) = v;
        }
    } else if (nt == (unsigned int)TILE_T) {
        /* full tile, generic RUN: RUN whole-wave stores of 64 consecutive elements of the chain-linear image */
        constexpr int G = 8;
        const bool full = n_valid == 64u;
#pragma unroll 1
        for (int k0 = 0; k0 < RUN; k0 += G) {
            T v[G];
            unsigned long long off[G];
            bool ok[G];
            MM_UNROLL
            for (int gi = 0; gi < G; ++gi) {
                const int idx = (k0 + gi) * 64 + lane;
                const int j = idx / RUN; /* compile-time divisor */
                const int e = idx - j * RUN;
                ok[gi] = (k0 + gi < RUN) && (full || (unsigned int)j < n_valid);
                v[gi] = tile[(k0 + gi < RUN) ? j * STRIDE + e : 0];
                off[gi] = (unsigned long long)j * chain_stride + e;
            }
            MM_UNROLL
            for (int gi = 0; gi < G; ++gi)
                if (ok[gi])
                    wbase[off[gi]] = v[gi];
        }
    } else if ((((int)nt * D) % EPL) == 0) {
        /* short tile (the last one of a run, or a staggered first one) whose runs are whole 16-byte pieces: the same
         * chain-linear cut with a run-time piece count per chain */
        typedef T mm_vec16 __attribute__((ext_vector_type(EPL), aligned(sizeof(T))));
        const unsigned int V = nt * (unsigned int)D / (unsigned int)EPL;
        const float rV = 1.0f / (float)V;
        for (unsigned int piece = (unsigned int)lane; piece < 64u * V; piece += 64u) {
            unsigned int j = (unsigned int)((float)piece * rV); /* piece / V for piece < 2^16: off by at most one */
            j -= (j * V > piece);
            j += ((j + 1u) * V <= piece);
            const unsigned int e = (piece - j * V) * (unsigned int)EPL;
            const T *src = tile + j * STRIDE + e;
            mm_vec16 v;
            MM_UNROLL
            for (int i = 0; i < EPL; ++i)
                v[i] = src[i];
            if (j < n_valid)
                *reinterpret_cast<mm_vec16 *>(wbase + (unsigned long long)j * chain_stride + e) = v;
        }
    } else {
        const int runv = (int)nt * D;
        for (unsigned int j = 0; j < n_valid; ++j)
            for (int e = lane; e < runv; e += 64)
                wbase[(unsigned long long)j * chain_stride + e] = tile[j * STRIDE + e];
    }
    __builtin_amdgcn_wave_barrier();
}

/* The noise of a transition is a pure function of (seed, chain, iteration), independent of the chain's state, and a
 * lone dependent chain of VALU instructions issues only one instruction per ~5 cycles on gfx950 while independent or
 * packed work fills the gaps (tools/valu_rate.hip; more waves per SIMD do not help).  Hence
 *   PIPE = 0: noise and transition of iteration t back to back (plain);
 *   PIPE = 1: the noise of t+1 is computed in the same basic block as transition t (software pipelining);
 *   PIPE = 2: the noise of TWO iterations (t+2, t+3) is computed at once -- both Philox counters interleaved, the
 *             Box-Muller / log arithmetic packed across the two iterations (v_pk_*_f32) -- next to transitions t, t+1.
 * LCT > 0: compile-time leapfrog count (the loop is unrolled into that block).  Results are bit-identical for
 * every (PIPE, LCT): all variants evaluate the same functions (mm_rng.h, mm_samplers.h). */
template <class T, class Tgt, int SAMPLER, int PIPE = 0, int LCT = 0>
__device__ __forceinline__ void mm_run_kernel_body(const mm_run_args<T> &a)
{
    constexpr int D = Tgt::dim;
    using Tile = mm_tile<T, D>;
    constexpr int TILE_T = Tile::tile_t, STRIDE = Tile::stride;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    T *lds = reinterpret_cast<T *>(mm_lds_raw + Tile::lds_bytes_table);
    /* where the f32 normals' table is read from: the block's LDS copy (f64 draws do not use a table) */
    using Tab = typename mm_cond<sizeof(T) == 4, mm_icdf_lds, mm_icdf_global>::type;
    constexpr bool MHP = mm_mh_paired<T, D, SAMPLER>::value; /* the MH sampler's paired f32 stream at D <= 2 (mm_rng.h) */
    Tab tab;
    if constexpr (sizeof(T) == 4) {
        mm_icdf_lds_fill(reinterpret_cast<float *>(mm_lds_raw), threadIdx.x, blockDim.x);
        mm_icdf_lds_check(mm_lds_raw);
        tab.tab = mm_icdf_lds_at0();
        __syncthreads();
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    T *tile = lds + (size_t)wave * 64 * STRIDE;
    const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long wave_c0 = c - lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;

    T x[D], g[D], lp;
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = active ? a.state[c * D + i] : T(0);
    if (SAMPLER == MM_SAMPLER_HMC)
        lp = Tgt::logp_grad(a.P, x, g);
    else
        lp = Tgt::logp(a.P, x);

    unsigned int it = a.iter0;
    unsigned long long n_acc = 0;
    T zc[D], ln_uc = 0; /* PIPE: noise of the current transition, drawn one transition ahead */
    if (PIPE) {
        T u;
        mm_draw_noise<D, Tab, MHP>(a.seed, chain, it, zc, &u, tab);
        ln_uc = mm_ln_accept(u, tab);
    }

    auto step = [&]() {
        int acc;
        if (PIPE) {
            T zn[D], un;
            mm_draw_noise<D, Tab, MHP>(a.seed, chain, it + 1, zn, &un, tab);
            const T ln_un = mm_ln_accept(un, tab);
            if (SAMPLER == MM_SAMPLER_HMC)
                acc = mm_hmc_step_noise<T, Tgt, LCT>(a.P, a.scale, a.n_leapfrog, x, &lp, g, zc, ln_uc);
            else
                acc = mm_mh_step_noise<T, Tgt>(a.P, a.scale, x, &lp, zc, ln_uc);
            MM_UNROLL
            for (int i = 0; i < D; ++i)
                zc[i] = zn[i];
            ln_uc = ln_un;
        } else if (SAMPLER == MM_SAMPLER_HMC) {
            acc = mm_hmc_step<T, Tgt>(a.P, a.scale, a.n_leapfrog, x, &lp, g, a.seed, chain, it);
        } else {
            acc = mm_mh_step<T, Tgt>(a.P, a.scale, x, &lp, a.seed, chain, it);
        }
        n_acc += (unsigned long long)acc; /* lanes past n_chains count too: their counters are dropped at the end */
        ++it;
    };

    if (PIPE == 2) {
        /* Pairs of transitions; the noise of the next pair is drawn alongside the current pair's transitions.
         * (za, zb) always hold the noise of iterations (it, it + 1).  The loops are kept free of branches other than
         * their back edges -- burn-in: pairs, then one lone transition if the count is odd; collection: one tile at
         * a time, TILE_T / 2 pairs with unconditional staging, then the flush -- a taken branch costs a wave more
         * than the three LDS writes it would skip. */
        T za[D], zb[D], lna, lnb;
        mm_draw_noise_pair<D, Tab, true, MHP>(a.seed, chain, it, za, &lna, zb, &lnb, tab);
        unsigned int n_acc32 = 0;
        /* every lambda here MUST be inlined: an out-of-line call that captures x / g / lp by reference pins them (and the
         * kernel arguments) in scratch memory for the whole kernel -- what the compiler did for the larger targets */
        auto transition = [&](T *z, T ln_u) __attribute__((always_inline)) {
            /* accepts are counted inside the step (one add under the accept mask) and the wave's total is summed once at
             * the end: a select, a ballot + s_bcnt1 and a 64-bit scalar add per transition were six issue slots */
            if (SAMPLER == MM_SAMPLER_HMC)
                (void)mm_hmc_step_noise<T, Tgt, LCT>(a.P, a.scale, a.n_leapfrog, x, &lp, g, z, ln_u, mm_no_hook(), &n_acc32);
            else
                (void)mm_mh_step_noise<T, Tgt>(a.P, a.scale, x, &lp, z, ln_u, &n_acc32);
        };
        /* An odd count ends with half a pair: the second transition of the last pair is handed ln u = NaN, against which
         * both accept rules (`>`, `>=`) are false, so it leaves state and counters alone; then the noise is drawn afresh
         * for (it + 1, it + 2).  One pair loop, two inlined transitions, no tail code. */
        const T never = (T)MM_NAN_F;
        /* n transitions without output */
        auto silent = [&](unsigned int n) __attribute__((always_inline)) {
            for (unsigned int i = 0; i < n; i += 2) {
                T zna[D], znb[D], lnna, lnnb;
                mm_draw_noise_pair<D, Tab, true, MHP>(a.seed, chain, it + 2u, zna, &lnna, znb, &lnnb, tab);
                transition(za, lna);
                transition(zb, (i + 1 < n) ? lnb : never);
                MM_UNROLL
                for (int k = 0; k < D; ++k) {
                    za[k] = zna[k];
                    zb[k] = znb[k];
                }
                lna = lnna;
                lnb = lnnb;
                it += 2u;
            }
            if (n & 1u) {
                it -= 1u;
                mm_draw_noise_pair<D, Tab, true, MHP>(a.seed, chain, it, za, &lna, zb, &lnb, tab);
            }
        };
        silent(a.n_discard + (a.out ? 0u : a.n_collect));
        if (a.out) {
            T *const row = tile + lane * STRIDE;
            for (unsigned int rows_out = 0; rows_out < a.n_collect;) {
                const unsigned int nt = min((unsigned int)TILE_T, a.n_collect - rows_out);
                T *dst = row;
                for (unsigned int i = 0; i < nt; i += 2) {
                    T zna[D], znb[D], lnna, lnnb;
                    mm_draw_noise_pair<D, Tab, true, MHP>(a.seed, chain, it + 2u, zna, &lnna, znb, &lnnb, tab);
                    transition(za, lna);
                    MM_UNROLL
                    for (int k = 0; k < D; ++k)
                        dst[k] = x[k];
                    transition(zb, (i + 1 < nt) ? lnb : never);
                    if (TILE_T % 2 == 0 || i + 1 < nt) { /* row nt of an odd tile: inside an even TILE_T, never flushed */
                        MM_UNROLL
                        for (int k = 0; k < D; ++k)
                            dst[D + k] = x[k];
                    }
                    dst += 2 * D;
                    MM_UNROLL
                    for (int k = 0; k < D; ++k) {
                        za[k] = zna[k];
                        zb[k] = znb[k];
                    }
                    lna = lnna;
                    lnb = lnnb;
                    it += 2u;
                }
                if (nt & 1u) {
                    it -= 1u;
                    if (rows_out + nt < a.n_collect)
                        mm_draw_noise_pair<D, Tab, true, MHP>(a.seed, chain, it, za, &lna, zb, &lnb, tab);
                }
                mm_flush_tile<T, D>(a, tile, lane, wave_c0, (unsigned long long)a.out_t0 + rows_out, nt);
                rows_out += nt;
            }
        }
        n_acc += n_acc32;
    } else {
        for (unsigned int i = 0; i < a.n_discard; ++i)
            step();
        unsigned int tcol = 0, rows_out = 0;
        const unsigned int cap = (unsigned int)TILE_T;
        for (unsigned int t = 0; t < a.n_collect; ++t) {
            step();
            if (a.out) {
                MM_UNROLL
                for (int i = 0; i < D; ++i)
                    tile[lane * STRIDE + tcol * D + i] = x[i];
                ++tcol;
                if (tcol == cap || t + 1 == a.n_collect) {
                    mm_flush_tile<T, D>(a, tile, lane, wave_c0, (unsigned long long)a.out_t0 + rows_out, tcol);
                    rows_out += tcol;
                    tcol = 0;
                }
            }
        }
    }

    if (active) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = x[i];
        if (a.accept)
            a.accept[c] += n_acc;
    }
    if (a.accept_total) {
        unsigned long long wave_acc = active ? n_acc : 0ull;
        MM_UNROLL
        for (int off = 32; off > 0; off >>= 1)
            wave_acc += __shfl_xor(wave_acc, off);
        if (lane == 0 && wave_acc)
            atomicAdd(a.accept_total, wave_acc);
    }
}

/* the kernel proper; the body is a device function so that a run-time compiled translation unit (user targets,
 * mm_rtc.hip) can wrap it in an extern "C" kernel of its own */
template <class T, class Tgt, int SAMPLER, int PIPE = 0, int LCT = 0>
__global__ __launch_bounds__(256) void mm_run_kernel(const mm_run_args<T> a)
{
    mm_run_kernel_body<T, Tgt, SAMPLER, PIPE, LCT>(a);
}

/* BatchedGradientTarget::unnorm_logp_batch / unnorm_logp_and_grad for n rows (parity tests) */
template <class T, class Tgt>
__global__ void mm_logp_grad_kernel(const mm_tparams<T> P, const T *x, T *logp, T *grad, unsigned long long n)
{
    constexpr int D = Tgt::dim;
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    T xv[D], gv[D];
    MM_UNROLL
    for (int k = 0; k < D; ++k)
        xv[k] = x[i * D + k];
    if (grad) {
        logp[i] = Tgt::logp_grad(P, xv, gv);
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            grad[i * D + k] = gv[k];
    } else {
        logp[i] = Tgt::logp(P, xv);
    }
}

/* the noise the sampling kernels draw for (chain, iteration) (parity tests) */
template <class T, int D>
__global__ void mm_noise_kernel(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration,
                                unsigned long long n, T *z, T *u)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    T zz[D], uu;
    mm_draw_noise<D>(seed, chain_offset + i, iteration, zz, &uu);
    MM_UNROLL
    for (int k = 0; k < D; ++k)
        z[i * D + k] = zz[k];
    u[i] = uu;
}

#if !defined(__HIPCC_RTC__)
/* host-side launchers + dispatch record for one (sampler, dtype, kind, dim) instance */
template <class T, class Tgt, int SAMPLER, int PIPE = 0, int LCT = 0>
hipError_t mm_launch_run(const mm_run_args<T> &a, unsigned int grid, unsigned int block, hipStream_t stream)
{
    const size_t lds = mm_tile<T, Tgt::dim>::lds_bytes_table + (size_t)(block / 64) * mm_tile<T, Tgt::dim>::lds_bytes_per_wave;
    hipLaunchKernelGGL((mm_run_kernel<T, Tgt, SAMPLER, PIPE, LCT>), dim3(grid), dim3(block), lds, stream, a);
    return hipGetLastError();
}

template <class T, class Tgt>
hipError_t mm_launch_logp_grad(const mm_tparams<T> &P, const T *x, T *logp, T *grad, unsigned long long n,
                               hipStream_t stream)
{
    const unsigned int block = 256;
    const unsigned int grid = (unsigned int)((n + block - 1) / block);
    hipLaunchKernelGGL((mm_logp_grad_kernel<T, Tgt>), dim3(grid), dim3(block), 0, stream, P, x, logp, grad, n);
    return hipGetLastError();
}

template <class T, int D>
hipError_t mm_launch_noise(unsigned long long seed, unsigned long long chain_offset, unsigned int iteration,
                           unsigned long long n, T *z, T *u, hipStream_t stream)
{
    const unsigned int block = 256;
    const unsigned int grid = (unsigned int)((n + block - 1) / block);
    hipLaunchKernelGGL((mm_noise_kernel<T, D>), dim3(grid), dim3(block), 0, stream, seed, chain_offset, iteration, n,
                       z, u);
    return hipGetLastError();
}

template <class T> struct mm_kernel_entry {
    int kind, dim;
    hipError_t (*run_mh)(const mm_run_args<T> &, unsigned int, unsigned int, hipStream_t);
    hipError_t (*run_hmc)(const mm_run_args<T> &, unsigned int, unsigned int, hipStream_t);
    hipError_t (*run_mh_pp)(const mm_run_args<T> &, unsigned int, unsigned int, hipStream_t);  /* PIPE = 2 */
    hipError_t (*run_hmc_pp)(const mm_run_args<T> &, unsigned int, unsigned int, hipStream_t); /* PIPE = 2 */
    hipError_t (*run_hmc_pp10)(const mm_run_args<T> &, unsigned int, unsigned int, hipStream_t); /* PIPE = 2, L = 10 */
    hipError_t (*logp_grad)(const mm_tparams<T> &, const T *, T *, T *, unsigned long long, hipStream_t);
    size_t lds_bytes_per_wave;
    /* noise waves + transition waves, two waves per SIMD (mm_split_kernels.h); null above dim 8 */
    hipError_t (*run_mh_split)(const mm_run_args<T> &, hipStream_t);
    hipError_t (*run_hmc_split)(const mm_run_args<T> &, hipStream_t);
    hipError_t (*run_hmc_split10)(const mm_run_args<T> &, hipStream_t); /* L = 10 unrolled */
};

template <class T> struct mm_noise_entry {
    int dim;
    hipError_t (*noise)(unsigned long long, unsigned long long, unsigned int, unsigned long long, T *, T *,
                        hipStream_t);
};

/* defined in mm_inst_f32.hip / mm_inst_f64.hip */
const mm_kernel_entry<float> *mm_kernel_table_f32(int *n);
const mm_kernel_entry<double> *mm_kernel_table_f64(int *n);
const mm_noise_entry<float> *mm_noise_table_f32(int *n);
const mm_noise_entry<double> *mm_noise_table_f64(int *n);

#endif /* !__HIPCC_RTC__ */

#endif /* MM_KERNELS_H */
