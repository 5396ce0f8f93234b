/*
 * mm_nuts_kernels.h -- NUTS on the GPU: one chain per lane, whole run in one launch.
 *
 * Replaces NUTS::run (nuts.rs:163-170: rayon par_iter_mut over per-chain structs) and NUTSChain::run /
 * run_progress stepping (nuts.rs:457-471, 491-522).  Each lane carries its chain's position and adaptation state
 * in registers across all transitions; trees of different chains have different depths, so lanes of a wave diverge
 * inside a transition and re-converge at its end (samples are staged and flushed per wave exactly as in
 * mm_run_kernel).  The pending-subtree stack of mm_nuts_step lives in LDS when it fits (lane-interleaved), else in
 * a global scratch area with the same interleaving.
 * D-vectors stay in VGPRs (fine for D <= 8; at D = 32 the compiler spills): the dense f64 Gaussian of
 * BASELINE.json config 5 has its own lane-group / MFMA kernel, mm_nuts_lg.h.
 */
#ifndef MM_NUTS_KERNELS_H
#define MM_NUTS_KERNELS_H

#include "mm_kernels.h"
#include "mm_nuts.h"

template <class TT, class ST> struct mm_nuts_args {
    mm_tparams<TT> P;
    TT *state;                 /* [C, D] */
    mm_nuts_adapt<ST> *adapt;  /* [C] */
    TT *out;                   /* [C, n_total, D] or NULL */
    unsigned long long *n_leapfrog; /* [C] running totals, or NULL */
    unsigned int *depth_hist;  /* [MM_NUTS_JMAX + 1] histogram of tree depths (atomics), or NULL */
    unsigned long long n_chains, seed, chain_offset, n_total;
    unsigned int m0;           /* global step count before this launch (self.m) */
    unsigned int n_pre;        /* transitions without output */
    unsigned int n_rec;        /* transitions whose result is written */
    unsigned int write_initial; /* write the current position as row out_t0 first (nuts.rs:534) */
    unsigned int out_t0;
    unsigned int n_discard;    /* adaptation horizon: epsilon adapts while m <= n_discard (nuts.rs:682) */
    int max_depth;
    ST target_accept_p;
    int stack_in_lds;
    unsigned int async_batch;  /* mm_nuts_async_kernel: lanes between transitions wait for this many (0 = default) */
    unsigned char *scratch;    /* global stack storage when !stack_in_lds: per wave mm_nuts_stack_bytes<>() */
};

template <class TT, class ST, int D> struct mm_nuts_stack_layout {
    using S = mm_nuts_stack<TT, ST, D>;
    /* per-wave bytes, regions 16-byte aligned: vec | alpha | cnt */
    static constexpr size_t vec_bytes = ((size_t)S::vec_slots * 64 * sizeof(TT) + 15) / 16 * 16;
    static constexpr size_t alpha_bytes = ((size_t)S::alpha_slots * 64 * sizeof(ST) + 15) / 16 * 16;
    static constexpr size_t cnt_bytes = ((size_t)S::cnt_slots * 64 * sizeof(uint32_t) + 15) / 16 * 16;
    static constexpr size_t bytes = vec_bytes + alpha_bytes + cnt_bytes;
    /* the same three regions for the first `levels` stack entries only (a run never uses more than max_depth) */
    __host__ __device__ static constexpr size_t bytes_for(int levels)
    {
        return (size_t)levels * 64 * (3 * D * sizeof(TT) + sizeof(ST) + 3 * sizeof(uint32_t));
    }
    __device__ static S make_levels(unsigned char *base, int lane, int levels)
    {
        S s;
        s.vec = reinterpret_cast<TT *>(base) + lane;
        s.alpha = reinterpret_cast<ST *>(base + (size_t)levels * 64 * 3 * D * sizeof(TT)) + lane;
        s.cnt = reinterpret_cast<uint32_t *>(base + (size_t)levels * 64 * (3 * D * sizeof(TT) + sizeof(ST))) + lane;
        s.stride = 64;
        return s;
    }
    __device__ static S make(unsigned char *base, int lane)
    {
        S s;
        s.vec = reinterpret_cast<TT *>(base) + lane;
        s.alpha = reinterpret_cast<ST *>(base + vec_bytes) + lane;
        s.cnt = reinterpret_cast<uint32_t *>(base + vec_bytes + alpha_bytes) + lane;
        s.stride = 64;
        return s;
    }
};

/* init_chain for every chain (nuts.rs:528-545): epsilon search on the first run, mu = ln(10 eps) */
template <class TT, class ST> struct mm_nuts_init_args { /* the same as one block, for the kernels of user targets (mm_rtc.hip) */
    mm_tparams<TT> P;
    const TT *state;
    mm_nuts_adapt<ST> *adapt;
    unsigned long long n_chains, seed, chain_offset;
    ST eps_tol;
};
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
__device__ __forceinline__ void mm_nuts_init_body(const mm_tparams<TT> &P, const TT *state, mm_nuts_adapt<ST> *adapt,
                                                  unsigned long long n_chains, unsigned long long seed,
                                                  unsigned long long chain_offset, ST eps_tol)
{
    constexpr int D = Tgt::dim;
    const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains)
        return;
    TT x[D];
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = state[c * D + i];
    mm_nuts_adapt<ST> ad = adapt[c];
    mm_nuts_init_chain<TT, ST, Tgt, Red>(P, x, &ad, eps_tol, seed, chain_offset + c);
    adapt[c] = ad;
}
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
__global__ void mm_nuts_init_kernel(const mm_tparams<TT> P, const TT *state, mm_nuts_adapt<ST> *adapt,
                                    unsigned long long n_chains, unsigned long long seed,
                                    unsigned long long chain_offset, ST eps_tol)
{
    mm_nuts_init_body<TT, ST, Tgt, Red>(P, state, adapt, n_chains, seed, chain_offset, eps_tol);
}

/* LDS_STACK is a template parameter, not a run-time choice: a pointer that may be LDS or HBM is a generic pointer and
 * every stack access a flat_load / flat_store (which is what these kernels did at first, stack "in LDS" included). */
template <class TT, class ST, class Tgt, bool LDS_STACK>
__device__ __forceinline__ void mm_nuts_run_body(const mm_nuts_args<TT, ST> &a)
{
    constexpr int D = Tgt::dim;
    using Tile = mm_tile<TT, D>;
    using Lay = mm_nuts_stack_layout<TT, ST, D>;
    constexpr int TILE_T = Tile::tile_t, STRIDE = Tile::stride;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    TT *tile = reinterpret_cast<TT *>(mm_lds_raw);
    const int lane = threadIdx.x & 63;
    /* the wave's depth histogram, flushed once (a global atomic per chain and transition, all on a dozen addresses,
     * serialises in L2: it was 90 % of this kernel's time on small targets) */
    __shared__ unsigned int hist_lds[MM_NUTS_JMAX + 1];
    if (lane <= MM_NUTS_JMAX)
        hist_lds[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long wave_c0 = (unsigned long long)blockIdx.x * 64;
    const unsigned long long c = wave_c0 + lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;

    const mm_nuts_stack<TT, ST, D> stk =
        LDS_STACK ? Lay::make(mm_lds_raw + (Tile::lds_bytes_per_wave + 15) / 16 * 16, lane)
                  : Lay::make(a.scratch + (size_t)blockIdx.x * Lay::bytes, lane);

    TT x[D];
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = active ? a.state[c * D + i] : TT(0);
    mm_nuts_adapt<ST> ad;
    if (active) {
        ad = a.adapt[c];
    } else {
        ad.epsilon = ST(0.1);
        ad.epsilon_bar = ST(1);
        ad.h_bar = ST(0);
        ad.mu = ST(0);
    }
    unsigned long long n_lf = 0;
    unsigned int m = a.m0;
    unsigned int tcol = 0, rows_out = 0;

    auto record = [&](bool last) {
        if (!a.out)
            return;
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            tile[lane * STRIDE + tcol * D + k] = x[k];
        ++tcol;
        if (tcol == (unsigned int)TILE_T || last) {
            mm_flush_tile_raw<TT, D>(a.out, a.n_total, a.n_chains, tile, lane, wave_c0,
                                     (unsigned long long)a.out_t0 + rows_out, tcol);
            rows_out += tcol;
            tcol = 0;
        }
    };

    if (a.write_initial)
        record(a.n_rec == 0);
    const unsigned int total = a.n_pre + a.n_rec;
    for (unsigned int t = 0; t < total; ++t) {
        ++m;
        /* inactive lanes take no transition: an arbitrary state could grow an arbitrarily deep tree */
        if (active) {
            const mm_nuts_info inf = mm_nuts_step<TT, ST, Tgt>(a.P, x, &ad, m, a.n_discard, a.target_accept_p,
                                                               a.max_depth, a.seed, chain, stk);
            n_lf += inf.n_leapfrog;
            if (a.depth_hist)
                atomicAdd(&hist_lds[inf.depth < MM_NUTS_JMAX ? inf.depth : MM_NUTS_JMAX], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        if (t >= a.n_pre)
            record(t + 1 == total);
    }

    __builtin_amdgcn_wave_barrier();
    if (a.depth_hist && lane <= MM_NUTS_JMAX && hist_lds[lane] != 0u)
        atomicAdd(&a.depth_hist[lane], hist_lds[lane]);
    if (active) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = x[i];
        a.adapt[c] = ad;
        if (a.n_leapfrog)
            a.n_leapfrog[c] += n_lf;
    }
}
template <class TT, class ST, class Tgt, bool LDS_STACK>
__global__ __launch_bounds__(64) void mm_nuts_run_kernel(const mm_nuts_args<TT, ST> a)
{
    mm_nuts_run_body<TT, ST, Tgt, LDS_STACK>(a);
}

#define MM_NUTS_ASYNC_LDS_LIMIT (40u * 1024u) /* four waves per CU must fit 160 KB */

/* The same run with lanes that advance independently ("asynchronous lanes", kernel variant 4).
 *
 * In mm_nuts_run_kernel a transition ends for the wave when its deepest tree ends: with trees of 2 .. 1024 leaves the
 * lanes sit idle most of the time (RosenbrockND(3), 65 536 chains: 22 leaves per transition on average, ~350 for
 * the deepest of 64 lanes => 6 % of the lane-slots do work).  Here every tick of the wave gives every lane ONE leaf
 * of ITS tree (mm_nuts_tree::leaf_step); a lane whose doubling or transition completes does the bookkeeping under the
 * exec mask and begins its next doubling / transition in the same tick.  The noise is a function of (chain, m), so the
 * results do not depend on when a lane gets there: bit-identical to the synchronous kernel and to the host build.
 * Samples are written by the owning lane (rows complete at different times, so there is no tile to coalesce). */
template <class TT, class ST, class Tgt, bool LDS_STACK>
__global__ __launch_bounds__(64) void mm_nuts_async_kernel(const mm_nuts_args<TT, ST> a)
{
    constexpr int D = Tgt::dim;
    using Lay = mm_nuts_stack_layout<TT, ST, D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int hist_lds[MM_NUTS_JMAX + 1]; /* see mm_nuts_run_kernel */
    if (lane <= MM_NUTS_JMAX)
        hist_lds[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long c = (unsigned long long)blockIdx.x * 64 + lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;
    /* no output tile here, so the stack (sized for max_depth levels) fits LDS in more cases than in the kernel above */
    const mm_nuts_stack<TT, ST, D> stk = LDS_STACK ? Lay::make_levels(mm_lds_raw, lane, a.max_depth)
                                                   : Lay::make(a.scratch + (size_t)blockIdx.x * Lay::bytes, lane);

    TT x[D];
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = active ? a.state[c * D + i] : TT(0);
    mm_nuts_adapt<ST> ad;
    if (active) {
        ad = a.adapt[c];
    } else {
        ad.epsilon = ST(0.1);
        ad.epsilon_bar = ST(1);
        ad.h_bar = ST(0);
        ad.mu = ST(0);
    }
    const unsigned int total = a.n_pre + a.n_rec;
    TT *const rows = a.out ? a.out + (c * a.n_total + a.out_t0) * D : nullptr; /* this chain's first row */
    if (a.write_initial && rows && active) {
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            rows[k] = x[k];
    }
    const unsigned int first = a.write_initial ? 1u : 0u;

    mm_nuts_tree<TT, ST, Tgt> T;
    unsigned long long n_lf = 0;
    unsigned int m = a.m0, done = 0;
    int phase = 0; /* 0: between transitions, 1: begin a doubling, 2: next leaf, 3: hand the subtree up */
    bool live = active && total > 0;
    bool closing = false; /* phase 0 with a finished transition still to be closed (finish + record) */
    /* measured at 65 536 chains (tools/nuts_small_d.py): 8 is best for D <= 4; at D = 8 waiting does not pay */
    const unsigned int batch = a.async_batch ? a.async_batch : (D <= 4 ? 8u : 1u);
    while (__ballot(live) != 0ull) {
        /* Closing a transition and opening the next (dual averaging: six f64 transcendentals; momentum draw; logp +
         * gradient) is the longest section and on average only one or two lanes want it per tick: lanes wait in phase
         * 0 until `batch` of them do, or until nobody has tree work left. */
        const bool turn = live && phase == 0;
        const unsigned long long turn_mask = __ballot(turn);
        if ((unsigned int)__popcll(turn_mask) >= batch || __ballot(live && phase != 0) == 0ull) {
            if (turn) {
                if (closing) {
                    T.finish(&ad, a.n_discard, a.target_accept_p);
                    n_lf += T.info.n_leapfrog;
                    if (a.depth_hist)
                        atomicAdd(&hist_lds[T.info.depth < MM_NUTS_JMAX ? T.info.depth : MM_NUTS_JMAX], 1u);
                    if (done >= a.n_pre && rows) {
                        TT *r = rows + (size_t)(first + (done - a.n_pre)) * D;
                        MM_UNROLL
                        for (int k = 0; k < D; ++k)
                            r[k] = x[k];
                    }
                    ++done;
                    closing = false;
                    live = done < total;
                }
                if (live) {
                    ++m;
                    T.begin(a.P, x, m, a.seed, chain);
                    phase = 1;
                }
            }
        }
        /* the window of auxiliary uniforms moves here, for all lanes that need it at once (mm_nuts_tree::aux): a lane
         * draws at most three uniforms per tick (begin / double_begin / one merge / double_end, never all four) */
        if (live && phase != 0 && T.aux_stale())
            T.aux_advance(a.seed, chain);
        if (live && phase != 0) {
            if (phase == 1) {
                T.template double_begin<true>(ad, a.seed, chain);
                phase = 2;
            }
            if (phase == 2) {
                T.leaf(a.P);
                phase = 3;
            }
            /* ONE step up per tick: the number of merges a leaf completes differs from lane to lane (the trailing
             * ones of its index), and a wave that waited for the longest chain of merges would idle again */
            const int r = T.template hand_up<true>(a.seed, chain, stk);
            if (r == T.HAND_NEXT_LEAF)
                phase = 2;
            if (r == T.HAND_DONE) {
                T.template double_end<true>(x, a.seed, chain, a.max_depth);
                phase = 1;
                if (!T.s) {
                    phase = 0;
                    closing = true;
                }
            }
        }
    }

    __builtin_amdgcn_wave_barrier();
    if (a.depth_hist && lane <= MM_NUTS_JMAX && hist_lds[lane] != 0u)
        atomicAdd(&a.depth_hist[lane], hist_lds[lane]);
    if (active) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = x[i];
        a.adapt[c] = ad;
        if (a.n_leapfrog)
            a.n_leapfrog[c] += n_lf;
    }
}

/* Asynchronous lanes, LEAVES IN PAIRS (kernel variant 5, the default for D <= 8).
 *
 * What the kernel above pays per leaf (SQ counters on RosenbrockND(3), 65 536 chains: ~700 vector + 210 scalar instructions
 * per tick, TWO ticks per leaf): a tick gives a lane one leaf and ONE step of handing the subtree up, and a leaf needs
 * (trailing ones of its index) + 1 such steps -- two on average, the second a mere push onto the stack --, while the wave
 * executes every section (leaf, merge, doubling begin / end, the uniforms' Philox block) in every tick because some lane
 * always wants it.  Here a tick is a PAIR of leaves:
 *   * the first leaf of a pair waits for its sibling in registers (a one-leaf subtree is its own first leaf and its own
 *     proposal: (x, p), alpha, n'), the level-0 merge runs on registers (mm_nuts_tree::pair_first / pair_second): level 0
 *     of the pending-subtree stack, half of all pushes and merges, never touches LDS;
 *   * after the pair one merge at level >= 1 (the step that draws a uniform) and then every step that draws none -- the
 *     push, a first child with s' = 0 moving up -- in the same tick (hand_up_free): 0.75 ticks per leaf instead of 2;
 *   * the auxiliary uniforms come from a ring of eight per lane in LDS, refilled for all lanes that have room whenever one
 *     lane runs low (a Philox block = two uniforms serves most of the wave each time it is evaluated, where the two-block
 *     register window above is advanced for the few lanes that ran dry), three are fetched per tick and handed to the
 *     tree's steps in order (double_begin_with / pair_second / hand_up_with / double_end_with: at most three per tick).
 * A chain goes through exactly the operations of mm_nuts_step in the same order (mm_nuts_step_pairs is this walk on the
 * host, compared bit for bit in tests/test_nuts_parity_cpu.py): samples, adaptation state, leapfrog counts and depth
 * histogram are identical to variants 0 and 4 and to the host build. */
#define MM_NUTS_RING 8 /* auxiliary uniforms per lane in LDS */
/* -DMM_NUTS_PROFILE (tools/nuts_pair_probe.hip): per-wave event counters, written over the first lanes' leapfrog totals */
#ifdef MM_NUTS_PROFILE
#define MM_NP_DECL unsigned long long np_[16] = {}
#define MM_NP_ADD(i, v) (np_[i] += (unsigned long long)(v))
#define MM_NP_LANES(i, cond) (np_[i] += (unsigned long long)__popcll(__ballot(cond)))
#else
#define MM_NP_DECL
#define MM_NP_ADD(i, v) ((void)0)
#define MM_NP_LANES(i, cond) ((void)0)
#endif
/* the kernel proper is a device function so that a run-time compiled unit (csrc/mm_rtc.hip: user targets, built-in targets
 * at dimensions without a compiled instance) can wrap it in an extern "C" kernel of its own, like mm_nuts_run_body */
template <class TT, class ST, class Tgt, bool LDS_STACK>
__device__ __forceinline__ void mm_nuts_pair_body(const mm_nuts_args<TT, ST> &a)
{
    constexpr int D = Tgt::dim;
    using Lay = mm_nuts_stack_layout<TT, ST, D>;
    using Tree = mm_nuts_tree<TT, ST, Tgt>;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int hist_lds[MM_NUTS_JMAX + 1]; /* see mm_nuts_run_kernel */
    if (lane <= MM_NUTS_JMAX)
        hist_lds[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long c = (unsigned long long)blockIdx.x * 64 + lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;
    double *const ring = reinterpret_cast<double *>(mm_lds_raw) + lane; /* slot s of this lane: ring[64 s] */
    constexpr size_t ring_bytes = (size_t)MM_NUTS_RING * 64 * sizeof(double);
    const mm_nuts_stack<TT, ST, D> stk = LDS_STACK ? Lay::make_levels(mm_lds_raw + ring_bytes, lane, a.max_depth)
                                                   : Lay::make(a.scratch + (size_t)blockIdx.x * Lay::bytes, lane);

    TT x[D];
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = active ? a.state[c * D + i] : TT(0);
    mm_nuts_adapt<ST> ad;
    if (active) {
        ad = a.adapt[c];
    } else {
        ad.epsilon = ST(0.1);
        ad.epsilon_bar = ST(1);
        ad.h_bar = ST(0);
        ad.mu = ST(0);
    }
    const unsigned int total = a.n_pre + a.n_rec;
    TT *const rows = a.out ? a.out + (c * a.n_total + a.out_t0) * D : nullptr; /* this chain's first row */
    if (a.write_initial && rows && active) {
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            rows[k] = x[k];
    }
    const unsigned int first = a.write_initial ? 1u : 0u;

    Tree T;
    unsigned long long n_lf = 0;
    unsigned int m = a.m0, done = 0;
    unsigned int aux_p = 0; /* uniforms of transition T.m produced so far (even); T.aux_k: consumed */
    T.aux_k = 0;
    int phase = 0; /* 0: between transitions, 1: begin a doubling, 2: next pair of leaves, 4: hand the subtree up */
    bool live = active && total > 0;
    bool closing = false;
    const unsigned int batch = a.async_batch ? a.async_batch : (D <= 4 ? 8u : 1u);
    MM_NP_DECL;
    while (__ballot(live) != 0ull) {
        const bool turn = live && phase == 0;
        const unsigned long long turn_mask = __ballot(turn);
        MM_NP_ADD(0, 1);            /* ticks */
        MM_NP_LANES(1, turn);       /* lane-ticks spent waiting between transitions */
        MM_NP_LANES(2, live);       /* lane-ticks alive */
        if ((unsigned int)__popcll(turn_mask) >= batch || __ballot(live && phase != 0) == 0ull) {
            MM_NP_ADD(3, 1);        /* transition sections executed */
            MM_NP_LANES(4, turn);   /* lanes served by them */
            if (turn) {
                if (closing) {
                    T.finish(&ad, a.n_discard, a.target_accept_p);
                    n_lf += T.info.n_leapfrog;
                    if (a.depth_hist)
                        atomicAdd(&hist_lds[T.info.depth < MM_NUTS_JMAX ? T.info.depth : MM_NUTS_JMAX], 1u);
                    if (done >= a.n_pre && rows) {
                        TT *r = rows + (size_t)(first + (done - a.n_pre)) * D;
                        MM_UNROLL
                        for (int k = 0; k < D; ++k)
                            r[k] = x[k];
                    }
                    ++done;
                    closing = false;
                    live = done < total;
                }
                if (live) {
                    ++m;
                    T.m = m;
                    double q[4];
                    T.aux_load(a.seed, chain, 0, q);
                    T.aux_load(a.seed, chain, 1, q + 2);
                    ring[64 * 1] = q[1];
                    ring[64 * 2] = q[2];
                    ring[64 * 3] = q[3];
                    aux_p = 4;
                    T.aux_k = 1; /* the transition's first uniform is the slice variable's Exp(1) */
                    T.begin_with(a.P, x, a.seed, chain, q[0]);
                    phase = 1;
                }
            }
        }
        const bool work = live && phase != 0;
        /* keep three uniforms ahead of every working lane; whoever has room for a block takes one along */
        while (__ballot(work && aux_p - T.aux_k < 3u) != 0ull) {
            MM_NP_ADD(5, 1);        /* Philox blocks evaluated for the ring */
            MM_NP_LANES(6, work && aux_p - T.aux_k <= (unsigned int)MM_NUTS_RING - 2u); /* lanes that took one */
            if (work && aux_p - T.aux_k <= (unsigned int)MM_NUTS_RING - 2u) {
                double q[2];
                T.aux_load(a.seed, chain, aux_p >> 1, q);
                ring[64 * (aux_p & (MM_NUTS_RING - 1u))] = q[0];
                ring[64 * ((aux_p + 1u) & (MM_NUTS_RING - 1u))] = q[1];
                aux_p += 2u;
            }
        }
        if (work) {
            const unsigned int k0 = T.aux_k;
            const double d0 = ring[64 * (k0 & (MM_NUTS_RING - 1u))];
            const double d1 = ring[64 * ((k0 + 1u) & (MM_NUTS_RING - 1u))];
            const double d2 = ring[64 * ((k0 + 2u) & (MM_NUTS_RING - 1u))];
            unsigned int used = 0;
            /* The next of the tick's (at most three) draws, selected among VALUES.  As a closure with d0, d1, d2 and `used`
             * captured by reference -- handed to hand_up_with, which calls it inside its merge branch -- the compiler selected
             * among their ADDRESSES and loaded through the one it picked: the three draws and a table of three pointers lived in
             * private memory, five stores and a dependent scratch load in every tick of every chain (round 5,
             * tools/scratch_scan.py: 64 - 72 bytes of scratch in every instance of this kernel). */
            auto next_draw = [](unsigned int k, double a0, double a1, double a2) -> double {
                return k == 0u ? a0 : (k == 1u ? a1 : a2);
            };
            MM_NP_LANES(7, phase == 1); /* doublings begun */
            if (phase == 1) {
                T.double_begin_with(ad.epsilon, next_draw(used, d0, d1, d2));
                ++used;
                phase = 2;
            }
            int r = Tree::HAND_MORE; /* phase 4: a subtree on its way up */
            bool second = false;
            MM_NP_LANES(8, phase == 2); /* first leaves */
            if (phase == 2) {
                T.leaf(a.P);
                r = T.pair_first();
                second = r == Tree::HAND_NEXT_LEAF;
            }
            MM_NP_LANES(9, second);     /* second leaves */
            if (second) {
                T.leaf(a.P);
                r = T.pair_second(next_draw(used, d0, d1, d2));
                ++used;
            }
            MM_NP_LANES(10, r == Tree::HAND_MORE && T.sibling_waits(stk)); /* merges at level >= 1 */
            if (r == Tree::HAND_MORE && T.sibling_waits(stk)) {
                /* a waiting sibling is merged -- and the draw taken -- unless S has reached the doubling's own level */
                const double u3 = next_draw(used, d0, d1, d2);
                const bool draws = T.S_level != (uint32_t)T.j;
                r = T.hand_up_with(stk, [u3]() -> double { return u3; });
                used += draws ? 1u : 0u;
            }
            while (__ballot(r == Tree::HAND_MORE && !T.sibling_waits(stk)) != 0ull) {
                MM_NP_ADD(11, 1);       /* trips of the free-step loop */
                MM_NP_LANES(12, r == Tree::HAND_MORE && !T.sibling_waits(stk));
                if (r == Tree::HAND_MORE && !T.sibling_waits(stk))
                    r = T.hand_up_free(stk);
            }
            phase = r == Tree::HAND_MORE ? 4 : 2;
            MM_NP_LANES(13, r == Tree::HAND_DONE); /* doublings ended */
            if (r == Tree::HAND_DONE) {
                T.double_end_with(x, a.max_depth, next_draw(used, d0, d1, d2));
                ++used;
                phase = 1;
                if (!T.s) {
                    phase = 0;
                    closing = true;
                }
            }
            T.aux_k = k0 + used;
        }
    }

    __builtin_amdgcn_wave_barrier();
    if (a.depth_hist && lane <= MM_NUTS_JMAX && hist_lds[lane] != 0u)
        atomicAdd(&a.depth_hist[lane], hist_lds[lane]);
    if (active) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = x[i];
        a.adapt[c] = ad;
        if (a.n_leapfrog)
            a.n_leapfrog[c] += n_lf;
    }
#ifdef MM_NUTS_PROFILE
    if (a.n_leapfrog && lane < 16)
        a.n_leapfrog[c] = np_[lane];
#endif
}

template <class TT, class ST, class Tgt, bool LDS_STACK>
__global__ __launch_bounds__(64) void mm_nuts_pair_kernel(const mm_nuts_args<TT, ST> a)
{
    mm_nuts_pair_body<TT, ST, Tgt, LDS_STACK>(a);
}

#if !defined(__HIPCC_RTC__) /* host side: not part of the run-time compiled kernels of user targets (mm_rtc.hip) */
template <class TT, class ST, class Tgt>
hipError_t mm_launch_nuts_run_pair(const mm_nuts_args<TT, ST> &a, hipStream_t stream)
{
    using Lay = mm_nuts_stack_layout<TT, ST, Tgt::dim>;
    const unsigned int grid = (unsigned int)((a.n_chains + 63) / 64);
    const size_t ring = (size_t)MM_NUTS_RING * 64 * sizeof(double);
    const size_t need = ring + Lay::bytes_for(a.max_depth);
    if (need <= MM_NUTS_ASYNC_LDS_LIMIT)
        hipLaunchKernelGGL((mm_nuts_pair_kernel<TT, ST, Tgt, true>), dim3(grid), dim3(64), need, stream, a);
    else
        hipLaunchKernelGGL((mm_nuts_pair_kernel<TT, ST, Tgt, false>), dim3(grid), dim3(64), ring, stream, a);
    return hipGetLastError();
}

template <class TT, class ST, class Tgt>
hipError_t mm_launch_nuts_run_async(const mm_nuts_args<TT, ST> &a, hipStream_t stream)
{
    using Lay = mm_nuts_stack_layout<TT, ST, Tgt::dim>;
    const unsigned int grid = (unsigned int)((a.n_chains + 63) / 64);
    const size_t need = Lay::bytes_for(a.max_depth);
    if (need <= MM_NUTS_ASYNC_LDS_LIMIT)
        hipLaunchKernelGGL((mm_nuts_async_kernel<TT, ST, Tgt, true>), dim3(grid), dim3(64), need, stream, a);
    else
        hipLaunchKernelGGL((mm_nuts_async_kernel<TT, ST, Tgt, false>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
hipError_t mm_launch_nuts_init(const mm_tparams<TT> &P, const TT *state, mm_nuts_adapt<ST> *adapt,
                               unsigned long long n_chains, unsigned long long seed, unsigned long long chain_offset,
                               hipStream_t stream)
{
    const ST eps_tol = sizeof(ST) == 4 ? (ST)1.1920929e-7 : (ST)2.220446049250313e-16; /* T::epsilon() */
    const unsigned int block = 64;
    const unsigned int grid = (unsigned int)((n_chains + block - 1) / block);
    hipLaunchKernelGGL((mm_nuts_init_kernel<TT, ST, Tgt, Red>), dim3(grid), dim3(block), 0, stream, P, state, adapt,
                       n_chains, seed, chain_offset, eps_tol);
    return hipGetLastError();
}

template <class TT, class ST, class Tgt>
hipError_t mm_launch_nuts_run(const mm_nuts_args<TT, ST> &a, hipStream_t stream)
{
    using Lay = mm_nuts_stack_layout<TT, ST, Tgt::dim>;
    const unsigned int grid = (unsigned int)((a.n_chains + 63) / 64);
    size_t lds = (mm_tile<TT, Tgt::dim>::lds_bytes_per_wave + 15) / 16 * 16;
    if (a.stack_in_lds)
        hipLaunchKernelGGL((mm_nuts_run_kernel<TT, ST, Tgt, true>), dim3(grid), dim3(64), lds + Lay::bytes, stream, a);
    else
        hipLaunchKernelGGL((mm_nuts_run_kernel<TT, ST, Tgt, false>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

/* dispatch record: one (kind, dim) for one (TT, ST) pair */
template <class TT, class ST> struct mm_nuts_entry {
    int kind, dim;
    hipError_t (*init)(const mm_tparams<TT> &, const TT *, mm_nuts_adapt<ST> *, unsigned long long,
                       unsigned long long, unsigned long long, hipStream_t);
    hipError_t (*run)(const mm_nuts_args<TT, ST> &, hipStream_t);
    hipError_t (*run_async)(const mm_nuts_args<TT, ST> &, hipStream_t); /* asynchronous lanes; NULL for dim > 8 */
    hipError_t (*run_pair)(const mm_nuts_args<TT, ST> &, hipStream_t);  /* asynchronous lanes, leaves in pairs; NULL where MM_NUTS_PAIR_AT(dim) is false (mm_nuts_inst.inc) */
    size_t stack_bytes_per_wave, tile_bytes_per_wave;
};

const mm_nuts_entry<float, double> *mm_nuts_table_m0(int *n);
const mm_nuts_entry<float, float> *mm_nuts_table_m1(int *n);
const mm_nuts_entry<double, double> *mm_nuts_table_m2(int *n);

/* lane-group / MFMA kernels (mm_nuts_lg.h): GaussianND, f64 */
struct mm_nuts_lg_args;
struct mm_nuts_lg_entry {
    int dim;
    hipError_t (*init)(const mm_tparams<double> &, const double *, mm_nuts_adapt<double> *, unsigned long long,
                       unsigned long long, unsigned long long, hipStream_t);
    hipError_t (*run)(const mm_nuts_lg_args &, hipStream_t);            /* all transitions, one launch */
    hipError_t (*run_transition)(mm_nuts_lg_args, hipStream_t);         /* one transition, tree-depth compaction */
    hipError_t (*run_queue)(const mm_nuts_lg_args &, unsigned int, int, hipStream_t); /* persistent scheduler (waves, waves per SIMD) */
    size_t scratch_doubles_per_wave;
    size_t rec_doubles_per_chain;
};
const mm_nuts_lg_entry *mm_nuts_lg_table(int *n);
#endif /* !__HIPCC_RTC__ */

#endif /* MM_NUTS_KERNELS_H */
