/*
 * mm_nuts_lg.h -- NUTS for the dense Gaussian target in f64, lane-GROUP mapping with the gradient on the matrix cores
 * (BASELINE.json config 5: 32-D ill-conditioned Gaussian, f64).  Device only; its bit-exact host twin is
 * mm_nuts_step<double, double, mm_target_gnd_grp4<D>, mm_red_grp4<D>> (mm_nuts.h), which in turn is checked against
 * the recursive restatement of nuts.rs (tests/test_nuts_parity_cpu.py).
 *
 * Why not one chain per lane here: at D = 32 in f64 a chain's vectors (both trajectory edges with momenta and
 * gradients, the proposal, the working leaf) are ~350 doubles -- the lane-per-chain kernel spills 1200 registers --
 * and the gradient -A x is a 32 x 32 matrix-vector product per leapfrog step, the one dense contraction of the path.
 *
 * Mapping.  A wave = 16 chains x 4 lanes.  Lane l = (c = l & 15, q = l >> 4) owns the coordinates d = 4 s + q
 * (s = 0..D/4-1) of chain c.  That is simultaneously
 *   - the B-operand layout of v_mfma_f64_16x16x4_f64 for X^T (k = 4 s + q, column c) at k-step s, and
 *   - the D-result layout of the product tile (row 16 t + 4 r + q, column c) = coordinate s = 4 t + r,
 * so G^T = A . X^T (A in the A-operand layout, loaded once per kernel into registers) returns the gradient in the
 * registers' own distribution: no lane movement between leapfrog steps, D/16 * D/4 MFMAs per gradient for 16 chains.
 * The MFMA accumulates k in order with one rounding per product (tools/mfma_f64_check.hip), i.e. exactly the fma
 * chain of the host twin.  Dot products over a chain's coordinates: per-lane fma chain over s, then a butterfly over
 * the four lanes ((c0 + c1) + (c2 + c3), identical in all four) = mm_red_grp4.
 *
 * Tree building in lock-step.  All chains of a wave start a transition together and run the same doubling j and
 * the same leaf index at the same time; a chain whose subtree stopped (or whose transition ended) idles until the
 * wave's last chain is done -- the cost of SIMT here, bounded by the 16 chains of a wave instead of 64.  Because the
 * live chains share the leaf index, their pending-subtree stacks have the same shape and the level index is
 * wave-uniform.  A chain that fails early walks up the remaining levels at once (merging where it is a second
 * child, drawing the merge uniform, exactly as the recursion returns through nuts.rs:858-929).
 *
 * Cost model (tools/f64_rate.hip, tools/lg_profile.py).  On gfx950 the f64 matrix rate equals the f64 vector rate
 * (v_mfma_f64_16x16x4: 64 cycles; v_fma_f64 / v_add_f64: ~5 cycles per wave instruction at ANY occupancy) and the two
 * do not overlap -- 2 MFMA + 8 FMA take 128 + 43 cycles whether one or four waves share the SIMD.  A leaf therefore
 * costs 16 MFMA = 1024 cycles plus ~5 cycles for every vector instruction around it, and occupancy buys nothing:
 * the kernel runs one wave per SIMD with the 512-register budget and 40 KB of LDS, and the work is to keep the
 * instruction count per leaf down:
 *   - registers hold the edge being extended (x, p, g), the subtree proposal, the current sample and the A-operand
 *     blocks; everything else is addressed by wave-uniform indices and lives in memory, lane-interleaved (slot i of
 *     lane l at base[i * 64 + l]);
 *   - pending first children, per level k: proposal, alpha sum, counts (10 slots).  Levels 0..2 in LDS, deeper
 *     (touched every 2^(k+1) leaves) in an HBM scratch area;
 *   - "first leaf" table: the (x, p) of the leaf that starts a subtree.  The first leaf of the pending sibling at
 *     level k of leaf i is leaf i0 = i with its k + 1 low bits cleared, and its data is filed under c = ctz(i0)
 *     (leaf 0 under c = JMAX): every even leaf is written once, under the highest level it starts, instead of being
 *     copied from stack entry to stack entry.  c = 1..3 in LDS, the rest in HBM;
 *   - both trajectory edges (HBM): the one being extended is loaded at the start of a doubling and written back at
 *     its end;
 *   - the U-turn test needs no per-lane orientation: with d = x_cur - x_first, A = d.p_first, B = d.p_cur it is
 *     (A >= 0 and B >= 0) for v = +1 and (A <= 0 and B <= 0) for v = -1 -- negating every term of an fma chain negates
 *     the result exactly, so this equals the twin's (x_plus - x_minus).p_minus/plus >= 0 bit for bit;
 *   - the records of the level-0 merge of an odd leaf are requested before the leapfrog, so their latency hides
 *     behind the MFMAs.
 */
#ifndef MM_NUTS_LG_H
#define MM_NUTS_LG_H

#include <hip/hip_runtime.h>

#include "mm_nuts.h"


/* Section timers for tools/lg_profile.hip (s_memtime deltas of lane 0 per wave); compiled out of the product. */
#ifdef MM_LG_PROFILE
#define MM_LG_TICK(sec)                                                                                           \
    do {                                                                                                          \
        const unsigned long long _now = __builtin_amdgcn_s_memtime();                                             \
        mm_lg_prof_acc[sec] += _now - mm_lg_prof_t;                                                               \
        mm_lg_prof_t = _now;                                                                                      \
    } while (0)
#define MM_LG_COUNT(sec) (mm_lg_prof_acc[sec] += 1)
#else
#define MM_LG_TICK(sec) ((void)0)
#define MM_LG_COUNT(sec) ((void)0)
#endif

struct mm_nuts_lg_args {
    const double *mat;              /* precision matrix A, row-major [D, D] */
    double *state;                  /* [C, D] */
    mm_nuts_adapt<double> *adapt;   /* [C] */
    double *out;                    /* [C, n_total, D] or NULL */
    unsigned long long *n_leapfrog; /* [C] or NULL */
    unsigned int *depth_hist;       /* [MM_NUTS_JMAX + 1] or NULL */
    unsigned long long n_chains, seed, chain_offset, n_total;
    unsigned int m0, n_pre, n_rec, write_initial, out_t0, n_discard;
    int max_depth;
    double target_accept_p;
    double *scratch;                /* per wave: mm_lg_cfg<D>::scratch_doubles_per_wave */
#ifdef MM_LG_PROFILE
    unsigned long long *prof;       /* [waves][8] */
#endif
};

template <int D> struct mm_lg_cfg {
    static_assert(D % 16 == 0, "lane-group kernel: D must be a multiple of 16");
    static constexpr int NS = D / 4;           /* coordinates per lane */
    static constexpr int NT = D / 16;          /* 16-row result tiles */
    static constexpr int ES = NS + 2;          /* pending entry: proposal[NS], alpha, (n | n_alpha << 32) */
    static constexpr int FS = 2 * NS;          /* first-leaf record: x[NS], p[NS] */
#ifndef MM_LG_WAVES_PER_SIMD
#define MM_LG_WAVES_PER_SIMD 1
#endif
#if MM_LG_WAVES_PER_SIMD == 1
    static constexpr int LE = 3;               /* entry(k), k < LE, in LDS */
    static constexpr int LF = 3;               /* first(c), 1 <= c <= LF, in LDS */
#else
    static constexpr int LE = 2;
    static constexpr int LF = 1;
#endif
    static constexpr int lds_E = 0, lds_F = LE * ES, lds_slots = LE * ES + LF * FS;
    static constexpr size_t lds_bytes = (size_t)lds_slots * 64 * sizeof(double);
    /* HBM slots per wave: entry(k), k = LE..JMAX-1 | first(c), c = LF+1..JMAX | minus edge (x, p, g) | plus edge */
    static constexpr int hbm_E = 0, hbm_F = (MM_NUTS_JMAX - LE) * ES, hbm_G = hbm_F + (MM_NUTS_JMAX - LF) * FS,
                         hbm_slots = hbm_G + 6 * NS;
    static constexpr size_t scratch_doubles_per_wave = (size_t)hbm_slots * 64;
};

typedef double mm_d4 __attribute__((ext_vector_type(4)));
/* LDS is addressed through an explicitly address-space-3 pointer: where an accessor picks LDS or HBM by a uniform
 * index, same-typed generic pointers let the optimiser merge the two branches into one flat_load / flat_store on a
 * selected pointer (which then waits on both memory counters); distinct pointer types keep ds_* and global_* apart */
typedef __attribute__((address_space(3))) double mm_lds_double;

/* sum over the four lanes of a chain (lanes c, c + 16, c + 32, c + 48): (c0 + c1) + (c2 + c3), the same value in all
 * four.  v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- / 32-lane rows between two registers in the VALU
 * -- no LDS round trip as with ds_bpermute: with both operands = c they return {own row pair's first, second}, whose
 * sum is the butterfly step (IEEE addition commutes, so all four lanes get the same bits). */
__device__ __forceinline__ double mm_lg_group_sum(double c)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    unsigned int lo = (unsigned int)__double2loint(c), hi = (unsigned int)__double2hiint(c);
    u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    c = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    lo = (unsigned int)__double2loint(c);
    hi = (unsigned int)__double2hiint(c);
    l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

template <int NS> __device__ __forceinline__ double mm_lg_dot(const double *a, const double *b)
{
    double c = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        c = fma(a[s], b[s], c);
    return mm_lg_group_sum(c);
}

/* y = A x for the 16 chains of the wave; returns logp = -1/2 x.y and g = -y */
template <int D>
__device__ __forceinline__ double mm_lg_logp_grad(const double (&Aop)[D / 16][D / 4], const double *x, double *g)
{
    constexpr int NS = D / 4, NT = D / 16;
    double y[NS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[t][s], x[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            y[4 * t + r] = acc[r];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
        g[s] = -y[s];
    return -0.5 * mm_lg_dot<NS>(x, y);
}

template <int D> __global__ __launch_bounds__(64, MM_LG_WAVES_PER_SIMD) void mm_nuts_lg_kernel(const mm_nuts_lg_args a)
{
    using Cfg = mm_lg_cfg<D>;
    constexpr int NS = Cfg::NS, NT = Cfg::NT, ES = Cfg::ES;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];

    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    const unsigned long long cl = (unsigned long long)blockIdx.x * 16 + c;
    const bool active = cl < a.n_chains;
    const unsigned long long chain = a.chain_offset + cl;
    mm_lds_double *const lds = (mm_lds_double *)mm_lds_raw + lane;
    double *const scr = a.scratch + (size_t)blockIdx.x * Cfg::scratch_doubles_per_wave + lane;

    /* records of the merge at level k (wave-uniform k, cc): the sibling's entry and its first leaf */
    struct rec {
        double fx[NS], fp[NS], prime[NS], alpha, cnt;
    };
    auto load_rec = [&](int k, int cc, rec &r) {
        if (cc <= Cfg::LF) {
            const mm_lds_double *f = lds + (size_t)(Cfg::lds_F + (cc - 1) * Cfg::FS) * 64;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                r.fx[s] = f[s * 64];
                r.fp[s] = f[(NS + s) * 64];
            }
        } else {
            const double *f = scr + (size_t)(Cfg::hbm_F + (cc - Cfg::LF - 1) * Cfg::FS) * 64;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                r.fx[s] = f[s * 64];
                r.fp[s] = f[(NS + s) * 64];
            }
        }
        if (k < Cfg::LE) {
            const mm_lds_double *e = lds + (size_t)(Cfg::lds_E + k * ES) * 64;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                r.prime[s] = e[s * 64];
            r.alpha = e[NS * 64];
            r.cnt = e[(NS + 1) * 64];
        } else {
            const double *e = scr + (size_t)(Cfg::hbm_E + (k - Cfg::LE) * ES) * 64;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                r.prime[s] = e[s * 64];
            r.alpha = e[NS * 64];
            r.cnt = e[(NS + 1) * 64];
        }
    };
    /* where the first leaf of the sibling at level k of `leaf` is filed */
    auto first_slot = [](unsigned int leaf, int k) -> int {
        const unsigned int i0 = leaf & ~((2u << k) - 1u);
        return i0 ? (__ffs((int)i0) - 1) : MM_NUTS_JMAX;
    };

    /* A-operand blocks: lane (i = l & 15, k = l >> 4) holds A[16 t + i][4 s + k] */
    double Aop[NT][NS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            Aop[t][s] = a.mat[(size_t)(16 * t + c) * D + 4 * s + q];

    double x[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        x[s] = active ? a.state[cl * D + 4 * s + q] : 0.0;
    mm_nuts_adapt<double> ad;
    if (active) {
        ad = a.adapt[cl];
    } else {
        ad.epsilon = 0.1;
        ad.epsilon_bar = 1.0;
        ad.h_bar = 0.0;
        ad.mu = 0.0;
    }
    unsigned long long n_lf = 0;
    unsigned int m = a.m0;
    unsigned int rows_out = 0;
#ifdef MM_LG_PROFILE
    unsigned long long mm_lg_prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long mm_lg_prof_t = __builtin_amdgcn_s_memtime();
#endif

    auto record = [&]() {
        if (a.out && active) {
            double *dst = a.out + (cl * a.n_total + a.out_t0 + rows_out) * D;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                dst[4 * s + q] = x[s];
        }
        ++rows_out;
    };

    if (a.write_initial)
        record();
    const unsigned int total = a.n_pre + a.n_rec;
    for (unsigned int t = 0; t < total; ++t) {
        ++m;
        /* ---------------- one transition of the wave's 16 chains (nuts.rs:550-691) ---------------- */
        /* auxiliary uniforms: draw k of (chain, m) is a half of Philox block AUX + (k >> 1) (mm_rng.h); the block is
         * kept so that two consecutive draws cost one evaluation */
        unsigned int aux_k = 0, aux_have = 0xffffffffu;
        mm_u32x4 aux_blk;
        aux_blk.w[0] = aux_blk.w[1] = aux_blk.w[2] = aux_blk.w[3] = 0u;
        auto aux_peek = [&]() -> double {
            const unsigned int b = aux_k >> 1;
            if (b != aux_have) {
                aux_blk = mm_block(a.seed, chain, m, MM_AUX_BLOCK + b);
                aux_have = b;
            }
            return (aux_k & 1u) ? mm_u53(aux_blk.w[2], aux_blk.w[3]) : mm_u53(aux_blk.w[0], aux_blk.w[1]);
        };

        double joint, logu;
        {
            double p0[NS], grad[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int d = 4 * s + q;
                mm_u32x4 blk = mm_block(a.seed, chain, m, (uint32_t)(d >> 1));
                double z0, z1;
                mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
                p0[s] = (d & 1) ? z1 : z0;
            }
            const double ulogp = mm_lg_logp_grad<D>(Aop, x, grad);
            joint = ulogp - mm_lg_dot<NS>(p0, p0) * 0.5;
            const double exp1_obs = -mm_log(aux_peek());
            aux_k += 1;
            logu = joint - exp1_obs;
            /* both edges start at (x, p0, grad) */
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                scr[(Cfg::hbm_G + s) * 64] = x[s];
                scr[(Cfg::hbm_G + NS + s) * 64] = p0[s];
                scr[(Cfg::hbm_G + 2 * NS + s) * 64] = grad[s];
                scr[(Cfg::hbm_G + 3 * NS + s) * 64] = x[s];
                scr[(Cfg::hbm_G + 4 * NS + s) * 64] = p0[s];
                scr[(Cfg::hbm_G + 5 * NS + s) * 64] = grad[s];
            }
        }
        int j = 0;
        unsigned int n = 1;
        bool alive = active; /* the reference's `s` */
        double alpha = 0.0;
        unsigned int n_alpha = 0;
        int depth = 0;
        MM_LG_TICK(0);

        while (__ballot(alive) != 0ull) {
            const double u_run_1 = aux_peek();
            if (alive)
                aux_k += 1;
            const bool neg = !(u_run_1 < 0.5); /* v = -1 */
            /* the outer edge in direction v, advanced in place: after the doubling it IS the returned edge */
            double *const edge = scr + (size_t)(Cfg::hbm_G + (neg ? 0 : 3 * NS)) * 64;
            double *const other = scr + (size_t)(Cfg::hbm_G + (neg ? 3 * NS : 0)) * 64;
            double cx[NS], cp[NS], cg[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                cx[s] = edge[s * 64];
                cp[s] = edge[(NS + s) * 64];
                cg[s] = edge[(2 * NS + s) * 64];
            }
            const double epsv = neg ? -ad.epsilon : ad.epsilon;
            const double h = epsv * 0.5;
            const unsigned int n_leaves = 1u << j;
            /* nothing in flight when the leaf loop starts: otherwise every use of these loop-carried registers gets a
             * conservative vmcnt wait that also drains the records requested ahead of the leapfrog */
            __builtin_amdgcn_s_waitcnt(0);

            bool done = !alive;
            unsigned int S_n = 0, S_nalpha = 0;
            bool S_s = true;
            double S_alpha = 0.0;
            double S_prime[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                S_prime[s] = 0.0;

            MM_LG_TICK(1);
            for (unsigned int leaf = 0; leaf < n_leaves; ++leaf) {
                if (__ballot(!done) == 0ull)
                    break;
                MM_LG_COUNT(6);
                /* an odd leaf merges with its left neighbour first: request those records now */
                rec r0;
                const bool merge0 = j > 0 && (leaf & 1u);
                if (merge0)
                    load_rec(0, first_slot(leaf, 0), r0);
                /* leapfrog of the outer edge (nuts.rs:979-996), in place; chains that are done keep their edge (the
                 * matrix product runs for all 64 lanes: MFMA has no per-lane mask, their columns are recomputed) */
                if (!done) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        cp[s] = fma(h, cg[s], cp[s]);
                        cx[s] = fma(epsv, cp[s], cx[s]);
                    }
                }
                double y[NS];
                const double lp = mm_lg_logp_grad<D>(Aop, cx, y); /* y = -A x */
                if (!done) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        cg[s] = y[s];
                        cp[s] = fma(h, y[s], cp[s]);
                    }
                }
                const double jointp = lp - mm_lg_dot<NS>(cp, cp) * 0.5;
                if (!done) {
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        S_prime[s] = cx[s];
                    n_lf += 1;
                    S_n = (logu < jointp) ? 1u : 0u;
                    S_s = (logu - 1000.0) < jointp;
                    S_alpha = fmin(1.0, mm_exp(jointp - joint));
                    S_nalpha = 1;
                    /* an even leaf starts subtrees: file its (x, p) under the highest level it starts */
                    if (j > 0 && (leaf & 1u) == 0u) {
                        const int cc = leaf ? (__ffs((int)leaf) - 1) : MM_NUTS_JMAX;
                        if (cc <= Cfg::LF) {
                            mm_lds_double *f = lds + (size_t)(Cfg::lds_F + (cc - 1) * Cfg::FS) * 64;
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                f[s * 64] = cx[s];
                                f[(NS + s) * 64] = cp[s];
                            }
                        } else {
                            double *f = scr + (size_t)(Cfg::hbm_F + (cc - Cfg::LF - 1) * Cfg::FS) * 64;
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                f[s * 64] = cx[s];
                                f[(NS + s) * 64] = cp[s];
                            }
                        }
                    }
                }
                /* hand S up the implicit recursion; walking lanes have S at level k at the top of iteration k */
                bool walking = !done;
                MM_LG_TICK(2);
                /* a sibling waits at level k: S is the second child, merge (nuts.rs:900-928) */
                auto merge = [&](const rec &r) {
                    const double u = aux_peek();
                    /* stop criterion on (first leaf of the sibling, current leaf): d = x_cur - x_first */
                    double ca = 0.0, cb = 0.0;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const double d = cx[s] - r.fx[s];
                        ca = fma(d, r.fp[s], ca);
                        cb = fma(d, cp[s], cb);
                    }
                    ca = mm_lg_group_sum(ca);
                    cb = mm_lg_group_sum(cb);
                    const bool crit = neg ? (ca <= 0.0 && cb <= 0.0) : (ca >= 0.0 && cb >= 0.0);
                    if (walking) {
                        aux_k += 1;
                        const unsigned long long cnt = (unsigned long long)__double_as_longlong(r.cnt);
                        const unsigned int n1 = (unsigned int)cnt, na1 = (unsigned int)(cnt >> 32);
                        unsigned int den = n1 + S_n;
                        if (den < 1)
                            den = 1;
                        const bool take2 = u < ((double)S_n / (double)den);
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            S_prime[s] = take2 ? S_prime[s] : r.prime[s];
                        S_n += n1;
                        S_alpha = r.alpha + S_alpha;
                        S_nalpha += na1;
                        S_s = S_s && crit;
                    }
                };
                /* first child at level k: wait for the sibling if still valid; with s' = 0 the parent returns it as it
                 * is, so it keeps walking */
                auto push = [&](int k) {
                    if (walking && S_s) {
                        const double cnt =
                            __longlong_as_double((long long)((unsigned long long)S_n | ((unsigned long long)S_nalpha << 32)));
                        if (k < Cfg::LE) {
                            mm_lds_double *e = lds + (size_t)(Cfg::lds_E + k * ES) * 64;
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                e[s * 64] = S_prime[s];
                            e[NS * 64] = S_alpha;
                            e[(NS + 1) * 64] = cnt;
                        } else {
                            double *e = scr + (size_t)(Cfg::hbm_E + (k - Cfg::LE) * ES) * 64;
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                e[s * 64] = S_prime[s];
                            e[NS * 64] = S_alpha;
                            e[(NS + 1) * 64] = cnt;
                        }
                        walking = false;
                    }
                };
                if (j > 0) {
                    MM_LG_COUNT(7);
                    if (merge0)
                        merge(r0); /* level 0, records requested before the leapfrog */
                    else
                        push(0);
                    for (int k = 1; k < j; ++k) {
                        if (__ballot(walking) == 0ull)
                            break;
                        MM_LG_COUNT(7);
                        if ((leaf >> k) & 1u) {
                            rec rk;
                            load_rec(k, first_slot(leaf, k), rk);
                            merge(rk);
                        } else {
                            push(k);
                        }
                    }
                }
                done = done || walking; /* reached level j: the doubling is complete, or was cut short */
                MM_LG_TICK(3);
            }

            /* write the advanced edge back; whole-trajectory criterion against the other edge (d = x_cur - x_other) */
            double ca = 0.0, cb = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double ox = other[s * 64], op = other[(NS + s) * 64];
                if (alive) {
                    edge[s * 64] = cx[s];
                    edge[(NS + s) * 64] = cp[s];
                    edge[(2 * NS + s) * 64] = cg[s];
                }
                const double d = cx[s] - ox;
                ca = fma(d, op, ca);
                cb = fma(d, cp[s], cb);
            }
            ca = mm_lg_group_sum(ca);
            cb = mm_lg_group_sum(cb);
            const bool crit_all = neg ? (ca <= 0.0 && cb <= 0.0) : (ca >= 0.0 && cb >= 0.0);
            const double tmp = fmin(1.0, (double)S_n / (double)n);
            const double u_run_2 = aux_peek();
            if (alive) {
                alpha = S_alpha;
                n_alpha = S_nalpha;
                aux_k += 1;
                if (S_s && (u_run_2 < tmp)) {
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        x[s] = S_prime[s];
                }
                n += S_n;
                bool s_new = S_s && crit_all;
                depth = j + 1;
                if (j + 1 >= a.max_depth)
                    s_new = false; /* depth cap: not in the reference */
                alive = s_new;
            }
            j += 1;
            MM_LG_TICK(4);
        }

        /* dual averaging (nuts.rs:676-690) */
        if (active) {
            double eta = 1.0 / (double)(m + MM_NUTS_T0);
            ad.h_bar = (1.0 - eta) * ad.h_bar + eta * (a.target_accept_p - alpha / (double)n_alpha);
            if (m <= a.n_discard) {
                const double _m = (double)m;
                ad.epsilon = mm_exp(ad.mu - sqrt(_m) / MM_NUTS_GAMMA * ad.h_bar);
                eta = mm_exp(-MM_NUTS_KAPPA * mm_log(_m));
                ad.epsilon_bar = mm_exp((1.0 - eta) * mm_log(ad.epsilon_bar) + eta * mm_log(ad.epsilon));
            } else {
                ad.epsilon = ad.epsilon_bar;
            }
            if (a.depth_hist && q == 0)
                atomicAdd(&a.depth_hist[depth < MM_NUTS_JMAX ? depth : MM_NUTS_JMAX], 1u);
        }
        if (t >= a.n_pre)
            record();
        MM_LG_TICK(5);
    }
#ifdef MM_LG_PROFILE
    if (lane == 0)
        for (int i = 0; i < 8; ++i)
            a.prof[(size_t)blockIdx.x * 8 + i] = mm_lg_prof_acc[i];
#endif

    if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            a.state[cl * D + 4 * s + q] = x[s];
        if (q == 0) {
            a.adapt[cl] = ad;
            if (a.n_leapfrog)
                a.n_leapfrog[cl] += n_lf;
        }
    }
}

template <int D> hipError_t mm_launch_nuts_lg(const mm_nuts_lg_args &a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 15) / 16);
    const size_t lds = mm_lg_cfg<D>::lds_bytes;
    hipLaunchKernelGGL((mm_nuts_lg_kernel<D>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

#endif /* MM_NUTS_LG_H */
