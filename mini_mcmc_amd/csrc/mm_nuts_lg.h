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
 * Tree building in lock-step.  The 16 chains of a wave run the same doubling j and the same leaf index at the same
 * time; a chain whose subtree stopped idles until the doubling ends.  Because the live chains share the leaf index,
 * their pending-subtree stacks have the same shape and the level index is wave-uniform.  A chain that fails early
 * walks up the remaining levels at once (merging where it is a second child, drawing the merge uniform, exactly as
 * the recursion returns through nuts.rs:858-929).
 *
 * Tree-depth compaction (BASELINE.json config 5: "divergent-tree wavefront compaction").  Trees of different chains
 * have different depths (at the config: 2^7 .. 2^9 leaves for most, a few per cent at 2^10), and a wave that keeps
 * its chains for the whole transition runs as long as its deepest tree: 38 % of the leapfrogs it executes are wanted.
 * So a transition is cut at the doubling boundaries: the `begin` kernel draws the momentum and runs doublings
 * 0 .. j0-1 with the chains in their natural waves; every chain that still wants to double is appended to a work
 * list, its state (sample, both edges, log-joint, slice, counters: mm_lg_rec) parked in HBM; then one `double` kernel
 * per level j >= j0 takes the list 16 chains at a time -- full waves of chains that all run 2^j leaves -- and appends
 * the survivors to the next list.  A chain whose transition ends is finished (dual averaging, output row) by the
 * kernel that ran its last doubling.  Which chains share a wave depends on the order of the atomic appends and
 * changes from run to run; what a chain computes does not (its stream is keyed by its global index, MFMA columns are
 * independent), so the samples are bit-identical to the single-launch kernel and to the host twin.
 *
 *
 * Cost model (tools/f64_rate.hip, tools/lg_profile.py, profiles/).  On gfx950 the f64 matrix rate equals the f64
 * vector rate (v_mfma_f64_16x16x4: 64 cycles; v_fma_f64 / v_add_f64: ~5 cycles per wave instruction at ANY occupancy)
 * and the two do not overlap -- 2 MFMA + 8 FMA take 128 + 43 cycles whether one or four waves share the SIMD.  A leaf
 * costs 16 MFMA = 1024 cycles plus ~4-5 cycles for every other instruction around it, and a second wave per SIMD
 * bought 1.25x at twice the register pressure: the kernels run one wave per SIMD with the 512-register budget and
 * 40 KB of LDS, and the work is to keep the instruction count per leaf down:
 *   - registers hold the edge being extended (x, p, g), the subtree proposal, the current sample and the A-operand
 *     blocks; everything else is addressed by wave-uniform indices and lives in memory;
 *   - pending first children, per level k: proposal, alpha sum, counts (10 slots, lane-interleaved: slot i of lane l
 *     at base[i * 64 + l]).  Levels 0..2 in LDS, deeper (touched every 2^(k+1) leaves) in a per-wave HBM scratch;
 *   - "first leaf" table: the (x, p) of the leaf that starts a subtree.  The first leaf of the pending sibling at
 *     level k of leaf i is leaf i0 = i with its k + 1 low bits cleared, and its data is filed under c = ctz(i0)
 *     (leaf 0 under c = JMAX): every even leaf is written once, under the highest level it starts, instead of being
 *     copied from stack entry to stack entry.  c = 1..3 in LDS, the rest in HBM;
 *   - both trajectory edges live in the chain's record (HBM): the one being extended is loaded at the start of a
 *     doubling and written back at its end;
 *   - the U-turn test needs no per-lane orientation: with d = x_cur - x_first, A = d.p_first, B = d.p_cur it is
 *     (A >= 0 and B >= 0) for v = +1 and (A <= 0 and B <= 0) for v = -1 -- negating every term of an fma chain negates
 *     the result exactly, so this equals the twin's (x_plus - x_minus).p_minus/plus >= 0 bit for bit;
 *   - the records of the level-0 merge of an odd leaf are requested before the leapfrog, so their latency hides
 *     behind the MFMAs.
 */
#ifndef MM_NUTS_LG_H
#define MM_NUTS_LG_H

#include <hip/hip_runtime.h>

#include <atomic>

#include "mm_nuts.h"

/* Section timers for tools/lg_profile.hip (s_memtime deltas of lane 0 per wave); compiled out of the product. */
#ifdef MM_LG_PROFILE
/* MM_LG_PROFILE_MASK: which sections are stamped (bit = section).  Every stamp costs ~100 cycles (s_memtime + the wait for
 * it), and the full set -- a dozen per leaf -- makes the profiled kernel 2.7 times slower than the product; a coarse mask
 * (0x3f: transition / doubling prologue and epilogue, end of a leaf, end of a walk) distorts little: the time of an unstamped
 * section flows into the next stamped one. */
#ifndef MM_LG_PROFILE_MASK
#define MM_LG_PROFILE_MASK 0xffffu
#endif
#define MM_LG_TICK(L, sec)                                                                                        \
    do {                                                                                                          \
        if ((MM_LG_PROFILE_MASK >> (sec)) & 1u) {                                                                 \
            const unsigned long long _now = __builtin_amdgcn_s_memtime();                                         \
            (L).prof_acc[sec] += _now - (L).prof_t;                                                               \
            (L).prof_t = _now;                                                                                    \
        }                                                                                                         \
    } while (0)
#define MM_LG_COUNT(L, sec) ((L).prof_acc[sec] += 1)
#else
#define MM_LG_TICK(L, sec) ((void)0)
#define MM_LG_COUNT(L, sec) ((void)0)
#endif

/* Work queues of the persistent kernel: queue 0 = chains ready to begin a transition, queue 1 + j = chains wanting
 * doubling j.  The queues exist MM_LGQ_SHARDS times; a wave appends to and takes from the shard of its workgroup
 * index (mod SHARDS: the XCD it runs on) and looks at the others only when its own has no full wave of work.  A
 * chain is in at most one queue, so c_pad entries per ring suffice. */
#define MM_LGQ_NQ (MM_NUTS_JMAX + 1)
#define MM_LGQ_SHARDS 8
#define MM_LGQ_ID_BITS 20 /* an entry = (lap tag << 20) | local chain index */
struct mm_lgq_ctrl {
    /* Per shard one 128-byte line, so that a wave reads the state of the shard's queues with ONE load (lane i reads
     * word i): word q < MM_LGQ_NQ = (entries appended << 32) | entries handed out, counted from the start of the run.
     * (Separate head / tail words read one by one by a thousand waves, all on one line, spent 80 % of the kernel in
     * the scheduler.) */
    unsigned long long w[MM_LGQ_SHARDS][16];
    unsigned long long remaining; /* chains with transitions left */
    unsigned long long error;     /* != 0 after a watchdog fired: every wave leaves */
    unsigned long long pad[14];
    unsigned long long stat_units, stat_chains, stat_polls; /* work units taken, chains in them, idle polls */
    unsigned long long stat_leaf_iters;                     /* leaf iterations (of 16 chain slots each) executed */
    unsigned long long stat_t[4];                           /* s_memtime ticks: pick, fetch, work, hand-over */
};

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
    double *rec;                    /* per-chain records (mm_lg_rec), c_pad chains */
    unsigned long long c_pad;       /* n_chains rounded up to a multiple of 16 */
    /* compaction kernels only */
    unsigned int *lists;            /* [MM_NUTS_JMAX][c_pad] local chain indices wanting doubling j */
    unsigned int *counts;           /* [2][MM_NUTS_JMAX + 1], by transition parity */
    int j0;                         /* doublings < j0 run in the begin kernel */
    int j;                          /* double kernel: the doubling to run */
    unsigned int m;                 /* 1-based transition index (self.m after the increment) */
    unsigned int row;               /* output row of this transition, or 0xffffffff */
    /* persistent kernel only */
    struct mm_lgq_ctrl *ctrl;       /* queue heads / tails, chains left, error flag */
    unsigned int patience;          /* idle polls before a wave settles for a unit of fewer than 16 chains */
    unsigned int min_unit;          /* a wave stays with the chains it kept when the queue of their level tops them up to this many */
    unsigned int *slots;            /* [MM_LGQ_SHARDS][MM_LGQ_NQ][c_pad] rings of tagged local chain indices */
#ifdef MM_LG_PROFILE
    unsigned long long *prof;       /* [waves][8] */
#endif
};

/* build switches of the experiments kept for comparison */
#ifndef MM_LG_ASM_MFMA
#define MM_LG_ASM_MFMA 1
#endif
#ifndef MM_LG_AUX_SHARED
#define MM_LG_AUX_SHARED 1
#endif
#ifndef MM_LG_PAIR_UNROLL
#define MM_LG_PAIR_UNROLL 0 /* two pairs per trip (leaf index mod 4 constant): 468 -> 465 ms for 30 % more code: off */
#endif
#ifndef MM_LG_LEAN
#define MM_LG_LEAN 1 /* round 5: the leaf loop without lane guards (mm_lg_doubling: "lean pair loop"); 0 = the guarded loop only */
#endif
#ifndef MM_LG_F_RECENT
#define MM_LG_F_RECENT 1 /* round 6: the last LDS first-leaf slot keeps the most recent leaf with c >= LF + 1 (load_rec); 0 = slot per c only */
#endif
#ifndef MM_LG_CHECK_FORM
#define MM_LG_CHECK_FORM 3 /* round 6 (profiles/r6i_, r6j_nuts_check_form*_probe.log; best case 2873 cycles per leaf iteration): 0 = the check and its rare case inside the pair loop (round 5); 1 = its branch marked unlikely (2879); 2 = the three conditions as 64-bit lane masks on the scalar unit (2833); 3 = 2 + the rare case handled OUTSIDE the fast loop, which is left and entered again (2789; without any check: 2725) */
#endif
#ifndef MM_LG_DEEP_EARLY
#define MM_LG_DEEP_EARLY 0 /* round 6 experiment: the level-(WU + 1) merge's HBM records requested ahead of the LDS-level merges */
#endif
#ifndef MM_LG_TOUCH
#define MM_LG_TOUCH 0 /* round 6 experiment: touch the HBM records of a pair's level-(LE + 1) merge two leaves ahead */
#endif
#ifndef MM_LG_PINGPONG
#define MM_LG_PINGPONG 1 /* round 6: the pair's two leaves write (x, p) alternately into the first-leaf copy and back (no register copies); 0 = in place + copy */
#endif
#ifndef MM_LG_WALK_UNROLL
#define MM_LG_WALK_UNROLL 3 /* config 5: 0 510 ms, 1 496, 2 473, 3 469, 4 470 */
#endif

/* OCC = waves per SIMD the kernel is built for: 1 = the 512-register budget and 40 KB of LDS per wave; 2 = 256 registers
 * and 20 KB (persistent scheduler only).  Level 0 of the pending-subtree stack never reaches memory: leaves are taken
 * in pairs and the first leaf's subtree waits for its sibling in registers (so does its (x, p) for the stop criterion).
 * Memory holds entry(k) for k >= 1 -- k <= LE in LDS, deeper levels in HBM -- and the first-leaf records c >= 2 (c = 1
 * would serve level-0 merges only) -- c <= 1 + LF in LDS */
template <int D, int OCC = 1> struct mm_lg_cfg {
    static_assert(D % 16 == 0, "lane-group kernel: D must be a multiple of 16");
    static constexpr int NS = D / 4;           /* coordinates per lane */
    static constexpr int NT = D / 16;          /* 16-row result tiles */
    static constexpr int ES = NS + 2;          /* pending entry: proposal[NS], alpha, (n | n_alpha << 32) */
    static constexpr int FS = 2 * NS;          /* first-leaf record: x[NS], p[NS] */
    static constexpr int LE = OCC == 1 ? 3 : 2; /* entry(k), 1 <= k <= LE, in LDS */
    static constexpr int LF = OCC == 1 ? 3 : 1; /* first(c), 2 <= c <= 1 + LF, in LDS */
    static constexpr int lds_E = 0, lds_F = LE * ES, lds_slots = LE * ES + LF * FS;
    static constexpr size_t lds_bytes = (size_t)lds_slots * 64 * sizeof(double);
    /* HBM slots per wave: entry(k), k = LE + 1 .. JMAX - 1 | first(c), c = LF + 2 .. JMAX */
    static constexpr int hbm_E = 0, hbm_F = (MM_NUTS_JMAX - 1 - LE) * ES, hbm_slots = hbm_F + (MM_NUTS_JMAX - 1 - LF) * FS;
    static constexpr size_t scratch_doubles_per_wave = (size_t)hbm_slots * 64;
    /* per-chain record, structure of arrays over the (padded) chains: vector v, coordinate 4 s + q of chain c at
     * rec[((v * NS + s) * c_pad + c) * 4 + q] -- a wave of 16 consecutive chains reads 512 contiguous bytes, a wave of
     * 16 arbitrary chains 16 x 32 bytes; then the scalars, field f of chain c at rec[vec_doubles + f * c_pad + c] */
    enum { V_XM = 0, V_PM = 1, V_GM = 2, V_XP = 3, V_PP = 4, V_GP = 5, V_X = 6, n_vec = 7 };
    enum { F_JOINT = 0, F_LOGU = 1, F_COUNTS = 2, F_M = 3, n_scalar = 4 }; /* counts = n | aux_k << 32; m: transitions done */
    static constexpr size_t rec_doubles(size_t c_pad) { return (size_t)(n_vec * D + n_scalar) * c_pad; }
};

typedef double mm_d4 __attribute__((ext_vector_type(4)));
/* fma(a, b, c) as the THREE-address v_fma_f64 with a destination of its own.  The compiler prefers the two-address
 * v_fmac_f64 (dst = dst + a b) and, where the old value of c must survive or lives elsewhere, puts a v_mov_b64 in front of
 * it: the lean pair loop of mm_lg_doubling carried 29 such copies per pair (round 6).  Same operation, same bits. */
__device__ __forceinline__ double mm_fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct mm_true_t { static constexpr bool value = true; };
struct mm_false_t { static constexpr bool value = false; };
/* LDS is addressed through an explicitly address-space-3 pointer: where an accessor picks LDS or HBM by a uniform
 * index, same-typed generic pointers let the optimiser merge the two branches into one flat_load / flat_store on a
 * selected pointer (which then waits on both memory counters); distinct pointer types keep ds_* and global_* apart */
typedef __attribute__((address_space(3))) double mm_lds_double;
typedef __attribute__((address_space(1))) double mm_glb_double;

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

/* A 64-bit constant the compiler must take from scalar registers where it is used: written inline, the leaf loop's
 * polynomial coefficients are hoisted into vector registers for the whole doubling (a v_fmac destroys its addend, so
 * each wants a VGPR copy), which is what pushed the loop over its register budget; from an SGPR pair v_fma_f64 reads
 * them as an operand. */
__device__ __forceinline__ double mm_lg_sconst(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}

/* min(1, exp(d)): the arithmetic of mm_exp (mm_math.h) operation for operation -- the host twin calls mm_exp -- but
 * without its three early returns: the range checks select at the end (the polynomial runs on whatever d is; an
 * out-of-range d only produces a value that the selects discard), which takes three levels of exec-mask branching
 * out of every leaf */
__device__ __forceinline__ double mm_lg_accept_prob(double d)
{
    const double kf = rint(d * mm_lg_sconst(1.44269504088896338700e+00));
    const int k = (int)kf;
    const double hi = fma(kf, mm_lg_sconst(-6.93147180369123816490e-01), d);
    const double lo = kf * mm_lg_sconst(1.90821492927058770002e-10);
    const double r = hi - lo;
    const double t = r * r;
    const double c =
        r - t * fma(t, fma(t, fma(t, fma(t, mm_lg_sconst(4.13813679705723846039e-08), mm_lg_sconst(-1.65339022054652515390e-06)),
                                  mm_lg_sconst(6.61375632143793436117e-05)),
                           mm_lg_sconst(-2.77777777770155933842e-03)),
                    mm_lg_sconst(1.66666666666666019037e-01));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k1 = k / 2, k2 = k - k1;
    const double s1 = mm_u2d((uint64_t)(k1 + 1023) << 52);
    const double s2 = mm_u2d((uint64_t)(k2 + 1023) << 52);
    double e = (y * s1) * s2;
    e = d < -745.1332191019411 ? 0.0 : e;
    e = d > 709.782712893384 ? (double)MM_INFINITY_F : e;
    e = d == d ? e : d;
    return fmin(1.0, e);
}

/* Two group sums at once: both land in all four lanes of the chain, each the same (c0 + c1) + (c2 + c3) as
 * mm_lg_group_sum (bit for bit), in 12 instructions instead of 2 x 10: the first swap pairs the rows of a with the
 * rows of b ([a0 b0 a2 b2] + [a1 b1 a3 b3]), the second folds the halves ([A B A B]), the third separates A from B. */
__device__ __forceinline__ void mm_lg_group_sum2(double a, double b, double *sum_a, double *sum_b)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    auto lo = [](double v) { return (unsigned int)__double2loint(v); };
    auto hi = [](double v) { return (unsigned int)__double2hiint(v); };
    u2 l = __builtin_amdgcn_permlane16_swap(lo(a), lo(b), false, false);
    u2 h = __builtin_amdgcn_permlane16_swap(hi(a), hi(b), false, false);
    double s = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]); /* [A01 B01 A23 B23] */
    l = __builtin_amdgcn_permlane32_swap(lo(s), lo(s), false, false);
    h = __builtin_amdgcn_permlane32_swap(hi(s), hi(s), false, false);
    s = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);        /* [A B A B] */
    l = __builtin_amdgcn_permlane16_swap(lo(s), lo(s), false, false);
    h = __builtin_amdgcn_permlane16_swap(hi(s), hi(s), false, false);
    *sum_a = __hiloint2double((int)h[0], (int)l[0]);
    *sum_b = __hiloint2double((int)h[1], (int)l[1]);
}

template <int NS> __device__ __forceinline__ double mm_lg_dot(const double *a, const double *b)
{
    double c = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        c = fma(a[s], b[s], c);
    return mm_lg_group_sum(c);
}

/* y = A x for the 16 chains of the wave; returns logp = -1/2 x.y.  The gradient is -y: the kernels carry y itself
 * (registers and edge records) and kick with -h -- fma(-h, y, p) and fma(h, -y, p) round the same exact product, so
 * this is the twin's fma(h, g, p) bit for bit, without eight negations per leaf */
template <int D, bool ALDS = false, class Lane>
__device__ __forceinline__ void mm_lg_ax(const Lane &L, const double *x, double *y)
{
    constexpr int NS = D / 4, NT = D / 16;
#if MM_LG_ASM_MFMA
    /* The A-operand blocks stay in accumulation registers and the MFMA reads them there ("a"), the products land in
     * ordinary registers ("v").  Through the builtin the compiler parks the blocks in AGPRs too, but copies each to a
     * VGPR before its MFMA (2 v_accvgpr_read + a wait state) and accumulates in AGPRs that it reads back (16 more):
     * 63 of a leaf's ~730 instructions.  Inline assembly is opaque to the hazard recogniser, so the wait states are
     * written out: 2 between a VALU write of x and the first MFMA, 18 after the last 16-pass MFMA before its result
     * is read (what the compiler itself emits around the builtin); accumulate chains need none. */
    if constexpr (!ALDS && D == 32) {
        mm_d4 acc0, acc1;
        asm volatile("s_nop 1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %2, %18, 0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %10, %18, 0\n"
                     "v_mfma_f64_16x16x4_f64 %0, %3, %19, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %11, %19, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %4, %20, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %12, %20, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %5, %21, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %13, %21, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %6, %22, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %14, %22, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %7, %23, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %15, %23, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %8, %24, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %16, %24, %1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %9, %25, %0\n"
                     "v_mfma_f64_16x16x4_f64 %1, %17, %25, %1\n"
                     "s_nop 15\n"
                     "s_nop 2\n"
                     : "=&v"(acc0), "=&v"(acc1)
                     : "a"(L.Aop[0][0]), "a"(L.Aop[0][1]), "a"(L.Aop[0][2]), "a"(L.Aop[0][3]), "a"(L.Aop[0][4]),
                       "a"(L.Aop[0][5]), "a"(L.Aop[0][6]), "a"(L.Aop[0][7]), "a"(L.Aop[1][0]), "a"(L.Aop[1][1]),
                       "a"(L.Aop[1][2]), "a"(L.Aop[1][3]), "a"(L.Aop[1][4]), "a"(L.Aop[1][5]), "a"(L.Aop[1][6]),
                       "a"(L.Aop[1][7]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]),
                       "v"(x[7]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[r] = acc0[r];
            y[4 + r] = acc1[r];
        }
        return;
    } else if constexpr (!ALDS && D == 16) {
        mm_d4 acc0;
        asm volatile("s_nop 1\n"
                     "v_mfma_f64_16x16x4_f64 %0, %1, %5, 0\n"
                     "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n"
                     "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n"
                     "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n"
                     "s_nop 15\n"
                     "s_nop 2\n"
                     : "=&v"(acc0)
                     : "a"(L.Aop[0][0]), "a"(L.Aop[0][1]), "a"(L.Aop[0][2]), "a"(L.Aop[0][3]), "v"(x[0]), "v"(x[1]),
                       "v"(x[2]), "v"(x[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r)
            y[r] = acc0[r];
        return;
    }
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            /* ALDS (two waves per SIMD: half the registers): the A-operand blocks are read from the workgroup's LDS
             * copy, one ds_read_b64 per MFMA (the read of block s + 1 is in flight during MFMA s) */
            const double a_ts = ALDS ? L.Alds[(size_t)(t * NS + s) * 64] : L.Aop[t][s];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ts, x[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            y[4 * t + r] = acc[r];
    }
}
template <int D, bool ALDS = false, class Lane>
__device__ __forceinline__ double mm_lg_logp_ax(const Lane &L, const double *x, double *y)
{
    mm_lg_ax<D, ALDS>(L, x, y);
    return -0.5 * mm_lg_dot<D / 4>(x, y);
}

/* Loads / stores of data that is handed from wave to wave INSIDE a kernel (persistent scheduler): agent-scope relaxed
 * atomics, i.e. global_load / global_store with sc1 -- they go past the per-CU cache and the per-XCD L2, so no
 * whole-cache write-back / invalidate (what an agent-scope fence costs; with 1024 waves doing that per work unit the
 * kernel ran 10x slower) is needed.  COH = false: plain accesses, for the kernels that hand over at launch boundaries */
template <bool COH> __device__ __forceinline__ double mm_lg_ld(const double *p)
{
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void mm_lg_st(double *p, double v)
{
    if (COH)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool COH> __device__ __forceinline__ mm_nuts_adapt<double> mm_lg_ld_adapt(const mm_nuts_adapt<double> *p)
{
    mm_nuts_adapt<double> ad;
    ad.epsilon = mm_lg_ld<COH>(&p->epsilon);
    ad.epsilon_bar = mm_lg_ld<COH>(&p->epsilon_bar);
    ad.h_bar = mm_lg_ld<COH>(&p->h_bar);
    ad.mu = mm_lg_ld<COH>(&p->mu);
    return ad;
}
template <bool COH> __device__ __forceinline__ void mm_lg_st_adapt(mm_nuts_adapt<double> *p, const mm_nuts_adapt<double> &ad)
{
    mm_lg_st<COH>(&p->epsilon, ad.epsilon);
    mm_lg_st<COH>(&p->epsilon_bar, ad.epsilon_bar);
    mm_lg_st<COH>(&p->h_bar, ad.h_bar);
    mm_lg_st<COH>(&p->mu, ad.mu);
}

/* What a lane carries for its chain across the doublings of a transition */
template <int D> struct mm_lg_lane {
    static constexpr int NS = D / 4, NT = D / 16;
    double Aop[NT][NS];   /* A-operand blocks: lane (i = l & 15, k = l >> 4) holds A[16 t + i][4 s + k] */
    const mm_lds_double *Alds; /* ... or (two-waves-per-SIMD build) their LDS copy: block (t, s) of lane l at [(t NS + s) 64 + l] */
    double x[NS];         /* current sample */
    double joint, logu;   /* log joint at the start of the transition, log slice level */
    unsigned int n;       /* points of the trajectory inside the slice */
    unsigned int aux_k, aux_have;
    mm_u32x4 aux_blk;     /* auxiliary uniforms: draw k of (chain, m) is a half of Philox block AUX + (k >> 1) */
    double alpha;         /* of the last doubling (Q11) */
    unsigned int n_alpha;
    int depth;
    unsigned long long n_lf;
    unsigned long long n_leaf_iters; /* leaf iterations of the wave this lane took part in (wave-uniform) */
    unsigned long long chain; /* global chain index (keys the stream) */
    unsigned long long cl;    /* local chain index (record, state, adaptation state) */
    unsigned int m;
    int lane, q;
    bool active;
#ifdef MM_LG_PROFILE
    unsigned long long prof_acc[16], prof_t;
#endif
};

template <int D> __device__ __forceinline__ void mm_lg_load_A(mm_lg_lane<D> &L, const double *mat)
{
    const int c = L.lane & 15;
#pragma unroll
    for (int t = 0; t < D / 16; ++t)
#pragma unroll
        for (int s = 0; s < D / 4; ++s)
            L.Aop[t][s] = mat[(size_t)(16 * t + c) * D + 4 * s + L.q];
}

template <int D> __device__ __forceinline__ double mm_lg_aux_peek(mm_lg_lane<D> &L, unsigned long long seed)
{
    /* Draw k is a half of block AUX + (k >> 1).  The four lanes of a chain hold four CONSECUTIVE blocks (lane q block
     * 4 g + q) instead of four copies of one, so the ten Philox rounds run once per eight draws, not once per two;
     * the draw is fetched from the lane that holds its block through the LDS crossbar (ds_bpermute: no memory).  The
     * lanes of a chain share aux_k, so they refill in the same call. */
#if !MM_LG_AUX_SHARED
    const unsigned int b1 = L.aux_k >> 1;
    if (b1 != L.aux_have) {
        L.aux_blk = mm_block(seed, L.chain, L.m, MM_AUX_BLOCK + b1);
        L.aux_have = b1;
    }
    return (L.aux_k & 1u) ? mm_u53(L.aux_blk.w[2], L.aux_blk.w[3]) : mm_u53(L.aux_blk.w[0], L.aux_blk.w[1]);
#endif
    const unsigned int b = L.aux_k >> 1, g = b >> 2;
    if (g != L.aux_have) {
        L.aux_blk = mm_block(seed, L.chain, L.m, MM_AUX_BLOCK + 4u * g + (unsigned int)L.q);
        L.aux_have = g;
    }
    const bool odd = (L.aux_k & 1u) != 0u;
    const int src = ((L.lane & 15) + 16 * (int)(b & 3u)) * 4;
    const unsigned int hi = (unsigned int)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? L.aux_blk.w[2] : L.aux_blk.w[0]));
    const unsigned int lo = (unsigned int)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? L.aux_blk.w[3] : L.aux_blk.w[1]));
    return mm_u53(hi, lo);
}

template <int D> __device__ __forceinline__ double *mm_lg_rec_vec(const mm_nuts_lg_args &a, const mm_lg_lane<D> &L, int v)
{
    return a.rec + ((size_t)v * (D / 4) * a.c_pad + L.cl) * 4 + L.q; /* coordinate s at [s * c_pad * 4] */
}
template <int D> __device__ __forceinline__ double *mm_lg_rec_scalar(const mm_nuts_lg_args &a, const mm_lg_lane<D> &L, int f)
{
    return a.rec + (size_t)mm_lg_cfg<D>::n_vec * D * a.c_pad + (size_t)f * a.c_pad + L.cl;
}

/* Both trajectory edges of the wave's chains in REGISTERS (the persistent scheduler, one wave per SIMD: round 5).  With the
 * edges in the chains' records every doubling starts with a load of the edge it extends and ends with a load of the other one
 * for the whole-trajectory criterion -- two dependent HBM round trips of ~2000 cycles each, which is two thirds of a leaf
 * iteration per doubling: the doublings 0 .. 3 of a transition (15 leaves) cost 44 000 cycles of leaves and 24 000 of waiting
 * (tools/nuts_lg_floor_probe.hip by level: 3113 cycles per leaf iteration at level 5, 2930 at level 9).  Here a unit loads both
 * edges once (a unit that begins a transition: none at all), every doubling of the unit finds them in registers -- the edge
 * not being extended is not touched by the leaf loop and waits in accumulation registers -- and the unit stores what it changed
 * when it parks the chains.  `cur` is the edge extended last, per lane (the direction is the chain's own draw): a doubling in
 * the other direction swaps the two under the lane mask. */
template <int D> struct mm_lg_edges {
    static constexpr int NS = D / 4;
    double cx[NS], cp[NS], cg[NS]; /* the edge extended last: position, momentum, A x */
    double ox[NS], op[NS], og[NS]; /* the other edge */
    bool cur_neg;                  /* cur is the minus edge (this lane's chain) */
};

/* start of a transition (nuts.rs:550-576): momentum, log joint, slice; both edges = (x, p0, grad), written to the chain's
 * record or (E != nullptr) left in registers */
template <int D, bool COH, int OCC, bool RES>
__device__ __forceinline__ void mm_lg_begin(mm_lg_lane<D> &L, const mm_nuts_lg_args &a, mm_lg_edges<D> &E)
{
    using Cfg = mm_lg_cfg<D, OCC>;
    constexpr int NS = Cfg::NS;
    L.aux_k = 0;
    L.aux_have = 0xffffffffu;
    L.aux_blk.w[0] = L.aux_blk.w[1] = L.aux_blk.w[2] = L.aux_blk.w[3] = 0u;
    double p0[NS], grad[NS];
    if constexpr (NS % 2 == 0) {
        /* Coordinate d = 4 s + q is a half of Philox block d >> 1 = 2 s + (q >> 1): the lanes q and q ^ 1 of a chain need the
         * SAME NS blocks, one the cosine and one the sine half of each Box-Muller pair.  Each evaluates half of the blocks in
         * full (even rows s < NS / 2, odd rows the rest) and the two trade the halves they do not use themselves -- rows q and
         * q ^ 1 are what v_permlane16_swap exchanges -- instead of both evaluating all NS (a Philox block, a logarithm, a square
         * root and a sine / cosine pair in f64 each: 6600 of the 29 500 cycles a unit that begins a transition spends outside its
         * leaves, tools/nuts_lg_floor_probe.hip).  Same functions of the same arguments: same bits. */
        constexpr int H = NS / 2;
        const bool odd = (L.q & 1) != 0;
        double mine[H], give[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const int s = odd ? H + i : i;
            mm_u32x4 blk = mm_block(a.seed, L.chain, L.m, (uint32_t)(2 * s + (L.q >> 1)));
            double z0, z1;
            mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
            mine[i] = odd ? z1 : z0;
            give[i] = odd ? z0 : z1;
        }
#pragma unroll
        for (int i = 0; i < H; ++i) {
            typedef unsigned int u2 __attribute__((ext_vector_type(2)));
            const unsigned int glo = (unsigned int)__double2loint(give[i]), ghi = (unsigned int)__double2hiint(give[i]);
            /* first result: even rows keep their own, odd rows get the even partner's; second: even rows get the odd partner's */
            const u2 l2 = __builtin_amdgcn_permlane16_swap(glo, glo, false, false);
            const u2 h2 = __builtin_amdgcn_permlane16_swap(ghi, ghi, false, false);
            const double got = odd ? __hiloint2double((int)h2[0], (int)l2[0]) : __hiloint2double((int)h2[1], (int)l2[1]);
            p0[i] = odd ? got : mine[i];
            p0[H + i] = odd ? mine[i] : got;
        }
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int d = 4 * s + L.q;
            mm_u32x4 blk = mm_block(a.seed, L.chain, L.m, (uint32_t)(d >> 1));
            double z0, z1;
            mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
            p0[s] = (d & 1) ? z1 : z0;
        }
    }
    const double ulogp = mm_lg_logp_ax<D, OCC == 2>(L, L.x, grad); /* grad holds A x = -gradient (see mm_lg_logp_ax) */
    L.joint = ulogp - mm_lg_dot<NS>(p0, p0) * 0.5;
    const double exp1_obs = -mm_log(mm_lg_aux_peek<D>(L, a.seed));
    L.aux_k += 1;
    L.logu = L.joint - exp1_obs;
    L.n = 1;
    L.alpha = 0.0;
    L.n_alpha = 0;
    L.depth = 0;
    if constexpr (RES) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            E.cx[s] = E.ox[s] = L.x[s];
            E.cp[s] = E.op[s] = p0[s];
            E.cg[s] = E.og[s] = grad[s];
        }
        E.cur_neg = false;
        return;
    }
    const size_t st = (size_t)a.c_pad * 4;
    double *xm = mm_lg_rec_vec<D>(a, L, Cfg::V_XM), *pm = mm_lg_rec_vec<D>(a, L, Cfg::V_PM),
           *gm = mm_lg_rec_vec<D>(a, L, Cfg::V_GM), *xp = mm_lg_rec_vec<D>(a, L, Cfg::V_XP),
           *pp = mm_lg_rec_vec<D>(a, L, Cfg::V_PP), *gp = mm_lg_rec_vec<D>(a, L, Cfg::V_GP);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        mm_lg_st<COH>(&xm[s * st], L.x[s]);
        mm_lg_st<COH>(&pm[s * st], p0[s]);
        mm_lg_st<COH>(&gm[s * st], grad[s]);
        mm_lg_st<COH>(&xp[s * st], L.x[s]);
        mm_lg_st<COH>(&pp[s * st], p0[s]);
        mm_lg_st<COH>(&gp[s * st], grad[s]);
    }
}

template <int D, bool COH = false, int OCC = 1> __device__ __forceinline__ void mm_lg_begin(mm_lg_lane<D> &L, const mm_nuts_lg_args &a)
{
    mm_lg_edges<D> none;
    mm_lg_begin<D, COH, OCC, false>(L, a, none);
}

/* doubling j of the wave's chains (one iteration of `while s`, nuts.rs:578-671); `alive` in: the chain takes part,
 * out: it wants another doubling */
#ifndef MM_LG_UNIFORM_J
#define MM_LG_UNIFORM_J 1 /* round 6: the doubling's level as an SGPR (below); 0 = as the caller hands it over */
#endif
template <int D, bool COH, int OCC, bool RES>
__device__ __forceinline__ void mm_lg_doubling(mm_lg_lane<D> &L, const mm_nuts_lg_args &a, int j_in, bool &alive,
                                               double epsilon, mm_lds_double *lds, double *scr, mm_lg_edges<D> &E)
{
    /* Round 6: the level of the doubling is the same in all 64 lanes BY CONSTRUCTION (a unit is taken from ONE queue), but the
     * persistent scheduler derives it from queue words its lanes loaded, so the compiler had to treat it -- and with it the
     * number of leaves, the pair loop's exit, `k >= j` of every walk level and the level of the push -- as a per-lane value:
     * the pair loop was compiled as a DIVERGENT loop (exit mask accumulated in SGPR pairs, s_and_saveexec around every walk
     * level, phi copies of the edge at the back edge, a full s_waitcnt at the join).  One v_readfirstlane makes all of it
     * scalar control flow. */
#if MM_LG_UNIFORM_J
    const int j = __builtin_amdgcn_readfirstlane(j_in);
#else
    const int j = j_in;
#endif
    using Cfg = mm_lg_cfg<D, OCC>;
    constexpr int NS = Cfg::NS, ES = Cfg::ES;
    const size_t st = (size_t)a.c_pad * 4;

    /* records of the merge at level k >= 1 (wave-uniform k, cc >= 2): the sibling's entry and its first leaf */
    struct rec {
        double fx[NS], fp[NS], prime[NS], alpha, cnt;
    };
    auto load_rec = [&](int k, int cc, rec &r) __attribute__((always_inline)) {
#ifdef MM_LG_EXPERIMENT_NO_HBM /* timing experiment only (wrong trees): every record in LDS */
        cc = cc > 1 + Cfg::LF + MM_LG_EXPERIMENT_NO_HBM ? 1 + Cfg::LF + MM_LG_EXPERIMENT_NO_HBM : cc;
        k = k > Cfg::LE + MM_LG_EXPERIMENT_NO_HBM ? Cfg::LE + MM_LG_EXPERIMENT_NO_HBM : k;
#endif
        /* Three cases, each ONE block that issues all 26 loads before anything waits: written as two independent
         * if / else (first-leaf record, then entry) the entry's loads were sunk below the stop criterion, i.e. behind
         * the first-leaf record's wait -- two memory latencies per merge where the records are in HBM.  An entry
         * in HBM (k > LE) implies its first-leaf record is too (cc >= k + 1 > 1 + LF, as LF <= LE). */
        static_assert(Cfg::LF <= Cfg::LE, "record placement: LF <= LE");
        auto take = [&](auto *f, auto *e) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                r.prime[s] = e[s * 64];
            r.alpha = e[NS * 64];
            r.cnt = e[(NS + 1) * 64];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                r.fx[s] = f[s * 64];
                r.fp[s] = f[(NS + s) * 64];
            }
        };
        /* Round 6 -- which first-leaf record a merge reads: the sibling's first leaf i0 (the leaf index with its k + 1 low
         * bits cleared) is filed under c = ctz(i0) >= k + 1.  For a merge at a level k <= LF either c <= LF (its own LDS
         * slot) or i0 is THE most recent multiple of 2^(LF + 1) -- so the last LDS slot keeps "the most recent leaf with
         * c >= LF + 1" whatever its c (a leaf with c >= LF + 2 is filed there AND under its c in HBM, for the merges at
         * levels > LF) and every merge at a level <= LF finds its first-leaf record in LDS.  Before, a quarter of all pairs
         * (1/4 of those with one trailing one in the pair index, 1/2 with two, all with three) waited for an HBM round
         * trip here, now 1/16 (the merges at levels > LE, whose entries are in HBM anyway). */
        if (k > Cfg::LE)
            take((const double *)(scr + (size_t)(Cfg::hbm_F + (cc - 2 - Cfg::LF) * Cfg::FS) * 64),
                 (const double *)(scr + (size_t)(Cfg::hbm_E + (k - 1 - Cfg::LE) * ES) * 64));
        else if (MM_LG_F_RECENT ? k > Cfg::LF : cc > 1 + Cfg::LF)
            take((const double *)(scr + (size_t)(Cfg::hbm_F + (cc - 2 - Cfg::LF) * Cfg::FS) * 64),
                 (const mm_lds_double *)(lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64));
        else
            take((const mm_lds_double *)(lds + (size_t)(Cfg::lds_F + ((cc > Cfg::LF + 1 ? Cfg::LF + 1 : cc) - 2) * Cfg::FS) * 64),
                 (const mm_lds_double *)(lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64));
    };
    /* where the first leaf of the sibling at level k of `leaf` is filed */
    auto first_slot = [](unsigned int leaf, int k) -> int {
        const unsigned int i0 = leaf & ~((2u << k) - 1u);
        return i0 ? (__ffs((int)i0) - 1) : MM_NUTS_JMAX;
    };

    const double u_run_1 = mm_lg_aux_peek<D>(L, a.seed);
    if (alive)
        L.aux_k += 1;
    const bool neg = !(u_run_1 < 0.5); /* v = -1 */
    /* the outer edge in direction v, advanced in place: after the doubling it IS the returned edge */
    double cx[NS], cp[NS], cg[NS];
    if constexpr (RES) {
        /* the edges are in registers (mm_lg_edges): a doubling in the other direction than the last swaps them */
        const bool sw = neg != E.cur_neg;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cx[s] = sw ? E.ox[s] : E.cx[s];
            cp[s] = sw ? E.op[s] : E.cp[s];
            cg[s] = sw ? E.og[s] : E.cg[s];
            E.ox[s] = sw ? E.cx[s] : E.ox[s];
            E.op[s] = sw ? E.cp[s] : E.op[s];
            E.og[s] = sw ? E.cg[s] : E.og[s];
        }
        E.cur_neg = neg;
    } else {
        const double *const ex0 = mm_lg_rec_vec<D>(a, L, neg ? Cfg::V_XM : Cfg::V_XP);
        const double *const ep0 = mm_lg_rec_vec<D>(a, L, neg ? Cfg::V_PM : Cfg::V_PP);
        const double *const eg0 = mm_lg_rec_vec<D>(a, L, neg ? Cfg::V_GM : Cfg::V_GP);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cx[s] = mm_lg_ld<COH>(&ex0[s * st]);
            cp[s] = mm_lg_ld<COH>(&ep0[s * st]);
            cg[s] = mm_lg_ld<COH>(&eg0[s * st]);
        }
    }
    const double epsv = neg ? -epsilon : epsilon;
    const double h = epsv * 0.5, nh = -h; /* cg holds A x = -gradient */
    const unsigned int n_leaves = 1u << j;
    /* nothing in flight when the leaf loop starts: otherwise every use of these loop-carried registers gets a
     * conservative vmcnt wait */
    __builtin_amdgcn_s_waitcnt(0);

    bool done = !alive;
    unsigned int lf = 0, leaf_iters = 0; /* added to the lane's 64-bit counters once, after the loop */
    unsigned int S_n = 0, S_nalpha = 0;
    bool S_s = true;
    double S_alpha = 0.0;
    double S_prime[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        S_prime[s] = 0.0;
    bool walking = false;

    /* one leaf: leapfrog of the outer edge (nuts.rs:979-996), in place, and the base case of build_tree
     * (nuts.rs:782-856); chains that are done keep their edge (the matrix product runs for all 64 lanes: MFMA has no
     * per-lane mask, their columns are recomputed) */
    /* defer: the leaf's acceptance statistic min(1, exp(d)) is not evaluated here, d is left in d_last (the pair loop
     * evaluates the two of a pair in ONE pass, each on half of the chain's lanes) */
    double d_last = 0.0;
    auto leaf_eval = [&](unsigned int leaf, auto defer) __attribute__((always_inline)) {
        MM_LG_COUNT(L, 6);
        leaf_iters += 1u;
        /* The VECTORS of a chain that is done (edge, proposal) are never read again: a chain is done before the last
         * leaf only when its doubling was cut short, which ends the transition (s' = 0: no proposal taken, edges
         * rebuilt by the next mm_lg_begin), and the columns of lanes that take no part are scratch.  So they are
         * updated for all 64 lanes -- a masked update costs a v_mov per register -- and only the scalars are guarded */
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp[s] = fma(nh, cg[s], cp[s]);
            cx[s] = fma(epsv, cp[s], cx[s]);
        }
        MM_LG_TICK(L, 8);
        mm_lg_ax<D, OCC == 2>(L, cx, cg); /* cg = A x */
        MM_LG_TICK(L, 9);
        double xy = 0.0, pp = 0.0; /* x . A x and p . p, reduced together */
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp[s] = fma(nh, cg[s], cp[s]);
            xy = fma(cx[s], cg[s], xy);
            pp = fma(cp[s], cp[s], pp);
        }
        mm_lg_group_sum2(xy, pp, &xy, &pp);
        const double lp = -0.5 * xy;
        const double jointp = lp - pp * 0.5;
        /* the proposal of a one-leaf subtree is the leaf itself: S_prime is not set here but by the level-0 merge,
         * which picks between the pair's two leaves (cx and the copy of the first one) */
        if (!done) {
            lf += 1u;
            S_n = (L.logu < jointp) ? 1u : 0u;
            S_s = (L.logu - 1000.0) < jointp;
            if constexpr (!decltype(defer)::value)
                S_alpha = mm_lg_accept_prob(jointp - L.joint);
            S_nalpha = 1;
            /* a leaf that starts a subtree of level >= 2 files its (x, p) under the highest level it starts (the first
             * leaf of a level-1 subtree waits in registers, below) */
            if (j > 1 && (leaf & 3u) == 0u) {
                int cc = leaf ? (__ffs((int)leaf) - 1) : MM_NUTS_JMAX;
#ifdef MM_LG_EXPERIMENT_NO_HBM
                cc = cc > 1 + Cfg::LF + MM_LG_EXPERIMENT_NO_HBM ? 1 + Cfg::LF + MM_LG_EXPERIMENT_NO_HBM : cc;
#endif
                if (MM_LG_F_RECENT || cc <= 1 + Cfg::LF) { /* always in LDS: under c, or in the last slot ("the most recent leaf with c >= LF + 1", load_rec) */
                    mm_lds_double *f = lds + (size_t)(Cfg::lds_F + ((cc > Cfg::LF + 1 ? Cfg::LF + 1 : cc) - 2) * Cfg::FS) * 64;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        f[s * 64] = cx[s];
                        f[(NS + s) * 64] = cp[s];
                    }
                }
                if (cc > 1 + Cfg::LF) { /* and under c in HBM for the merges at levels > LF */
                    double *f = scr + (size_t)(Cfg::hbm_F + (cc - 2 - Cfg::LF) * Cfg::FS) * 64;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        f[s * 64] = cx[s];
                        f[(NS + s) * 64] = cp[s];
                    }
                }
            }
        }
        d_last = jointp - L.joint;
        walking = !done;
        MM_LG_TICK(L, 2);
    };
    /* a sibling waits: S is the second child, merge (nuts.rs:900-928); (fx, fp) = the sibling's first leaf */
    auto merge = [&](const double *fx, const double *fp, const double *prime, auto level0, double alpha,
                     double cnt_d) __attribute__((always_inline)) {
        MM_LG_TICK(L, 3);
        MM_LG_COUNT(L, 12);
        const double u = mm_lg_aux_peek<D>(L, a.seed);
        /* stop criterion on (first leaf of the sibling, current leaf): d = x_cur - x_first */
        double ca = 0.0, cb = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double d = cx[s] - fx[s];
            ca = fma(d, fp[s], ca);
            cb = fma(d, cp[s], cb);
        }
        mm_lg_group_sum2(ca, cb, &ca, &cb);
        const bool crit = neg ? (ca <= 0.0 && cb <= 0.0) : (ca >= 0.0 && cb >= 0.0);
        bool take2 = false;
        if (walking) {
            L.aux_k += 1;
            const unsigned long long cnt = (unsigned long long)__double_as_longlong(cnt_d);
            const unsigned int n1 = (unsigned int)cnt, na1 = (unsigned int)(cnt >> 32);
            unsigned int den = n1 + S_n;
            if (den < 1)
                den = 1;
            take2 = u < ((double)S_n / (double)den);
            S_n += n1;
            S_alpha = alpha + S_alpha;
            S_nalpha += na1;
            S_s = S_s && crit;
        }
        /* the proposal of a lane that is not walking is dead (filed, or done): select for all lanes */
#pragma unroll
        for (int s = 0; s < NS; ++s)
        {
            /* the second child's proposal: at level 0 the leaf itself (leaf_eval does not copy it to S_prime) */
            double second;
            if constexpr (decltype(level0)::value)
                second = cx[s];
            else
                second = S_prime[s];
            const double first = prime[s];
            S_prime[s] = take2 ? second : first;
        }
        MM_LG_TICK(L, 10);
    };
    /* first child at level k >= 1: wait for the sibling if still valid; with s' = 0 the parent returns it as it is, so
     * it keeps walking */
    auto push = [&](int k) __attribute__((always_inline)) {
#ifdef MM_LG_EXPERIMENT_NO_HBM
        k = k > Cfg::LE + MM_LG_EXPERIMENT_NO_HBM ? Cfg::LE + MM_LG_EXPERIMENT_NO_HBM : k;
#endif
        MM_LG_TICK(L, 3);
        MM_LG_COUNT(L, 13);
        if (walking && S_s) {
            const double cnt =
                __longlong_as_double((long long)((unsigned long long)S_n | ((unsigned long long)S_nalpha << 32)));
            if (k <= Cfg::LE) {
                mm_lds_double *e = lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    e[s * 64] = S_prime[s];
                e[NS * 64] = S_alpha;
                e[(NS + 1) * 64] = cnt;
            } else {
                double *e = scr + (size_t)(Cfg::hbm_E + (k - 1 - Cfg::LE) * ES) * 64;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    e[s * 64] = S_prime[s];
                e[NS * 64] = S_alpha;
                e[(NS + 1) * 64] = cnt;
            }
            walking = false;
        }
        MM_LG_TICK(L, 11);
    };
    /* hand S up the implicit recursion from level 1: walking lanes have S at level k at the top of iteration k */
    auto walk_level = [&](unsigned int leaf, int k) __attribute__((always_inline)) {
        MM_LG_COUNT(L, 7);
        if ((leaf >> k) & 1u) {
            rec rk;
            load_rec(k, first_slot(leaf, k), rk);
            merge(rk.fx, rk.fp, rk.prime, mm_false_t(), rk.alpha, rk.cnt);
        } else {
            push(k);
        }
    };
    auto walk_up = [&](unsigned int leaf) __attribute__((always_inline)) {
        /* levels 1 .. MM_LG_WALK_UNROLL (7 of 8 merges at 2 levels) are straight-line code with the level a constant:
         * fixed LDS addresses, no loop-carried copies of the subtree's scalars */
        constexpr int WU = OCC == 2 ? 0 : MM_LG_WALK_UNROLL; /* the two-waves-per-SIMD build is slower with it (518 -> 582 ms) */
        bool more = true;
#pragma unroll
        for (int k = 1; k <= WU; ++k) {
            more = more && k < j && __ballot(walking) != 0ull;
            if (more)
                walk_level(leaf, k);
        }
        if (more) {
            for (int k = WU + 1; k < j; ++k) {
                if (__ballot(walking) == 0ull)
                    break;
                walk_level(leaf, k);
            }
        }
        done = done || walking; /* reached level j: the doubling is complete, or was cut short */
        MM_LG_TICK(L, 3);
    };

    /* the same for the FIRST leaf of a pair, which walks only when it is not valid (s' = 0; a valid one waits): s' stays
     * 0 all the way up, so no proposal is kept, no criterion evaluated and nothing filed -- only the counts of the
     * siblings it meets are added (and their draws consumed).  Keeps the proposal registers dead across the first
     * leaf and the sibling records out of the register file. */
    auto walk_up_invalid = [&](unsigned int leaf) __attribute__((always_inline)) {
        for (int k = 1; k < j; ++k) {
            if (__ballot(walking) == 0ull)
                break;
            MM_LG_COUNT(L, 7);
            if ((leaf >> k) & 1u) {
                double alpha, cnt_d;
                if (k <= Cfg::LE) {
                    const mm_lds_double *e = lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64;
                    alpha = e[NS * 64];
                    cnt_d = e[(NS + 1) * 64];
                } else {
                    const double *e = scr + (size_t)(Cfg::hbm_E + (k - 1 - Cfg::LE) * ES) * 64;
                    alpha = e[NS * 64];
                    cnt_d = e[(NS + 1) * 64];
                }
                MM_LG_COUNT(L, 12);
                if (walking) {
                    L.aux_k += 1;
                    const unsigned long long cnt = (unsigned long long)__double_as_longlong(cnt_d);
                    S_n += (unsigned int)cnt;
                    S_alpha = alpha + S_alpha;
                    S_nalpha += (unsigned int)(cnt >> 32);
                }
            }
        }
        done = done || walking;
        MM_LG_TICK(L, 3);
    };

    MM_LG_TICK(L, 1);
    /* ---- the lean pair loop (round 5) -------------------------------------------------------------------------------
     * The guarded loop below protects every scalar of a chain that is done (or takes no part) with lane masks and lets
     * every lane decide by itself whether it still walks; measured on the product's own doubling with all 16 chains live
     * and valid (tools/nuts_lg_floor_probe.hip) that costs 3300 cycles per leaf iteration where the mandatory arithmetic
     * takes 2340.  Here NOTHING is guarded.  The live chains of a wave share the leaf index, so the walk of a pair has ONE
     * shape -- merges at the levels of the pair index's trailing ones, then a push -- and runs as wave-uniform
     * straight-line code for all 64 lanes; a lane whose chain is dead (takes no part, or its doubling was cut short)
     * computes on garbage: its columns of the matrix product, its own slots of the records and its scalars are private,
     * and nothing of it is read again -- except the three sums a chain that dies DURING the doubling hands to the dual
     * averaging (alpha, n_alpha, n), which are frozen in shadow registers at the moment it dies.  A chain dies when a
     * subtree of its is not valid (s' = 0, nuts.rs:858-899): from there the recursion only returns, adding the counts and
     * acceptance sums of the first children it passes -- `retire` does exactly that, at once, under the mask of the lanes
     * concerned (rare: once per chain and transition at most).  The auxiliary uniforms are drawn in lock-step (a chain
     * alive at doubling j has consumed 2^j + j of them), so the draw index is one scalar.  Every live lane executes the
     * same operations on the same values as in the guarded loop: the bits do not move (tests: test_nuts_lane_group_*,
     * test_nuts_config5_full_size_*). */
    bool lean = false;
    unsigned int ka = 0;
    if constexpr (MM_LG_LEAN && OCC == 1) {
        const unsigned long long am = __ballot(alive);
        if (j >= 1 && am != 0ull) {
            ka = (unsigned int)__builtin_amdgcn_readlane((int)L.aux_k, (int)__ffsll((long long)am) - 1);
            lean = __ballot(alive && L.aux_k != ka) == 0ull;
        }
    }
    if (lean) {
        bool dead = !alive, died = false;
        unsigned int live01 = alive ? 1u : 0u;
        unsigned int F_n = 0, F_nalpha = 0;
        double F_alpha = 0.0;
        unsigned int have_s = 0xffffffffu; /* the first draw refills */
        const int lane15x4 = (L.lane & 15) * 4;
        auto draw = [&]() __attribute__((always_inline)) -> double {
#ifdef MM_LG_EXP_CHEAP_DRAW
            ka += 1u;
            return 0.25 + 1e-9 * (double)ka;
#endif
#if MM_LG_UNIFORM_J
            const unsigned int kau = (unsigned int)__builtin_amdgcn_readfirstlane((int)ka); /* lock-step draws: one index for the wave */
#else
            const unsigned int kau = ka;
#endif
            const unsigned int b = kau >> 1, g = b >> 2;
            if (g != have_s) {
                L.aux_blk = mm_block(a.seed, L.chain, L.m, MM_AUX_BLOCK + 4u * g + (unsigned int)L.q);
                have_s = g;
            }
            const bool odd = (kau & 1u) != 0u;
            const int src = lane15x4 + 64 * (int)(b & 3u);
            const unsigned int hi = (unsigned int)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? L.aux_blk.w[2] : L.aux_blk.w[0]));
            const unsigned int lo = (unsigned int)__builtin_amdgcn_ds_bpermute(src, (int)(odd ? L.aux_blk.w[3] : L.aux_blk.w[1]));
            ka += 1u;
            return mm_u53(hi, lo);
        };
        /* (xi, pi) -> (xo, po): the pair's first leaf reads the edge and writes the first-leaf copy, the second reads that
         * and writes the edge (MM_LG_PINGPONG); in place when xo == xi */
        auto leaf_io = [&](unsigned int leaf, const double *xi, const double *pi, double *xo, double *po) __attribute__((always_inline)) {
            MM_LG_COUNT(L, 6);
            leaf_iters += 1u;
            double ph[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#if MM_LG_PINGPONG
                ph[s] = mm_fma3(nh, cg[s], pi[s]);
                xo[s] = mm_fma3(epsv, ph[s], xi[s]);
#else
                ph[s] = fma(nh, cg[s], pi[s]);
                xo[s] = fma(epsv, ph[s], xi[s]);
#endif
            }
            mm_lg_ax<D, false>(L, xo, cg);
            double xy = 0.0, pp = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#if MM_LG_PINGPONG
                po[s] = mm_fma3(nh, cg[s], ph[s]);
#else
                po[s] = fma(nh, cg[s], ph[s]);
#endif
                xy = fma(xo[s], cg[s], xy);
                pp = fma(po[s], po[s], pp);
            }
            mm_lg_group_sum2(xy, pp, &xy, &pp);
            const double lp = -0.5 * xy;
            const double jointp = lp - pp * 0.5;
            lf += live01;
            S_n = (L.logu < jointp) ? 1u : 0u;
            S_s = (L.logu - 1000.0) < jointp;
            S_nalpha = 1;
#ifdef MM_LG_EXP_NO_FILING /* timing experiments of tools/experiments/nuts_lg_lean_strip.sh: wrong trees */
            if (false) {
#else
            if (j > 1 && (leaf & 3u) == 0u) {
#endif
                const int cc = leaf ? (__ffs((int)leaf) - 1) : MM_NUTS_JMAX;
                if (MM_LG_F_RECENT || cc <= 1 + Cfg::LF) { /* always in LDS: under c, or in the last slot ("the most recent leaf with c >= LF + 1", load_rec) */
                    mm_lds_double *f = lds + (size_t)(Cfg::lds_F + ((cc > Cfg::LF + 1 ? Cfg::LF + 1 : cc) - 2) * Cfg::FS) * 64;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        f[s * 64] = xo[s];
                        f[(NS + s) * 64] = po[s];
                    }
                }
                if (cc > 1 + Cfg::LF) { /* and under c in HBM for the merges at levels > LF */
                    double *f = scr + (size_t)(Cfg::hbm_F + (cc - 2 - Cfg::LF) * Cfg::FS) * 64;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        f[s * 64] = xo[s];
                        f[(NS + s) * 64] = po[s];
                    }
                }
            }
            d_last = jointp - L.joint;
        };
        auto merge_l = [&](const double *mfx, const double *mfp, const double *prime, auto level0, double alpha, unsigned int n1,
                           unsigned int na1) __attribute__((always_inline)) {
            const double u = draw();
            double ca = 0.0, cb = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double d = cx[s] - mfx[s];
                ca = fma(d, mfp[s], ca);
                cb = fma(d, cp[s], cb);
            }
            mm_lg_group_sum2(ca, cb, &ca, &cb);
            const bool crit = neg ? (ca <= 0.0 && cb <= 0.0) : (ca >= 0.0 && cb >= 0.0);
            unsigned int den = n1 + S_n;
            if (den < 1)
                den = 1;
            const bool take2 = u < ((double)S_n / (double)den);
            S_n += n1;
            S_alpha = alpha + S_alpha;
            S_nalpha += na1;
            S_s = S_s && crit;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                double second;
                if constexpr (decltype(level0)::value)
                    second = cx[s];
                else
                    second = S_prime[s];
                S_prime[s] = take2 ? second : prime[s];
            }
        };
        /* the lanes of `who` return up the recursion with s' = 0 from level k_from: the sums of the first children they pass
         * (what the guarded loop's merges add for a lane that keeps walking); then they are dead and their sums frozen */
        auto retire = [&](unsigned int leaf, int k_from, bool who) __attribute__((always_inline)) {
            for (int k = k_from; k < j; ++k) {
                if ((leaf >> k) & 1u) {
                    double alpha, cnt_d;
                    if (k <= Cfg::LE) {
                        const mm_lds_double *e = lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64;
                        alpha = e[NS * 64];
                        cnt_d = e[(NS + 1) * 64];
                    } else {
                        const double *e = scr + (size_t)(Cfg::hbm_E + (k - 1 - Cfg::LE) * ES) * 64;
                        alpha = e[NS * 64];
                        cnt_d = e[(NS + 1) * 64];
                    }
                    if (who) {
                        const unsigned long long cnt = (unsigned long long)__double_as_longlong(cnt_d);
                        S_n += (unsigned int)cnt;
                        S_alpha = alpha + S_alpha;
                        S_nalpha += (unsigned int)(cnt >> 32);
                    }
                }
            }
            if (who) {
                F_n = S_n;
                F_alpha = S_alpha;
                F_nalpha = S_nalpha;
                died = true;
                dead = true;
                live01 = 0u;
            }
        };
        constexpr int WU = MM_LG_WALK_UNROLL;
        /* what the pair hands to the validity check (and, MM_LG_CHECK_FORM 3, out of the fast loop) */
        unsigned int P_n = 0, P_nalpha = 0, lf1 = 0;
        double P_alpha = 0.0;
        bool first_ok = true;
        int k_stop = 0;
#if MM_LG_CHECK_FORM == 3
        /* Round 6: the validity check LEAVES the pair loop instead of handling its case inside it.  The fast loop then changes
         * none of dead / live01 / died / F_* (they are invariant in it), carries no join behind a rare block, and its only
         * vector-dependent branch is one scalar test of three lane masks; the slow path (a chain retires: at most once per chain
         * and transition) runs between two entries of the fast loop. */
        unsigned int leaf = 0;
        for (;;) {
            if (__ballot(!dead) == 0ull)
                break;
            unsigned long long m_inv1 = 0ull, m_inv2 = 0ull;
            bool slow = false;
            for (; leaf < n_leaves; leaf += 2) {
#else
        for (unsigned int leaf = 0; leaf < n_leaves; leaf += 2) {
            if (__ballot(!dead) == 0ull)
                break;
#endif
#if MM_LG_TOUCH
            /* Round 6: a pair whose index ends in >= LE ones will merge at level LE + 1 with records that live in HBM (1/16 of
             * the pairs at LE = 3; ~2000 cycles of exposed latency each: one wave per SIMD has nobody to hide it behind).  Their
             * cache lines are TOUCHED here, two leaves ahead -- one 4-byte load per 128-byte line into a register nobody reads,
             * two loads per lane -- so that the merge's own loads find them in the L1 / L2. */
            if ((((leaf >> 1) + 1u) & ((1u << (Cfg::LE + 1)) - 1u)) == 0u && (int)Cfg::LE + 1 < j) {
                const unsigned int lfp = leaf | 1u;
                const unsigned int i0 = lfp & ~((2u << (Cfg::LE + 1)) - 1u);
                const int cc = i0 ? (__ffs((int)i0) - 1) : MM_NUTS_JMAX;
                const double *const scr0 = scr - L.lane; /* `scr` is the lane's own column of the wave's lane-interleaved slots */
                const float *e = reinterpret_cast<const float *>(scr0 + (size_t)(Cfg::hbm_E + 0 * ES) * 64);
                const float *f = reinterpret_cast<const float *>(scr0 + (size_t)(Cfg::hbm_F + (cc - 2 - Cfg::LF) * Cfg::FS) * 64);
                float t0 = 0.f, t1 = 0.f;
                if (L.lane * 32 < ES * 128)
                    t0 = __builtin_nontemporal_load(e + L.lane * 32);
                t1 = __builtin_nontemporal_load(f + L.lane * 32);
                asm volatile("" ::"v"(t0), "v"(t1));
            }
#endif
            /* ---- the first leaf of the pair: its one-leaf subtree waits for the sibling in registers */
            double pfx[NS], pfp[NS];
#if MM_LG_PINGPONG
            leaf_io(leaf, cx, cp, pfx, pfp);
#else
            leaf_io(leaf, cx, cp, cx, cp);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                pfx[s] = cx[s];
                pfp[s] = cp[s];
            }
#endif
            const double d_first = d_last;
            P_n = S_n;
            P_nalpha = S_nalpha;
            /* a first leaf that is not valid is handed up as it is and its sibling never built (nuts.rs:858-899): here the lane
             * runs on regardless and is set right at the end of the pair (ONE check per pair: a branch on a freshly computed
             * lane mask drains the wave's pipeline) */
            first_ok = S_s;
            /* ---- its sibling, both acceptance statistics in one pass (even rows the first leaf's d, odd rows the second's),
             *      the merge at level 0 */
#if MM_LG_PINGPONG
            leaf_io(leaf | 1u, pfx, pfp, cx, cp);
#else
            leaf_io(leaf | 1u, cx, cp, cx, cp);
#endif
            {
                const double e = mm_lg_accept_prob((L.q & 1) ? d_last : d_first);
                typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                const unsigned int elo = (unsigned int)__double2loint(e), ehi = (unsigned int)__double2hiint(e);
                const u2 l2 = __builtin_amdgcn_permlane16_swap(elo, elo, false, false);
                const u2 h2 = __builtin_amdgcn_permlane16_swap(ehi, ehi, false, false);
                P_alpha = __hiloint2double((int)h2[0], (int)l2[0]);
                S_alpha = __hiloint2double((int)h2[1], (int)l2[1]);
            }
            merge_l(pfx, pfp, pfx, mm_true_t(), P_alpha, P_n, P_nalpha);
            /* ---- the pair is handed up: merges at the levels of the trailing ones of the pair index, then a push */
            lf1 = leaf | 1u;
            k_stop = 0; /* the level of the push; j: the doubling is complete */
            auto level = [&](int k) __attribute__((always_inline)) {
                if (k >= j) {
                    k_stop = j;
                } else if ((lf1 >> k) & 1u) {
                    rec rk;
                    load_rec(k, first_slot(lf1, k), rk);
                    const unsigned long long cnt = (unsigned long long)__double_as_longlong(rk.cnt);
                    merge_l(rk.fx, rk.fp, rk.prime, mm_false_t(), rk.alpha, (unsigned int)cnt, (unsigned int)(cnt >> 32));
                } else {
                    const double cnt =
                        __longlong_as_double((long long)((unsigned long long)S_n | ((unsigned long long)S_nalpha << 32)));
                    if (k <= Cfg::LE) {
                        mm_lds_double *e = lds + (size_t)(Cfg::lds_E + (k - 1) * ES) * 64;
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            e[s * 64] = S_prime[s];
                        e[NS * 64] = S_alpha;
                        e[(NS + 1) * 64] = cnt;
                    } else {
                        double *e = scr + (size_t)(Cfg::hbm_E + (k - 1 - Cfg::LE) * ES) * 64;
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            e[s * 64] = S_prime[s];
                        e[NS * 64] = S_alpha;
                        e[(NS + 1) * 64] = cnt;
                    }
                    k_stop = k;
                }
            };
#ifdef MM_LG_EXP_FIXED_WALK
            {
                rec rk;
                load_rec(1, 2, rk);
                const unsigned long long cnt = (unsigned long long)__double_as_longlong(rk.cnt);
                merge_l(rk.fx, rk.fp, rk.prime, mm_false_t(), rk.alpha, (unsigned int)cnt & 1u, (unsigned int)(cnt >> 32) & 1u);
                mm_lds_double *e = lds + (size_t)(Cfg::lds_E) * 64;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    e[s * 64] = S_prime[s];
                e[NS * 64] = S_alpha;
                e[(NS + 1) * 64] = __longlong_as_double((long long)((unsigned long long)S_n | ((unsigned long long)S_nalpha << 32)));
                k_stop = (lf1 + 1u >= n_leaves) ? j : 1;
            }
#else
#if MM_LG_DEEP_EARLY
            /* Round 6 experiment: a pair whose index ends in WU + 1 ones merges at level WU + 1 with records from HBM.  Their
             * loads depend on nothing the LDS-level merges compute, so they are issued BEFORE those (~1800 cycles of merges at
             * levels 1 .. WU under the latency) instead of after them. */
            constexpr int KD = WU + 1;
            const bool deep = KD < j && (lf1 & ((2u << KD) - 1u)) == ((2u << KD) - 1u);
            rec rdeep;
            if (deep)
                load_rec(KD, first_slot(lf1, KD), rdeep);
#pragma unroll
            for (int k = 1; k <= WU; ++k)
                if (k_stop == 0)
                    level(k);
            if (k_stop == 0) {
                if (deep) {
                    const unsigned long long cnt = (unsigned long long)__double_as_longlong(rdeep.cnt);
                    merge_l(rdeep.fx, rdeep.fp, rdeep.prime, mm_false_t(), rdeep.alpha, (unsigned int)cnt, (unsigned int)(cnt >> 32));
                } else {
                    level(KD);
                }
            }
            for (int k = KD + 1; k_stop == 0; ++k)
                level(k);
#else
#pragma unroll
            for (int k = 1; k <= WU; ++k)
                if (k_stop == 0)
                    level(k);
            for (int k = WU + 1; k_stop == 0; ++k)
                level(k);
#endif
#endif
#if MM_LG_CHECK_FORM == 3
                {
                    const unsigned long long m_dead = __ballot(dead), m_first = __ballot(first_ok), m_ok = __ballot(S_s);
                    m_inv1 = ~m_dead & ~m_first;
                    m_inv2 = ~m_dead & m_first & ~m_ok;
                    if (__builtin_expect((m_inv1 | m_inv2) != 0ull, 0)) {
                        slow = true;
                        break;
                    }
                }
                if (k_stop >= j)
                    break; /* reached the doubling's own level: complete */
            }
            if (!slow)
                break; /* complete, or cut short by the scheduler's leaf count */
            {
                const bool inv1 = ((m_inv1 >> L.lane) & 1ull) != 0ull, inv2 = ((m_inv2 >> L.lane) & 1ull) != 0ull;
                if (inv1) {
                    S_n = P_n;
                    S_nalpha = P_nalpha;
                    S_alpha = P_alpha; /* min(1, exp(d)) of the first leaf */
                    lf -= 1u;          /* the sibling's leapfrog step was not the chain's */
                }
                retire(leaf, 1, inv1);
                retire(lf1, k_stop + 1, inv2);
            }
            if (k_stop >= j)
                break;
            leaf += 2;
            if (leaf >= n_leaves)
                break;
        }
#else
#ifndef MM_LG_EXP_NO_RETIRE
#if MM_LG_CHECK_FORM == 2
            {
                const unsigned long long m_dead = __ballot(dead), m_first = __ballot(first_ok), m_ok = __ballot(S_s);
                const unsigned long long m_inv1 = ~m_dead & ~m_first, m_inv2 = ~m_dead & m_first & ~m_ok;
                if (__builtin_expect((m_inv1 | m_inv2) != 0ull, 0)) {
                    const bool inv1 = ((m_inv1 >> L.lane) & 1ull) != 0ull, inv2 = ((m_inv2 >> L.lane) & 1ull) != 0ull;
#else
            {
                const bool inv1 = !dead && !first_ok;       /* the first leaf was not valid: what the sibling added is undone */
                const bool inv2 = !dead && first_ok && !S_s; /* the pair's subtree turned: it does not wait, it returns */
#if MM_LG_CHECK_FORM == 1
                if (__builtin_expect(__ballot(inv1 || inv2) != 0ull, 0)) {
#else
                if (__ballot(inv1 || inv2) != 0ull) {
#endif
#endif
                    if (inv1) {
                        S_n = P_n;
                        S_nalpha = P_nalpha;
                        S_alpha = P_alpha; /* min(1, exp(d)) of the first leaf */
                        lf -= 1u;          /* the sibling's leapfrog step was not the chain's */
                    }
                    retire(leaf, 1, inv1);
                    retire(lf1, k_stop + 1, inv2);
                }
            }
#endif
            if (k_stop >= j)
                break; /* reached the doubling's own level: complete */
        }
#endif
        MM_LG_TICK(L, 2); /* tools/lg_profile.py: the whole lean loop lands in section 2 */
        if (died) {
            S_n = F_n;
            S_alpha = F_alpha;
            S_nalpha = F_nalpha;
            S_s = false;
        }
        L.aux_k = ka;
        L.aux_have = have_s;
    } else
    if (j == 0) {
        if (__ballot(!done) != 0ull) {
            leaf_eval(0u, mm_false_t());
#pragma unroll
            for (int s = 0; s < NS; ++s)
                S_prime[s] = cx[s];
            done = done || walking;
        }
    } else {
        /* one pair of leaves; `leaf` = its first leaf.  Returns false when every chain of the wave is done */
        auto leaf_pair = [&](unsigned int leaf) __attribute__((always_inline)) -> bool {
            if (__ballot(!done) == 0ull)
                return false;
            /* ---- the first leaf of a level-1 subtree: its one-leaf subtree (level 0) waits for the sibling in registers */
            leaf_eval(leaf, mm_true_t());
            const double d_first = d_last;
            const bool live_first = !done;
            /* its proposal is its own x: (fx, fp) serve as the first leaf of the pair AND as the waiting proposal */
            double fx[NS], fp[NS], P_alpha = 0.0;
            unsigned int P_n, P_nalpha;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fx[s] = cx[s];
                fp[s] = cp[s];
            }
            P_n = S_n;
            P_nalpha = S_nalpha;
            MM_LG_COUNT(L, 7);
            MM_LG_COUNT(L, 13);
            walking = walking && !S_s; /* valid: it waits (nuts.rs:858-899); not valid: handed up as it is */
            /* The two acceptance statistics of a pair are evaluated together after the second leaf: the four lanes of a
             * chain all compute the same scalar, so even rows take the first leaf's d and odd rows the second's, one
             * pass of mm_lg_accept_prob (~45 instructions) instead of two, and a row swap hands both to every lane.
             * Only a first leaf that walks up now (not valid) needs its own at once: then both are evaluated in full */
            bool first_ready = false;
            if (__ballot(walking) != 0ull) {
                P_alpha = mm_lg_accept_prob(d_first);
                first_ready = true;
                if (live_first)
                    S_alpha = P_alpha;
            }
            walk_up_invalid(leaf);
            if (__ballot(!done) == 0ull)
                return false;
            /* ---- its sibling: merge at level 0 with the waiting subtree, then hand the pair up */
            leaf_eval(leaf | 1u, mm_true_t());
            {
                double a_second;
                if (first_ready) {
                    a_second = mm_lg_accept_prob(d_last);
                } else {
                    const double e = mm_lg_accept_prob((L.q & 1) ? d_last : d_first);
                    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                    const unsigned int elo = (unsigned int)__double2loint(e), ehi = (unsigned int)__double2hiint(e);
                    const u2 l = __builtin_amdgcn_permlane16_swap(elo, elo, false, false);
                    const u2 h = __builtin_amdgcn_permlane16_swap(ehi, ehi, false, false);
                    P_alpha = __hiloint2double((int)h[0], (int)l[0]);  /* rows 0, 2: the first leaf's */
                    a_second = __hiloint2double((int)h[1], (int)l[1]); /* rows 1, 3: the second leaf's */
                }
                if (!done)
                    S_alpha = a_second;
            }
            MM_LG_COUNT(L, 7);
            merge(fx, fp, fx, mm_true_t(), P_alpha,
                  __longlong_as_double((long long)((unsigned long long)P_n | ((unsigned long long)P_nalpha << 32))));
            walk_up(leaf | 1u);
            return true;
        };
#if MM_LG_PAIR_UNROLL
        /* two pairs per trip: the leaf index modulo 4 is a constant in each copy (which leaf files a first-leaf record,
         * whether level 1 merges or waits) */
        if (j == 1) {
            (void)leaf_pair(0u);
        } else {
            for (unsigned int leaf4 = 0; leaf4 < n_leaves; leaf4 += 4) {
                const unsigned int base = leaf4 & ~3u;
                if (!leaf_pair(base))
                    break;
                if (!leaf_pair(base | 2u))
                    break;
            }
        }
#else
        for (unsigned int leaf = 0; leaf < n_leaves; leaf += 2)
            if (!leaf_pair(leaf))
                break;
#endif
    }

    L.n_lf += lf;
    L.n_leaf_iters += leaf_iters;
    /* write the advanced edge back; whole-trajectory criterion against the other edge (d = x_cur - x_other).  The
     * record addresses are formed again from an opaque copy of the chain index: kept from the top of the doubling they
     * would hold ten registers across the leaf loop */
    double ca = 0.0, cb = 0.0;
    if constexpr (RES) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double d = cx[s] - E.ox[s];
            ca = fma(d, E.op[s], ca);
            cb = fma(d, cp[s], cb);
            E.cx[s] = cx[s];
            E.cp[s] = cp[s];
            E.cg[s] = cg[s];
        }
    } else {
        unsigned long long cl_end = L.cl;
        asm volatile("" : "+v"(cl_end));
        auto vec_end = [&](int v) __attribute__((always_inline)) {
            return a.rec + ((size_t)v * NS * a.c_pad + cl_end) * 4 + L.q;
        };
        double *const ex = vec_end(neg ? Cfg::V_XM : Cfg::V_XP);
        double *const ep = vec_end(neg ? Cfg::V_PM : Cfg::V_PP);
        double *const eg = vec_end(neg ? Cfg::V_GM : Cfg::V_GP);
        const double *const ox = vec_end(neg ? Cfg::V_XP : Cfg::V_XM);
        const double *const op = vec_end(neg ? Cfg::V_PP : Cfg::V_PM);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double oxs = mm_lg_ld<COH>(&ox[s * st]), ops = mm_lg_ld<COH>(&op[s * st]);
            if (alive) {
                mm_lg_st<COH>(&ex[s * st], cx[s]);
                mm_lg_st<COH>(&ep[s * st], cp[s]);
                mm_lg_st<COH>(&eg[s * st], cg[s]);
            }
            const double d = cx[s] - oxs;
            ca = fma(d, ops, ca);
            cb = fma(d, cp[s], cb);
        }
    }
    mm_lg_group_sum2(ca, cb, &ca, &cb);
    const bool crit_all = neg ? (ca <= 0.0 && cb <= 0.0) : (ca >= 0.0 && cb >= 0.0);
    const double tmp = fmin(1.0, (double)S_n / (double)L.n);
    const double u_run_2 = mm_lg_aux_peek<D>(L, a.seed);
    if (alive) {
        L.alpha = S_alpha;
        L.n_alpha = S_nalpha;
        L.aux_k += 1;
        if (S_s && (u_run_2 < tmp)) {
            if constexpr (OCC == 2) {
                /* the current sample lives in the chain's record (V_X) between the pieces of a transition: nothing of
                 * it is carried in registers across the leaf loop */
                double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    mm_lg_st<COH>(&xs[s * st], S_prime[s]);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    L.x[s] = S_prime[s];
            }
        }
        L.n += S_n;
        bool s_new = S_s && crit_all;
        L.depth = j + 1;
        if (j + 1 >= a.max_depth)
            s_new = false; /* depth cap: not in the reference */
        alive = s_new;
    }
    MM_LG_TICK(L, 4);
}
template <int D, bool COH = false, int OCC = 1>
__device__ __forceinline__ void mm_lg_doubling(mm_lg_lane<D> &L, const mm_nuts_lg_args &a, int j, bool &alive,
                                               double epsilon, mm_lds_double *lds, double *scr)
{
    mm_lg_edges<D> none;
    mm_lg_doubling<D, COH, OCC, false>(L, a, j, alive, epsilon, lds, scr, none);
}

/* Two waves per SIMD (OCC = 2): the doubling as an OUT-OF-LINE call with its state passed and returned by value.
 * Inlined into the scheduler kernel under a 256-register cap the allocator spilled across the leaf loop (46 scratch
 * reloads per leaf pair); as a function of its own the loop is allocated by itself (255 registers, no spills) and the
 * caller's live values are saved once per doubling.  By value, because a reference to the lane state would be a
 * generic pointer into scratch (every access a flat_load / flat_store). */
struct mm_lg_call_io {
    double joint, logu, alpha;
    unsigned long long chain, cl, n_lf, n_leaf_iters;
    unsigned int n, aux_k, aux_have, n_alpha, m;
    int depth, lane, q;
    mm_u32x4 aux_blk;
    int alive;
};
struct mm_lg_call_env {
    mm_glb_double *rec;
    unsigned long long c_pad, seed;
    int max_depth;
};
template <int D>
__device__ __attribute__((noinline)) mm_lg_call_io mm_lg_doubling_ool(mm_lg_call_io io, mm_lg_call_env env, int j, double epsilon,
                                                                    mm_lds_double *lds, mm_glb_double *scr, const mm_lds_double *Alds)
{
    mm_lg_lane<D> L;
    L.Alds = Alds;
    L.joint = io.joint;
    L.logu = io.logu;
    L.alpha = io.alpha;
    L.chain = io.chain;
    L.cl = io.cl;
    L.n_lf = io.n_lf;
    L.n_leaf_iters = io.n_leaf_iters;
    L.n = io.n;
    L.aux_k = io.aux_k;
    L.aux_have = io.aux_have;
    L.n_alpha = io.n_alpha;
    L.m = io.m;
    L.depth = io.depth;
    L.lane = io.lane;
    L.q = io.q;
    L.aux_blk = io.aux_blk;
    L.active = true;
    mm_nuts_lg_args a;
    /* pointers that cross a call lose their address space: name it again, or every record access becomes a flat_ one */
    a.rec = (double *)env.rec;
    a.c_pad = env.c_pad;
    a.seed = env.seed;
    a.max_depth = env.max_depth;
    bool alive = io.alive != 0;
    mm_lg_doubling<D, true, 2>(L, a, j, alive, epsilon, lds, (double *)scr);
    io.alpha = L.alpha;
    io.n_lf = L.n_lf;
    io.n_leaf_iters = L.n_leaf_iters;
    io.n = L.n;
    io.aux_k = L.aux_k;
    io.aux_have = L.aux_have;
    io.n_alpha = L.n_alpha;
    io.depth = L.depth;
    io.aux_blk = L.aux_blk;
    io.alive = alive ? 1 : 0;
    return io;
}

/* end of a transition: dual averaging (nuts.rs:676-690), depth histogram */
template <int D>
__device__ __forceinline__ void mm_lg_finish(const mm_lg_lane<D> &L, const mm_nuts_lg_args &a, mm_nuts_adapt<double> &ad,
                                             unsigned int *hist)
{
    double eta = 1.0 / (double)(L.m + MM_NUTS_T0);
    ad.h_bar = (1.0 - eta) * ad.h_bar + eta * (a.target_accept_p - L.alpha / (double)L.n_alpha);
    if (L.m <= a.n_discard) {
        const double _m = (double)L.m;
        ad.epsilon = mm_exp(ad.mu - sqrt(_m) / MM_NUTS_GAMMA * ad.h_bar);
        eta = mm_exp(-MM_NUTS_KAPPA * mm_log(_m));
        ad.epsilon_bar = mm_exp((1.0 - eta) * mm_log(ad.epsilon_bar) + eta * mm_log(ad.epsilon));
    } else {
        ad.epsilon = ad.epsilon_bar;
    }
    /* hist: the wave's histogram in LDS where the kernel has one (flushed once: a global atomic per chain and
     * transition, all on a dozen addresses, serialises in L2), else the global one */
    if (hist && L.q == 0)
        atomicAdd(&hist[L.depth < MM_NUTS_JMAX ? L.depth : MM_NUTS_JMAX], 1u);
}

/* the wave's depth histogram in LDS: zero / flush */
__device__ __forceinline__ void mm_lg_hist_zero(unsigned int *h, int lane)
{
    if (lane <= MM_NUTS_JMAX)
        h[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void mm_lg_hist_flush(const unsigned int *h, unsigned int *global_hist, int lane)
{
    __builtin_amdgcn_wave_barrier();
    if (global_hist && lane <= MM_NUTS_JMAX && h[lane] != 0u)
        atomicAdd(&global_hist[lane], h[lane]);
}

template <int D> __device__ __forceinline__ void mm_lg_write_row(const mm_lg_lane<D> &L, const mm_nuts_lg_args &a, unsigned long long row)
{
    double *dst = a.out + (L.cl * a.n_total + row) * D;
#pragma unroll
    for (int s = 0; s < D / 4; ++s)
        dst[4 * s + L.q] = L.x[s];
}

/* ------------------------------------------------------------------------------------------------------------------
 * Single launch: every wave keeps its 16 chains for all transitions (no compaction)
 * ---------------------------------------------------------------------------------------------------------------- */
template <int D> __global__ __launch_bounds__(64) void mm_nuts_lg_kernel(const mm_nuts_lg_args a)
{
    using Cfg = mm_lg_cfg<D>;
    constexpr int NS = Cfg::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    __shared__ unsigned int hist_lds[MM_NUTS_JMAX + 1];
    mm_lg_hist_zero(hist_lds, (int)(threadIdx.x & 63));
    mm_lg_lane<D> L;
    L.lane = threadIdx.x & 63;
    L.q = L.lane >> 4;
    L.cl = (unsigned long long)blockIdx.x * 16 + (L.lane & 15);
    L.active = L.cl < a.n_chains;
    L.chain = a.chain_offset + L.cl;
    L.n_lf = 0;
    L.n_leaf_iters = 0;
    L.m = a.m0;
    mm_lds_double *const lds = (mm_lds_double *)mm_lds_raw + L.lane;
    double *const scr = a.scratch + (size_t)blockIdx.x * Cfg::scratch_doubles_per_wave + L.lane;
    mm_lg_load_A<D>(L, a.mat);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        L.x[s] = L.active ? a.state[L.cl * D + 4 * s + L.q] : 0.0;
    mm_nuts_adapt<double> ad;
    if (L.active) {
        ad = a.adapt[L.cl];
    } else {
        ad.epsilon = 0.1;
        ad.epsilon_bar = 1.0;
        ad.h_bar = 0.0;
        ad.mu = 0.0;
    }
    unsigned int rows_out = 0;
#ifdef MM_LG_PROFILE
    for (int i = 0; i < 16; ++i)
        L.prof_acc[i] = 0;
    L.prof_t = __builtin_amdgcn_s_memtime();
#endif
    if (a.write_initial) {
        if (a.out && L.active)
            mm_lg_write_row<D>(L, a, a.out_t0 + rows_out);
        ++rows_out;
    }
    const unsigned int total = a.n_pre + a.n_rec;
    for (unsigned int t = 0; t < total; ++t) {
        L.m += 1;
        mm_lg_begin<D>(L, a);
        bool alive = L.active; /* the reference's `s` */
        MM_LG_TICK(L, 0);
        for (int j = 0; __ballot(alive) != 0ull; ++j)
            mm_lg_doubling<D>(L, a, j, alive, ad.epsilon, lds, scr);
        if (L.active)
            mm_lg_finish<D>(L, a, ad, a.depth_hist ? hist_lds : nullptr);
        if (t >= a.n_pre) {
            if (a.out && L.active)
                mm_lg_write_row<D>(L, a, a.out_t0 + rows_out);
            ++rows_out;
        }
        MM_LG_TICK(L, 5);
    }
#ifdef MM_LG_PROFILE
    if (L.lane == 0)
        for (int i = 0; i < 16; ++i)
            a.prof[(size_t)blockIdx.x * 16 + i] = L.prof_acc[i];
#endif
    mm_lg_hist_flush(hist_lds, a.depth_hist, L.lane);
    if (L.active) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            a.state[L.cl * D + 4 * s + L.q] = L.x[s];
        if (L.q == 0) {
            a.adapt[L.cl] = ad;
            if (a.n_leapfrog)
                a.n_leapfrog[L.cl] += L.n_lf;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Tree-depth compaction: one `begin` launch and one `double` launch per level j0 <= j < max_depth per transition
 * ---------------------------------------------------------------------------------------------------------------- */

/* the chain's transition is over: dual averaging, state, output row; or it wants doubling j_next: park it */
template <int D>
__device__ __forceinline__ void mm_lgc_hand_over(mm_lg_lane<D> &L, const mm_nuts_lg_args &a, bool valid, bool alive,
                                                 mm_nuts_adapt<double> &ad, int j_next)
{
    using Cfg = mm_lg_cfg<D>;
    constexpr int NS = Cfg::NS;
    const size_t st = (size_t)a.c_pad * 4;
    unsigned int *const count = a.counts + (size_t)(L.m & 1u) * (MM_NUTS_JMAX + 1) + j_next;
    /* one atomic per wave: chains are ranked by lane among the lanes q == 0 that append */
    const unsigned long long app = __ballot(valid && alive && L.q == 0);
    unsigned int base = 0;
    if (app != 0ull) {
        if (L.lane == (int)__ffsll((long long)app) - 1)
            base = atomicAdd(count, (unsigned int)__popcll(app));
        base = (unsigned int)__shfl((int)base, (int)__ffsll((long long)app) - 1, 64);
    }
    if (valid && alive) {
        double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            xs[s * st] = L.x[s];
        if (L.q == 0) {
            *mm_lg_rec_scalar<D>(a, L, Cfg::F_JOINT) = L.joint;
            *mm_lg_rec_scalar<D>(a, L, Cfg::F_LOGU) = L.logu;
            *mm_lg_rec_scalar<D>(a, L, Cfg::F_COUNTS) =
                __longlong_as_double((long long)((unsigned long long)L.n | ((unsigned long long)L.aux_k << 32)));
            const unsigned int rank = (unsigned int)__popcll(app & ((1ull << L.lane) - 1ull));
            a.lists[(size_t)j_next * a.c_pad + base + rank] = (unsigned int)L.cl;
        }
    } else if (valid) {
        mm_lg_finish<D>(L, a, ad, a.depth_hist);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            a.state[L.cl * D + 4 * s + L.q] = L.x[s];
        if (a.out && a.row != 0xffffffffu)
            mm_lg_write_row<D>(L, a, a.row);
        if (L.q == 0)
            a.adapt[L.cl] = ad;
    }
    if (valid && L.q == 0 && a.n_leapfrog)
        a.n_leapfrog[L.cl] += L.n_lf;
}

template <int D> __global__ __launch_bounds__(64) void mm_nuts_lgc_begin_kernel(const mm_nuts_lg_args a)
{
    using Cfg = mm_lg_cfg<D>;
    constexpr int NS = Cfg::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    mm_lg_lane<D> L;
    L.lane = threadIdx.x & 63;
    L.q = L.lane >> 4;
    L.cl = (unsigned long long)blockIdx.x * 16 + (L.lane & 15);
    L.active = L.cl < a.n_chains;
    L.chain = a.chain_offset + L.cl;
    L.n_lf = 0;
    L.n_leaf_iters = 0;
    L.m = a.m;
    /* the next transition's counters (this parity was last used two transitions ago) */
    if (blockIdx.x == 0 && L.lane <= MM_NUTS_JMAX)
        a.counts[(size_t)((a.m + 1u) & 1u) * (MM_NUTS_JMAX + 1) + L.lane] = 0u;
    mm_lds_double *const lds = (mm_lds_double *)mm_lds_raw + L.lane;
    double *const scr = a.scratch + (size_t)blockIdx.x * Cfg::scratch_doubles_per_wave + L.lane;
    mm_lg_load_A<D>(L, a.mat);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        L.x[s] = L.active ? a.state[L.cl * D + 4 * s + L.q] : 0.0;
    mm_nuts_adapt<double> ad;
    if (L.active) {
        ad = a.adapt[L.cl];
    } else {
        ad.epsilon = 0.1;
        ad.epsilon_bar = 1.0;
        ad.h_bar = 0.0;
        ad.mu = 0.0;
    }
    if (a.write_initial && a.out && L.active)
        mm_lg_write_row<D>(L, a, a.out_t0); /* row 0 = the initial position (nuts.rs:534) */
    mm_lg_begin<D>(L, a);
    bool alive = L.active;
    for (int j = 0; j < a.j0 && __ballot(alive) != 0ull; ++j)
        mm_lg_doubling<D>(L, a, j, alive, ad.epsilon, lds, scr);
    mm_lgc_hand_over<D>(L, a, L.active, alive, ad, a.j0);
}

template <int D> __global__ __launch_bounds__(64) void mm_nuts_lgc_double_kernel(const mm_nuts_lg_args a)
{
    using Cfg = mm_lg_cfg<D>;
    constexpr int NS = Cfg::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    const unsigned int cnt = a.counts[(size_t)(a.m & 1u) * (MM_NUTS_JMAX + 1) + a.j];
    const unsigned int base = blockIdx.x * 16u;
    if (base >= cnt)
        return;
    mm_lg_lane<D> L;
    L.lane = threadIdx.x & 63;
    L.q = L.lane >> 4;
    const unsigned int idx = base + (unsigned int)(L.lane & 15);
    const bool valid = idx < cnt;
    /* a short last wave repeats its first chain in the unused columns: read-only there */
    L.cl = a.lists[(size_t)a.j * a.c_pad + (valid ? idx : base)];
    L.active = valid;
    L.chain = a.chain_offset + L.cl;
    L.n_lf = 0;
    L.n_leaf_iters = 0;
    L.m = a.m;
    mm_lds_double *const lds = (mm_lds_double *)mm_lds_raw + L.lane;
    double *const scr = a.scratch + (size_t)blockIdx.x * Cfg::scratch_doubles_per_wave + L.lane;
    mm_lg_load_A<D>(L, a.mat);
    const size_t st = (size_t)a.c_pad * 4;
    const double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        L.x[s] = xs[s * st];
    L.joint = *mm_lg_rec_scalar<D>(a, L, Cfg::F_JOINT);
    L.logu = *mm_lg_rec_scalar<D>(a, L, Cfg::F_LOGU);
    const unsigned long long pk = (unsigned long long)__double_as_longlong(*mm_lg_rec_scalar<D>(a, L, Cfg::F_COUNTS));
    L.n = (unsigned int)pk;
    L.aux_k = (unsigned int)(pk >> 32);
    L.aux_have = 0xffffffffu;
    L.aux_blk.w[0] = L.aux_blk.w[1] = L.aux_blk.w[2] = L.aux_blk.w[3] = 0u;
    L.alpha = 0.0;
    L.n_alpha = 0;
    L.depth = a.j;
    mm_nuts_adapt<double> ad = a.adapt[L.cl];
    bool alive = valid;
    mm_lg_doubling<D>(L, a, a.j, alive, ad.epsilon, lds, scr);
    mm_lgc_hand_over<D>(L, a, valid, alive, ad, a.j + 1);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Persistent scheduler: compaction without launch boundaries.  With one launch per tree level (above) every launch
 * lasts as long as its longest wave and the late levels of a transition hold too few chains to fill the GPU; here
 * one resident wave per SIMD loops { take up to 16 chains from one queue, run that unit of work, append each chain
 * to the queue of its next unit }, units being "begin a transition and run doublings 0 .. j0-1" (queue 0) and
 * "doubling j" (queue 1 + j).  Chains advance independently -- a chain parked before a 512-leaf doubling does not
 * hold back the others, which go on to their next transitions -- so every chain carries its own transition count.
 * Queues are rings of chain indices; an append reserves entries with one atomic add on `tail` per wave and then
 * fills them (CAS from EMPTY), a take claims entries with a CAS on `head` and then waits for each to be filled
 * (exchange with EMPTY).  Data handed from wave to wave goes through write-through stores and cache-bypassing loads
 * (mm_lg_ld / mm_lg_st<true>), completed (s_waitcnt) before the chain is appended.
 * ---------------------------------------------------------------------------------------------------------------- */
template <int D> __global__ void mm_nuts_lgq_init_kernel(const mm_nuts_lg_args a, size_t scalar_base)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_slots = (size_t)MM_LGQ_SHARDS * MM_LGQ_NQ * a.c_pad;
    /* queue 0 of shard s starts with the chains of the 16-blocks b = s (mod SHARDS), in order; entry e of a ring is
     * tagged with its lap e / c_pad + 1, so the zero-filled rest never looks like an entry */
    if (i < n_slots) {
        const size_t shard = i / ((size_t)MM_LGQ_NQ * a.c_pad), r = i % ((size_t)MM_LGQ_NQ * a.c_pad);
        unsigned int v = 0u;
        if (r < a.c_pad) {
            const size_t blk = (r / 16) * MM_LGQ_SHARDS + shard, chain = blk * 16 + (r % 16);
            if (chain < a.n_chains)
                v = (1u << MM_LGQ_ID_BITS) | (unsigned int)chain;
        }
        a.slots[i] = v;
    }
    if (i < a.c_pad)
        a.rec[scalar_base + (size_t)3 * a.c_pad + i] = __longlong_as_double((long long)a.m0); /* F_M */
    if (i < MM_LGQ_SHARDS) {
        for (int q = 0; q < 16; ++q)
            a.ctrl->w[i][q] = 0ull;
        /* chains of shard i: full 16-blocks b = i (mod SHARDS) plus possibly a short last one */
        const size_t n_blk = (a.n_chains + 15) / 16;
        size_t cnt = 0;
        for (size_t b = i; b < n_blk; b += MM_LGQ_SHARDS)
            cnt += (b * 16 + 16 <= a.n_chains) ? 16 : (a.n_chains - b * 16);
        a.ctrl->w[i][0] = (unsigned long long)cnt << 32;
    }
    if (i == 0) {
        a.ctrl->remaining = a.n_chains;
        a.ctrl->error = 0ull;
        a.ctrl->stat_units = a.ctrl->stat_chains = a.ctrl->stat_polls = a.ctrl->stat_leaf_iters = 0ull;
        a.ctrl->stat_t[0] = a.ctrl->stat_t[1] = a.ctrl->stat_t[2] = a.ctrl->stat_t[3] = 0ull;
    }
}

#define MM_LGQ_SPIN_LIMIT (1u << 23) /* polls of ~1000 cycles: several seconds without progress */

__device__ __forceinline__ unsigned int *mm_lgq_slot(const mm_nuts_lg_args &a, int shard, int qi, unsigned int e, unsigned int *tag)
{
    const unsigned int cap = (unsigned int)a.c_pad;
    *tag = (e / cap + 1u) << MM_LGQ_ID_BITS;
    return a.slots + ((size_t)shard * MM_LGQ_NQ + (size_t)qi) * a.c_pad + (size_t)(e % cap);
}

/* append the chains flagged by `pred` (their q == 0 lanes) to queue qi of `shard`: one atomic add reserves the
 * entries, write-through stores fill them */
template <int D>
__device__ __forceinline__ void mm_lgq_append(const mm_lg_lane<D> &L, const mm_nuts_lg_args &a, int shard, int qi, bool pred)
{
    const unsigned long long app = __ballot(pred && L.q == 0);
    if (app == 0ull)
        return;
    const int leader = (int)__ffsll((long long)app) - 1;
    unsigned long long base = 0ull;
    if (L.lane == leader)
        base = atomicAdd(&a.ctrl->w[shard][qi], (unsigned long long)__popcll(app) << 32) >> 32;
    base = (unsigned long long)__shfl((long long)base, leader, 64);
    if (pred && L.q == 0) {
        unsigned int tag;
        unsigned int *slot =
            mm_lgq_slot(a, shard, qi, (unsigned int)base + (unsigned int)__popcll(app & ((1ull << L.lane) - 1ull)), &tag);
        __hip_atomic_store(slot, tag | (unsigned int)L.cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* OCC = 1: one wave per workgroup, A-operand blocks in registers.  OCC = 2: workgroups of 8 waves (two per SIMD) that
 * share ONE LDS copy of the A-operand blocks (8 KB at D = 32) and are otherwise independent: a wave's index in the launch
 * is blockIdx.x * WPB + its index in the workgroup */
template <int D, int OCC = 1>
__global__ __launch_bounds__(OCC == 2 ? 512 : 64, OCC == 2 ? 2 : 1) void mm_nuts_lgq_kernel(const mm_nuts_lg_args a)
{
    using Cfg = mm_lg_cfg<D, OCC>;
    constexpr int NS = Cfg::NS;
    constexpr unsigned int WPB = OCC == 2 ? 8u : 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];
    __shared__ unsigned int hist_all[WPB][MM_NUTS_JMAX + 1];
    const unsigned int wib = threadIdx.x >> 6, wid = blockIdx.x * WPB + wib;
    unsigned int *const hist_lds = hist_all[wib];
    mm_lg_hist_zero(hist_lds, (int)(threadIdx.x & 63));
    mm_lg_lane<D> L;
    L.lane = threadIdx.x & 63;
    L.q = L.lane >> 4;
    L.cl = 0;
    L.n_leaf_iters = 0;
    /* LDS: [OCC 2: the A-operand blocks, NT NS slots of 64 doubles][per wave: Cfg::lds_slots slots] */
    constexpr size_t A_SLOTS = OCC == 2 ? (size_t)(D / 16) * NS : 0;
    mm_lds_double *const lds = (mm_lds_double *)mm_lds_raw + (A_SLOTS + (size_t)wib * Cfg::lds_slots) * 64 + L.lane;
    double *const scr = a.scratch + (size_t)wid * Cfg::scratch_doubles_per_wave + L.lane;
    L.Alds = (const mm_lds_double *)mm_lds_raw + L.lane;
    if (OCC == 2) {
        if (wib == 0) {
            const int c = L.lane & 15;
#pragma unroll
            for (int t = 0; t < D / 16; ++t)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    ((mm_lds_double *)mm_lds_raw)[(size_t)(t * NS + s) * 64 + L.lane] = a.mat[(size_t)(16 * t + c) * D + 4 * s + L.q];
        }
        __syncthreads(); /* the only workgroup-wide rendezvous of the kernel */
    } else {
        mm_lg_load_A<D>(L, a.mat);
    }
    const size_t st = (size_t)a.c_pad * 4;
    const unsigned int m_end = a.m0 + a.n_pre + a.n_rec;
    mm_lgq_ctrl *const ctrl = a.ctrl;
    const unsigned int col = (unsigned int)(L.lane & 15);
    const int home = (int)(wid % MM_LGQ_SHARDS);
    unsigned long long st_units = 0, st_chains = 0, st_polls = 0, st_t[4] = {0, 0, 0, 0};
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#define MM_LGQ_T(i)                                                                                               \
    do {                                                                                                          \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();                                               \
        st_t[i] += _n - t_prev;                                                                                   \
        t_prev = _n;                                                                                              \
    } while (0)

    mm_lg_edges<D> E; /* OCC 1: both edges of the unit's chains */
    /* chains this wave keeps from its last unit (they all want doubling keep_level): their columns stay theirs */
    bool keep = false;
    int keep_level = 0;

    for (;;) {
        /* ---- compose the next unit: the kept chains plus entries of the queue of their level; with nothing kept (or
         *      too few to be worth a unit), the deepest level that fills a wave -- own shard first, then the others --
         *      else, after a while, the fullest queue ---- */
        const unsigned int keep16 = (unsigned int)(__ballot(keep && L.q == 0) & 0xffffull);
        const unsigned int n_keep = (unsigned int)__popc(keep16);
        const int keep_q = 1 + keep_level;
        int qi = -1, shard = home;
        unsigned int n_take = 0, first = 0;
        bool give_back = false; /* too few kept chains and nothing to top them up with: queue them (own shard) */
        bool quit = false;
        /* claim `want` entries of queue q of shard sh, whose word was read as `old`; retries on the value the failed
         * compare-and-swap returned while the queue still holds enough */
        auto claim = [&](int sh, int q, unsigned long long old, unsigned int want_max, unsigned int want_min) -> bool {
            for (int tries = 0; tries < 4; ++tries) {
                const unsigned int av = (unsigned int)(old >> 32) - (unsigned int)old;
                const unsigned int want = av < want_max ? av : want_max;
                if (want < want_min || want == 0u)
                    return false;
                unsigned long long got = old;
                if (L.lane == 0)
                    got = atomicCAS(&ctrl->w[sh][q], old, old + want);
                got = (unsigned long long)__shfl((long long)got, 0, 64);
                if (got == old) {
                    shard = sh;
                    qi = q;
                    n_take = want;
                    first = (unsigned int)old;
                    return true;
                }
                old = got;
            }
            return false;
        };
        for (unsigned int polls = 0;;) {
            if (n_keep == 16u) { /* every chain of the last unit goes on: nothing to ask the queues */
                qi = keep_q;
                shard = home;
                break;
            }
            /* the state of the own shard's queues in one load: lane i < 16 reads word i */
            unsigned long long w = 0ull;
            if (L.lane < 16)
                w = __hip_atomic_load(&ctrl->w[home][L.lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool is_queue = L.lane == 0 || (L.lane > a.j0 && L.lane <= a.max_depth);
            unsigned int avail = is_queue ? (unsigned int)(w >> 32) - (unsigned int)w : 0u;
            if (n_keep > 0u && !give_back) {
                /* stay with the kept chains if the queue of their level tops them up to min_unit (16: a full unit) or more */
                unsigned int want = (unsigned int)__shfl((int)avail, keep_q, 64);
                want = want < 16u - n_keep ? want : 16u - n_keep;
                if (n_keep + want >= a.min_unit) {
                    if (want > 0u)
                        (void)claim(home, keep_q, (unsigned long long)__shfl((long long)w, keep_q, 64), want, 1u);
                    qi = keep_q; /* a lost race: the kept chains run alone */
                    shard = home;
                    break;
                }
                give_back = true;
            }
            /* the deepest level that fills a wave (the kept chains about to be queued count for their level) */
            const unsigned int eff = avail + ((give_back && L.lane == keep_q) ? n_keep : 0u);
            const unsigned long long full = __ballot(eff >= 16u) & 0xffffull;
            if (full != 0ull) {
                const int best = 63 - (int)__clzll((long long)full);
                const unsigned long long old = (unsigned long long)__shfl((long long)w, best, 64);
                if (give_back && best == keep_q) {
                    /* the kept chains complete a wave after all: top up instead of queueing them */
                    give_back = false;
                    (void)claim(home, best, old, 16u - n_keep, 1u);
                    qi = best;
                    shard = home;
                    break;
                }
                if (claim(home, best, old, 16u, 16u))
                    break;
                continue; /* lost the race for it: look again */
            }
            /* nothing here fills a wave: the other shards, nearest first */
            bool found = false;
            for (int k = 1; k < MM_LGQ_SHARDS && !found; ++k) {
                const int sh = (home + k) % MM_LGQ_SHARDS;
                unsigned long long wo = 0ull;
                if (L.lane < 16)
                    wo = __hip_atomic_load(&ctrl->w[sh][L.lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int avo = is_queue ? (unsigned int)(wo >> 32) - (unsigned int)wo : 0u;
                const unsigned long long fo = __ballot(avo >= 16u) & 0xffffull;
                if (fo != 0ull) {
                    const int best = 63 - (int)__clzll((long long)fo);
                    found = claim(sh, best, (unsigned long long)__shfl((long long)wo, best, 64), 16u, 16u);
                }
            }
            if (found)
                break;
            if (polls >= a.patience) {
                /* waited long enough: the fullest queue of the own shard, whatever it holds; else of another shard */
                unsigned int mx = 0u;
                int best = -1;
                for (int q = 0; q <= a.max_depth; ++q) {
                    const unsigned int e = (unsigned int)__shfl((int)eff, q, 64);
                    if (e > mx) {
                        mx = e;
                        best = q;
                    }
                }
                if (best >= 0) {
                    const unsigned long long old = (unsigned long long)__shfl((long long)w, best, 64);
                    if (give_back && best == keep_q) {
                        give_back = false;
                        (void)claim(home, best, old, 16u - n_keep, 1u);
                        qi = best;
                        shard = home;
                        break;
                    }
                    if (claim(home, best, old, 16u, 1u))
                        break;
                } else {
                    for (int k = 1; k < MM_LGQ_SHARDS && !found; ++k) {
                        const int sh = (home + k) % MM_LGQ_SHARDS;
                        unsigned long long wo = 0ull;
                        if (L.lane < 16)
                            wo = __hip_atomic_load(&ctrl->w[sh][L.lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned int avo = is_queue ? (unsigned int)(wo >> 32) - (unsigned int)wo : 0u;
                        const unsigned long long fo = __ballot(avo >= 1u) & 0xffffull;
                        if (fo != 0ull) {
                            const int bq = 63 - (int)__clzll((long long)fo);
                            found = claim(sh, bq, (unsigned long long)__shfl((long long)wo, bq, 64), 16u, 1u);
                        }
                    }
                    if (found)
                        break;
                }
            }
            /* idle: is the run over? */
            unsigned long long rem = 1ull, err = 0ull;
            if (L.lane == 0) {
                rem = __hip_atomic_load(&ctrl->remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                err = __hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            rem = (unsigned long long)__shfl((long long)rem, 0, 64);
            err = (unsigned long long)__shfl((long long)err, 0, 64);
            if ((rem == 0ull || err != 0ull) && !(give_back && n_keep > 0u)) {
                quit = true;
                break;
            }
            if (err != 0ull) {
                quit = true;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
            ++st_polls;
            if (++polls > MM_LGQ_SPIN_LIMIT) {
                if (L.lane == 0)
                    atomicExch(&ctrl->error, 1ull);
                quit = true;
                break;
            }
        }
        if (give_back) {
            /* their state is already parked in the records (hand-over below) */
            mm_lgq_append<D>(L, a, home, keep_q, keep);
            keep = false;
        }
        MM_LGQ_T(0);
#if MM_LG_UNIFORM_J
        /* the queue (= the level of the unit) is the same in all 64 lanes by construction: say so, and everything derived from it
         * -- the kind of unit, the doublings it runs, the hand-over's queue -- is scalar control flow (mm_lg_doubling, round 6) */
        qi = __builtin_amdgcn_readfirstlane(qi);
        shard = __builtin_amdgcn_readfirstlane(shard);
#endif
        if (quit || qi < 0)
            break;
        const bool use_keep = n_keep > 0u && !give_back; /* wave-uniform; then qi == keep_q in the own shard */
        /* free columns take the claimed entries in order */
        const unsigned int free16 = use_keep ? (~keep16 & 0xffffu) : 0xffffu;
        const unsigned int my_rank = (unsigned int)__popc(free16 & ((1u << col) - 1u));
        const bool is_new = ((free16 >> col) & 1u) && my_rank < n_take;
        unsigned int id = (unsigned int)L.cl;
        if (is_new && L.q == 0) {
            /* wait for the appender's store: the entry carries the tag of its lap */
            unsigned int tag;
            const unsigned int *slot = mm_lgq_slot(a, shard, qi, first + my_rank, &tag);
            unsigned int v, spins = 0;
            while (((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> MM_LGQ_ID_BITS) !=
                   (tag >> MM_LGQ_ID_BITS)) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > MM_LGQ_SPIN_LIMIT) {
                    atomicExch(&ctrl->error, 3ull);
                    break;
                }
            }
            id = v & ((1u << MM_LGQ_ID_BITS) - 1u);
        }
        id = (unsigned int)__shfl((int)id, (int)col, 64); /* from the chain's q == 0 lane */
        const bool valid = is_new || (use_keep && ((keep16 >> col) & 1u));
        const unsigned long long vmask = __ballot(valid);
        if (vmask == 0ull)
            continue; /* a top-up lost its race and nothing was kept: look again */
        const unsigned int id_any = (unsigned int)__shfl((int)id, (int)__ffsll((long long)vmask) - 1, 64);
        /* unused columns repeat a valid chain: read-only there */
        L.cl = valid ? id : id_any;
        L.active = valid;
        L.chain = a.chain_offset + L.cl;
        L.n_lf = 0;
        keep = false;
        st_units += 1;
        st_chains += (unsigned long long)__popcll(vmask) / 4ull;
        MM_LGQ_T(1);
        /* what the previous owner of these chains wrote is read with coherent loads (mm_lg_ld<true>), whose addresses
         * depend on the index just taken from the queue */
        mm_nuts_adapt<double> ad = mm_lg_ld_adapt<true>(&a.adapt[L.cl]);
        bool alive = valid;
        bool ran_on = false; /* the unit ran more doublings than it was taken for */
        int j_first, j_next;
        if (qi == 0) {
            const unsigned int m_done = (unsigned int)__double_as_longlong(mm_lg_ld<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_M)));
            L.m = m_done + 1u;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                L.x[s] = mm_lg_ld<true>(&a.state[L.cl * D + 4 * s + L.q]);
            if (a.write_initial && a.out && valid && m_done == a.m0)
                mm_lg_write_row<D>(L, a, a.out_t0); /* row 0 = the initial position (nuts.rs:534) */
            if constexpr (OCC == 1)
                mm_lg_begin<D, true, OCC, true>(L, a, E); /* both edges stay in registers until the unit parks its chains */
            else
                mm_lg_begin<D, true, OCC>(L, a);
            if (OCC == 2 && valid) {
                /* two waves per SIMD: the current sample lives in the chain's record from here to the end of the transition */
                double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    mm_lg_st<true>(&xs[s * st], L.x[s]);
            }
            j_first = 0;
            j_next = a.j0;
        } else {
            const int j = qi - 1;
            if (OCC != 2) {
                const double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    L.x[s] = mm_lg_ld<true>(&xs[s * st]);
            }
            if constexpr (OCC == 1) {
                /* both edges, in the same batch of loads as everything else of the chain: one memory latency per unit */
                const double *xm = mm_lg_rec_vec<D>(a, L, Cfg::V_XM), *pm = mm_lg_rec_vec<D>(a, L, Cfg::V_PM),
                             *gm = mm_lg_rec_vec<D>(a, L, Cfg::V_GM), *xp = mm_lg_rec_vec<D>(a, L, Cfg::V_XP),
                             *pp = mm_lg_rec_vec<D>(a, L, Cfg::V_PP), *gp = mm_lg_rec_vec<D>(a, L, Cfg::V_GP);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    E.cx[s] = mm_lg_ld<true>(&xp[s * st]);
                    E.cp[s] = mm_lg_ld<true>(&pp[s * st]);
                    E.cg[s] = mm_lg_ld<true>(&gp[s * st]);
                    E.ox[s] = mm_lg_ld<true>(&xm[s * st]);
                    E.op[s] = mm_lg_ld<true>(&pm[s * st]);
                    E.og[s] = mm_lg_ld<true>(&gm[s * st]);
                }
                E.cur_neg = false;
            }
            L.joint = mm_lg_ld<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_JOINT));
            L.logu = mm_lg_ld<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_LOGU));
            const unsigned long long pk =
                (unsigned long long)__double_as_longlong(mm_lg_ld<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_COUNTS)));
            L.n = (unsigned int)pk;
            L.aux_k = (unsigned int)(pk >> 32);
            L.aux_have = 0xffffffffu;
            L.aux_blk.w[0] = L.aux_blk.w[1] = L.aux_blk.w[2] = L.aux_blk.w[3] = 0u;
            L.alpha = 0.0;
            L.n_alpha = 0;
            L.depth = j;
            L.m = (unsigned int)__double_as_longlong(mm_lg_ld<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_M))) + 1u;
            j_first = j;
            j_next = j + 1;
        }
        /* one copy of the doubling code for both kinds of unit (the kernel's hot loop should stay small) */
        if constexpr (OCC == 2) {
            mm_lg_call_io io;
            io.joint = L.joint;
            io.logu = L.logu;
            io.alpha = L.alpha;
            io.chain = L.chain;
            io.cl = L.cl;
            io.n_lf = L.n_lf;
            io.n_leaf_iters = L.n_leaf_iters;
            io.n = L.n;
            io.aux_k = L.aux_k;
            io.aux_have = L.aux_have;
            io.n_alpha = L.n_alpha;
            io.m = L.m;
            io.depth = L.depth;
            io.lane = L.lane;
            io.q = L.q;
            io.aux_blk = L.aux_blk;
            io.alive = alive ? 1 : 0;
            mm_lg_call_env env;
            env.rec = (mm_glb_double *)a.rec;
            env.c_pad = a.c_pad;
            env.seed = a.seed;
            env.max_depth = a.max_depth;
            for (int j = j_first; j < j_next && __ballot(io.alive != 0) != 0ull; ++j)
                io = mm_lg_doubling_ool<D>(io, env, j, ad.epsilon, lds, (mm_glb_double *)scr, L.Alds);
            L.alpha = io.alpha;
            L.n_lf = io.n_lf;
            L.n_leaf_iters = io.n_leaf_iters;
            L.n = io.n;
            L.aux_k = io.aux_k;
            L.aux_have = io.aux_have;
            L.n_alpha = io.n_alpha;
            L.depth = io.depth;
            L.aux_blk = io.aux_blk;
            alive = io.alive != 0;
        } else {
            /* the doublings the unit was taken for -- and on: a full unit whose 16 chains ALL go on is the next unit as it
             * stands, it runs without parking the chains, asking the queues and loading them again (some 25 000 cycles of
             * memory round trips; rare where trees differ in depth, the rule where they do not) */
            int j = j_first;
            while (__ballot(alive) != 0ull && (j < j_next || (j < a.max_depth && __ballot(alive) == ~0ull))) {
                mm_lg_doubling<D, true, OCC, true>(L, a, j, alive, ad.epsilon, lds, scr, E);
                ++j;
            }
            if (j > j_next) {
                ran_on = true;
                j_next = j;
            }
        }

        MM_LGQ_T(2);
        /* ---- hand the chains on ---- */
        bool again = false; /* finished this transition, more to go */
        if (valid && alive) {
            if (OCC != 2) { /* OCC 2: the record already holds the current sample */
                double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    mm_lg_st<true>(&xs[s * st], L.x[s]);
            }
            if constexpr (OCC == 1) {
                /* the edges the unit changed: the one extended last; both after the several doublings of a unit that began
                 * the transition (the record holds neither yet) or ran on */
                double *cxr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_XM : Cfg::V_XP),
                       *cpr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_PM : Cfg::V_PP),
                       *cgr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_GM : Cfg::V_GP);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    mm_lg_st<true>(&cxr[s * st], E.cx[s]);
                    mm_lg_st<true>(&cpr[s * st], E.cp[s]);
                    mm_lg_st<true>(&cgr[s * st], E.cg[s]);
                }
                if (qi == 0 || ran_on) {
                    double *oxr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_XP : Cfg::V_XM),
                           *opr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_PP : Cfg::V_PM),
                           *ogr = mm_lg_rec_vec<D>(a, L, E.cur_neg ? Cfg::V_GP : Cfg::V_GM);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        mm_lg_st<true>(&oxr[s * st], E.ox[s]);
                        mm_lg_st<true>(&opr[s * st], E.op[s]);
                        mm_lg_st<true>(&ogr[s * st], E.og[s]);
                    }
                }
            }
            if (L.q == 0) {
                mm_lg_st<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_JOINT), L.joint);
                mm_lg_st<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_LOGU), L.logu);
                mm_lg_st<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_COUNTS),
                               __longlong_as_double((long long)((unsigned long long)L.n | ((unsigned long long)L.aux_k << 32))));
            }
        } else if (valid) {
            mm_lg_finish<D>(L, a, ad, a.depth_hist ? hist_lds : nullptr);
            if (OCC == 2) {
                const double *xs = mm_lg_rec_vec<D>(a, L, Cfg::V_X);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    L.x[s] = mm_lg_ld<true>(&xs[s * st]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
                mm_lg_st<true>(&a.state[L.cl * D + 4 * s + L.q], L.x[s]);
            const unsigned int t = L.m - a.m0 - 1u;
            if (a.out && t >= a.n_pre)
                mm_lg_write_row<D>(L, a, (unsigned long long)a.out_t0 + (a.write_initial ? 1u : 0u) + (t - a.n_pre));
            if (L.q == 0) {
                mm_lg_st_adapt<true>(&a.adapt[L.cl], ad);
                mm_lg_st<true>(mm_lg_rec_scalar<D>(a, L, Cfg::F_M), __longlong_as_double((long long)L.m));
            }
            again = L.m < m_end;
        }
        if (valid && L.q == 0 && a.n_leapfrog)
            atomicAdd(&a.n_leapfrog[L.cl], L.n_lf);
        /* the write-through stores above have completed before a chain becomes visible in a queue */
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        /* chains that go on doubling stay with this wave (their records are parked all the same: the next unit loads
         * every column alike); finished ones go back to queue 0 or retire */
        keep = valid && alive;
        keep_level = j_next;
        mm_lgq_append<D>(L, a, home, 0, again);
        const unsigned long long fin = __ballot(valid && !alive && !again && L.q == 0);
        if (fin != 0ull && L.lane == (int)__ffsll((long long)fin) - 1)
            atomicAdd(&ctrl->remaining, 0ull - (unsigned long long)__popcll(fin));
        MM_LGQ_T(3);
    }
    mm_lg_hist_flush(hist_lds, a.depth_hist, L.lane);
    if (L.lane == 0) {
        for (int i = 0; i < 4; ++i)
            atomicAdd(&ctrl->stat_t[i], st_t[i]);
        atomicAdd(&ctrl->stat_units, st_units);
        atomicAdd(&ctrl->stat_chains, st_chains);
        atomicAdd(&ctrl->stat_polls, st_polls);
        atomicAdd(&ctrl->stat_leaf_iters, L.n_leaf_iters);
    }
#undef MM_LGQ_T
}

/* occ: waves per SIMD the scheduler kernel is built for (1 or 2); the caller sizes n_waves and the scratch accordingly */
template <int D> hipError_t mm_launch_nuts_lgq(const mm_nuts_lg_args &a, unsigned int n_waves, int occ, hipStream_t stream)
{
    const size_t n_slots = (size_t)MM_LGQ_SHARDS * MM_LGQ_NQ * a.c_pad;
    const size_t scalar_base = (size_t)mm_lg_cfg<D>::n_vec * D * a.c_pad;
    hipLaunchKernelGGL((mm_nuts_lgq_init_kernel<D>), dim3((unsigned int)((n_slots + 255) / 256)), dim3(256), 0, stream, a, scalar_base);
    using Cfg1 = mm_lg_cfg<D, 1>;
    using Cfg2 = mm_lg_cfg<D, 2>;
    if (occ == 2) {
        /* workgroups of 8 waves: n_waves is rounded down to whole workgroups (the caller asks for two per SIMD) */
        const size_t lds2 = ((size_t)(D / 16) * (D / 4) * 64 + 8 * (size_t)Cfg2::lds_slots * 64) * sizeof(double);
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mm_nuts_lgq_kernel<D, 2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            if (e != hipSuccess)
                return e;
            if (dev < 64)
                attr_set |= 1ull << dev;
        }
        hipLaunchKernelGGL((mm_nuts_lgq_kernel<D, 2>), dim3(n_waves / 8 ? n_waves / 8 : 1), dim3(512), lds2, stream, a);
    } else
        hipLaunchKernelGGL((mm_nuts_lgq_kernel<D, 1>), dim3(n_waves), dim3(64), Cfg1::lds_bytes, stream, a);
    return hipGetLastError();
}

template <int D> hipError_t mm_launch_nuts_lg(const mm_nuts_lg_args &a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 15) / 16);
    hipLaunchKernelGGL((mm_nuts_lg_kernel<D>), dim3(grid), dim3(64), mm_lg_cfg<D>::lds_bytes, stream, a);
    return hipGetLastError();
}

/* one transition with compaction: a.m, a.row, a.write_initial set by the caller */
template <int D> hipError_t mm_launch_nuts_lgc_transition(mm_nuts_lg_args a, hipStream_t stream)
{
    const unsigned int grid = (unsigned int)((a.n_chains + 15) / 16);
    hipLaunchKernelGGL((mm_nuts_lgc_begin_kernel<D>), dim3(grid), dim3(64), mm_lg_cfg<D>::lds_bytes, stream, a);
    for (int j = a.j0; j < a.max_depth; ++j) {
        a.j = j;
        hipLaunchKernelGGL((mm_nuts_lgc_double_kernel<D>), dim3(grid), dim3(64), mm_lg_cfg<D>::lds_bytes, stream, a);
    }
    return hipGetLastError();
}

#endif /* MM_NUTS_LG_H */
