/*
 * mm_split_kernels.h -- the many-chain MH / HMC kernel with several waves per SIMD: noise waves and transition waves.
 *
 * Why.  One chain per lane makes a transition a long dependent stream of f32 VALU instructions.  On gfx950 a wave
 * issues at most one VALU instruction per ~4.4 cycles, whatever its instruction-level parallelism, while a SIMD
 * issues one per ~2.2 cycles when TWO waves feed it (tools/issue_rate.hip: plain FMA / integer streams 4.4 -> 2.2
 * cycles per instruction per SIMD from one wave to two; 32 x 32 multiplies stay at ~4.3).  BASELINE's chain counts
 * (65 536 = 1024 waves = ONE per SIMD) leave half of every SIMD's issue slots empty in mm_run_kernel, and more chains
 * per GPU is not what configs 2 / 3 ask for.  So a chain is given two lanes in two waves of one SIMD:
 *
 *   workgroup = 512 threads = 8 waves for 256 chains; waves w and w + 4 sit on the same SIMD (a workgroup's waves
 *   are dealt to the SIMDs cyclically) and share 64 chains:
 *     wave w     ("transition wave"): state in registers, runs mm_hmc_step_noise / mm_mh_step_noise -- exactly the
 *                arithmetic of mm_run_kernel, bit for bit -- and stages the collected states in an LDS tile;
 *     wave w + 4 ("noise wave"): draws the transitions' noise -- a pure function of (seed, chain, iteration), so it
 *                needs nothing from its partner -- two iterations at a time (mm_draw_noise_pair) into an LDS ring, and
 *                writes the full tiles to HBM (the stores' issue latency costs the transition wave nothing).
 *   The ring has two halves of RB transitions, there are two tiles per pair; one s_barrier per RB transitions hands a
 *   filled ring half and a full tile over and the drained / flushed ones back.  Where the noise outweighs the
 *   transition (MH before round 5: ~115 of ~140 instructions) the transition wave draws QP of the RB / 2 noise pairs of a
 *   batch itself, so that both waves carry about the same load (mm_split_mh_qp; round 5: with the table-free accept test
 *   and the MH sampler's paired stream -- one Philox block per two transitions, mm_rng.h -- the noise is the light role
 *   and keeps all of it, QP = 0).
 *   NN > 1 (f32 instances: 3): NN noise waves per pair in a 256 (1 + NN)-thread workgroup, the noise pairs of a batch dealt
 *   to them in turn -- neither role is near its own issue bound, both wait on LDS table reads and dependent issue, and
 *   with the chain count fixing the number of transition waves the noise is the part that can be spread over more
 *   waves per SIMD (config 3: 0.2057 -> 0.1924 ms, config 2 MH 0.275 -> 0.260 ms).
 *
 * Same functions of (seed, chain, iteration) as every other variant: samples, states and accept counts are
 * bit-identical to mm_run_kernel and to the host build (tests/test_gpu_parity.py::test_kernel_variants_agree).
 *
 * Replaces (like mm_kernels.h): run_chain core.rs:55-73, ChainRunner::run core.rs:176-186, HMC::run hmc.rs:137-158.
 */
#ifndef MM_SPLIT_KERNELS_H
#define MM_SPLIT_KERNELS_H

#include "mm_kernels.h"

/* LDS plan of one workgroup (4 wave pairs): [f32 tables][4 x 2 tiles][4 rings of 2 x RB transitions].
 * Who writes the tiles out is decided by the balance of the two roles.  HMC: the transition wave is the busy one either
 * way and flushes its own (single) tile -- measured faster than handing it over (config 3: 0.225 vs 0.233 ms).  MH: the
 * noise dominates, the transition wave takes a share of it and the noise wave the flushing; two tiles per pair.
 * Candidates (ring half RB, elements per chain and tile) in order of preference, the first that fits 160 KB wins: HMC wants
 * large tiles (few, long store bursts), MH wants RB = 8 (short transitions: barriers dearer than stores). */
/* NT, tiles per pair: 2 = the last noise wave writes the tiles out while the transition wave stages into the other one
 * (MH; HMC at D <= 3 with a ring half of 8: mm_split_hmc_pf), 1 = the transition wave flushes its own tile (HMC otherwise).
 * -DMM_SPLIT_HMC_PFLUSH=1 (tools/split_probe.hip) makes 2 the default for HMC as well. */
#ifndef MM_SPLIT_HMC_PFLUSH
#define MM_SPLIT_HMC_PFLUSH 0
#endif
#define MM_SPLIT_NTILE(MH) (((MH) || MM_SPLIT_HMC_PFLUSH) ? 2 : 1)
/* PW: wave pairs (64 chains each) per workgroup -- 4 (one workgroup per CU, its barrier joins 4 (1 + NN) waves) or 1
 * (round 6: workgroup = ONE pair, 1 + NN waves, four workgroups per CU each with a quarter of the LDS and its own copy of
 * the table; the barrier joins 1 + NN waves and a pair never waits for another pair's transition wave) */
template <class T, int D, int RB, int WANT, int NTILE, int PW = 4> struct mm_split_try {
    static constexpr int epl = 16 / (int)sizeof(T);
    static constexpr int nw = (D + 1 + epl - 1) / epl * epl; /* noise row of one chain: z[D], ln u, padded to 16 bytes */
    static constexpr size_t row_bytes = (size_t)64 * nw * sizeof(T); /* one transition of one wave pair */
    static constexpr size_t ring_bytes = 2 * (size_t)RB * row_bytes;
    static constexpr int tile_want = WANT / (int)(sizeof(T) / 4) / D;
    static constexpr int tile_t = tile_want / RB * RB; /* a whole number of batches, so that full tiles take the fast flush */
    using Tile = mm_tile_t<T, D, (tile_t > 0 ? tile_t : RB)>;
    static constexpr size_t tile_bytes = (Tile::lds_bytes_per_wave + 15) / 16 * 16;
    static constexpr size_t table_bytes = (Tile::lds_bytes_table + 15) / 16 * 16;
    static constexpr size_t lds_bytes = table_bytes + PW * NTILE * tile_bytes + PW * ring_bytes;
    static constexpr int ntile = NTILE, pw = PW;
    static constexpr size_t lds_budget = (size_t)160 * 1024 * PW / 4; /* 4 / PW workgroups share a CU's 160 KB */
    static constexpr bool ok = tile_t >= RB && lds_bytes <= lds_budget;
};
template <class T, int D, bool MH, int NT, int I = 0, int PW = 4> struct mm_split_pick {
    static constexpr int rb_hmc[12] = {8, 4, 8, 4, 2, 8, 4, 2, 2, 8, 4, 2}, want_hmc[12] = {48, 48, 32, 32, 48, 24, 24, 32, 24, 16, 16, 16};
    static constexpr int rb_mh[12] = {8, 8, 8, 4, 4, 4, 8, 2, 2, 2, 4, 2}, want_mh[12] = {48, 32, 24, 48, 32, 24, 16, 48, 32, 24, 16, 16};
    static constexpr int rb = MH ? rb_mh[I] : rb_hmc[I], want = MH ? want_mh[I] : want_hmc[I];
    using Try = mm_split_try<T, D, rb, want, NT, PW>;
    using type = typename mm_cond<Try::ok, Try, typename mm_split_pick<T, D, MH, NT, I + 1, PW>::type>::type;
};
template <class T, int D, bool MH, int NT, int PW> struct mm_split_pick<T, D, MH, NT, 12, PW> {
    using type = mm_split_try<T, D, 2, 2 * (int)(sizeof(T) / 4) * D, NT, PW>; /* RB = TILE_T = 2: always fits up to dim 8 */
};
/* RBF != 0: a ring half of exactly RBF transitions (several noise waves per pair want a batch their pairs divide), the
 * largest tile that fits beside it */
template <class T, int D, bool MH, int RBF, int NT, int I = 0, int PW = 4> struct mm_split_pick_rb {
    static constexpr int want[6] = {48, 40, 32, 24, 16, 8};
    using Try = mm_split_try<T, D, RBF, want[I], NT, PW>;
    using type = typename mm_cond<Try::ok, Try, typename mm_split_pick_rb<T, D, MH, RBF, NT, I + 1, PW>::type>::type;
};
template <class T, int D, bool MH, int RBF, int NT, int PW> struct mm_split_pick_rb<T, D, MH, RBF, NT, 6, PW> {
    using type = mm_split_try<T, D, RBF, RBF * (int)(sizeof(T) / 4) * D, NT, PW>;
};
template <class T, int D, bool MH, int RBF = 0, int NT = MM_SPLIT_NTILE(MH), int PW = 4>
struct mm_split_plan : mm_cond<RBF == 0, typename mm_split_pick<T, D, MH, NT, 0, PW>::type, typename mm_split_pick_rb<T, D, MH, (RBF ? RBF : 2), NT, 0, PW>::type>::type {
    using Base = typename mm_cond<RBF == 0, typename mm_split_pick<T, D, MH, NT, 0, PW>::type, typename mm_split_pick_rb<T, D, MH, (RBF ? RBF : 2), NT, 0, PW>::type>::type;
    static constexpr int rb = (int)(Base::ring_bytes / 2 / Base::row_bytes);
    static_assert(Base::tile_t >= rb && Base::lds_bytes <= Base::lds_budget, "LDS plan of the split kernel");
};
/* HMC with the tiles written out by the least-loaded noise wave: where two tiles and a ring half of 8 transitions fit the
 * 160 KB (f32 up to D = 3).  Config 3 in tools/split_probe.hip (tools/experiments/split_hmc_pflush.sh, three rounds):
 * 0.1877 / 0.1850 / 0.1821 ms with the transition wave flushing its own tile, 0.1839 / 0.1814 / 0.1768 so -- a wave's store
 * instructions are issue-bound (~100 cycles each) and the transition wave is the kernel's critical path; with the plan's own
 * ring half (4) or the middle noise wave flushing it is slower (0.188-0.193).  Same bits. */
template <class T, int D> struct mm_split_hmc_pf {
    using Try = typename mm_split_pick_rb<T, D, false, 8, 2>::type;
    static constexpr bool ok = sizeof(T) == 4 && Try::ok && Try::tile_t >= 8 && Try::lds_bytes <= 160 * 1024;
};

#ifndef MM_SPLIT_MH_UFILT
#define MM_SPLIT_MH_UFILT 1 /* 0: ln u from the table in the noise waves (the form before round 5; A/B in tools/split_probe.hip) */
#endif
/* share of the noise the transition wave draws itself (pairs per batch): MH two of the four pairs of a batch of 8 */
#ifndef MM_SPLIT_MH_QP
#define MM_SPLIT_MH_QP 2 /* config 2: 0.312 / 0.298 / 0.275 ms with 0 / 1 / 2 pairs */
#endif
/* noise waves per pair of the f32 instances (mm_inst.inc): config 3 HMC 0.226 / 0.213 / 0.208 ms with 1 / 2 / 3 in
 * tools/split_probe.hip; in the library (tools/time_cfg23.py, one box, back to back): HMC 0.215 -> 0.201-0.204 ms,
 * config 2 MH 0.288 -> 0.253 ms with 3 */
#ifndef MM_SPLIT_NOISE_WAVES
#define MM_SPLIT_NOISE_WAVES 3
#endif
template <class T, int D, int NN = 1> struct mm_split_mh_qp {
    static constexpr int rb = mm_split_plan<T, D, true>::rb;
    static constexpr int one = rb >= 8 ? MM_SPLIT_MH_QP : (rb >= 4 && MM_SPLIT_MH_QP ? 1 : 0);
    /* with three noise waves, in the library (tools/time_cfg23.py, config 2): 0 / 1 / 2 pairs 0.2514 / 0.2527 / 0.2503 ms */
    /* round 5, f32 with the table-free accept filter (MM_SPLIT_MH_UFILT): the noise lost the logarithm, the transition gained the
     * band test, so the transition wave takes ONE pair of the batch instead of two -- 0 / 1 / 2 pairs: 0.2046 / 0.2040 / 0.2304 ms
     * against 0.2072 for the table form at 2 (tools/experiments/split_mh_ufilt.sh, profiles/r5d_split_mh_ufilt*.log) */
    /* the sampler's paired stream (f32, D <= 2: one Philox block per two transitions, mm_rng.h): the noise is now the light
     * role and keeps all of it -- 0 / 1 / 2 pairs: 0.1730 / 0.1927 / 0.2230 ms (profiles/r5i_split_mh_paired_layouts.log) */
    static constexpr int value = (sizeof(T) == 4 && D <= 2 && MM_SPLIT_MH_UFILT) ? 0 : ((sizeof(T) == 4 && MM_SPLIT_MH_UFILT && one > 1) ? 1 : one);
};

/* Role timing for tools/split_probe.hip (-DMM_SPLIT_PROFILE): s_memtime ticks each role spends at the batch barrier
 * and in total, summed over waves into mm_split_prof[role][0 / 1]; compiled out of the product. */
#ifdef MM_SPLIT_PROFILE
__device__ unsigned long long mm_split_prof[2][3]; /* [role][barrier wait / total / the transition wave's own draws] */
#define MM_SPLIT_SYNC()                                                                                           \
    do {                                                                                                          \
        const unsigned long long _t0 = __builtin_amdgcn_s_memtime();                                              \
        __syncthreads();                                                                                          \
        prof_wait += __builtin_amdgcn_s_memtime() - _t0;                                                          \
    } while (0)
#define MM_SPLIT_PROF_BEGIN() unsigned long long prof_wait = 0, prof_draw = 0; const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime()
#define MM_SPLIT_PROF_DRAW(stmt)                                                                                  \
    do {                                                                                                          \
        const unsigned long long _d0 = __builtin_amdgcn_s_memtime();                                              \
        stmt;                                                                                                     \
        prof_draw += __builtin_amdgcn_s_memtime() - _d0;                                                          \
    } while (0)
#define MM_SPLIT_PROF_END(role)                                                                                   \
    do {                                                                                                          \
        if ((threadIdx.x & 63) == 0) {                                                                            \
            atomicAdd(&mm_split_prof[role][0], prof_wait);                                                        \
            atomicAdd(&mm_split_prof[role][1], __builtin_amdgcn_s_memtime() - prof_t0);                           \
            atomicAdd(&mm_split_prof[role][2], prof_draw);                                                        \
        }                                                                                                         \
    } while (0)
#else
#define MM_SPLIT_SYNC() __syncthreads()
#define MM_SPLIT_PROF_BEGIN() ((void)0)
#define MM_SPLIT_PROF_END(role) ((void)0)
#define MM_SPLIT_PROF_DRAW(stmt) stmt
#endif

/* NN noise waves per pair (1: the workgroup above; 2, 3: 768 / 1024 threads, three / four waves per SIMD -- the noise
 * pairs of a batch are dealt to the noise waves in turn, the last noise wave -- dealt the fewest pairs -- writes the tiles out): more waves to cover each
 * other's LDS and issue latencies where the step is short (MH).  RBF: ring half (0 = the plan's choice). */
/* the kernel proper is a device function so that a run-time compiled translation unit (user targets, csrc/mm_rtc.hip) can
 * wrap it in an extern "C" kernel of its own, like mm_run_kernel_body */
template <class T, class Tgt, int SAMPLER, int LCT = 0, int QP = 0, int NN = 1, int RBF = 0, int NT = MM_SPLIT_NTILE(SAMPLER == MM_SAMPLER_MH), int PW = 4>
__device__ __forceinline__ void mm_run_split_body(const mm_run_args<T> &a)
{
    constexpr int D = Tgt::dim;
    static_assert(PW == 4 || PW == 2 || PW == 1, "wave pairs per workgroup");
    using Plan = mm_split_plan<T, D, SAMPLER == MM_SAMPLER_MH, RBF, NT, PW>;
    using Tile = typename Plan::Tile;
    constexpr int TILE_T = Tile::tile_t, STRIDE = Tile::stride, NW = Plan::nw, RB = Plan::rb, EPL = Plan::epl;
    constexpr int QN = 2 * QP; /* transitions per batch whose noise the transition wave draws itself */
    constexpr bool PFLUSH = Plan::ntile == 2; /* the noise wave writes the tiles out */
    /* MH in f32: the ring carries the accept uniform u itself, not ln u, and the transition decides `ratio > ln u` through
     * the table-free filter (mm_rng.h: mm_ratio_exceeds_lnu_f32) -- the noise loses the logarithm's 16-byte LDS gather and
     * ~12 instructions per transition, the transition gains ~5; the same decisions bit for bit.  HMC keeps ln u in the ring:
     * its transition wave is the kernel's critical path and the noise waves have slack. */
    constexpr bool UFILT = SAMPLER == MM_SAMPLER_MH && sizeof(T) == 4 && MM_SPLIT_MH_UFILT;
    /* MH, f32, D <= 2: the sampler's paired stream (mm_rng.h) -- ONE Philox block per pair of transitions; the ring carries
     * u_hi (the upper end of the accept uniform's interval), the transition evaluates the uniform's low byte (a second block)
     * only when the decision needs it (mm_accept_mhp_f32) */
    constexpr bool MHP = mm_mh_paired<T, D, SAMPLER>::value;
    static_assert(!MHP || UFILT || !MM_SPLIT_MH_UFILT, "paired MH stream");
    static_assert(QN <= RB, "QP");
    typedef T mm_vrow __attribute__((ext_vector_type(EPL)));
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_lds_raw[];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    /* the workgroup's waves in groups of PW: group `tw` holds the transition waves, the NN others the noise waves.  PW = 4:
     * tw = 0 (a group is one wave per SIMD).  PW < 4: a group covers PW of the four SIMDs, and tw rotates with the workgroup
     * index so that the 4 / PW workgroups of a CU put their transition waves on different SIMDs */
    const int group = wave / PW;
    const int tw = PW == 4 ? 0 : (int)(blockIdx.x % (unsigned int)(4 / PW)) % (1 + NN);
    const int pair = wave % PW;
    const bool noise_wave = group != tw;
    const int noise_rank = (group - tw - 1 + (1 + NN)) % (1 + NN); /* which of the pair's NN noise waves */
    T *const tiles = reinterpret_cast<T *>(mm_lds_raw + Plan::table_bytes + (size_t)pair * Plan::ntile * Plan::tile_bytes);
    constexpr size_t TILE_ELEMS = Plan::tile_bytes / sizeof(T);
    T *const ring = reinterpret_cast<T *>(mm_lds_raw + Plan::table_bytes + PW * Plan::ntile * Plan::tile_bytes + (size_t)pair * Plan::ring_bytes);
    const unsigned long long c = (unsigned long long)blockIdx.x * (64ull * PW) + (unsigned long long)(pair * 64 + lane);
    const unsigned long long wave_c0 = c - lane;
    const unsigned long long chain = a.chain_offset + c;
    const unsigned int n_silent = a.n_discard + (a.out ? 0u : a.n_collect);
    const unsigned int n_loud = a.out ? a.n_collect : 0u;
    /* both roles walk the same schedule: silent transitions in batches of RB (the last one short), then the collected
     * ones in batches of RB (the last one short), one barrier before each batch and one after the last; a tile is
     * complete after TILE_T collected transitions, or with the last one */
    using Tab = typename mm_cond<sizeof(T) == 4, mm_icdf_lds, mm_icdf_global>::type;
    Tab tab;
    if constexpr (sizeof(T) == 4) {
        mm_icdf_lds_fill(reinterpret_cast<float *>(mm_lds_raw), threadIdx.x, 64u * PW * (1u + NN));
        mm_icdf_lds_check(mm_lds_raw);
        tab.tab = mm_icdf_lds_at0();
    }
#ifdef MM_SPLIT_PRIO_Q /* experiment hook of tools/split_probe.hip */
    if (!noise_wave)
        __builtin_amdgcn_s_setprio(MM_SPLIT_PRIO_Q);
#endif
#ifdef MM_SPLIT_PRIO_N /* experiment hook: 1 = younger noise waves first (against the arbiter's oldest-first), 2 = older first */
    if (noise_wave) {
        const int pr = MM_SPLIT_PRIO_N == 1 ? noise_rank : MM_SPLIT_PRIO_N == 2 ? 2 - noise_rank : MM_SPLIT_PRIO_N == 3 ? 1 : MM_SPLIT_PRIO_N == 4 ? noise_rank + 1 : (noise_rank == 2 ? 2 : 0);
        if (pr <= 0)
            __builtin_amdgcn_s_setprio(0);
        else if (pr == 1)
            __builtin_amdgcn_s_setprio(1);
        else if (pr == 2)
            __builtin_amdgcn_s_setprio(2);
        else
            __builtin_amdgcn_s_setprio(3);
    }
#endif

    if (noise_wave) {
        __syncthreads(); /* the table is complete */
        MM_SPLIT_PROF_BEGIN();
        unsigned int it = a.iter0;
        int half = 0;
        /* noise of iterations [it + QN, it + nb) of a batch into rows QN.. of the current ring half */
        /* PAIRS noise pairs are drawn back to back before any of them is stored: their Philox rounds and table reads
         * overlap (a lone pair waits out every LDS table read: the noise wave has no other work to cover it) */
#ifndef MM_SPLIT_FLUSH_RANK
#define MM_SPLIT_FLUSH_RANK (NN - 1) /* the noise wave that is dealt the fewest pairs of a batch writes the tiles out */
#endif
#ifndef MM_SPLIT_FILL_PAIRS
#define MM_SPLIT_FILL_PAIRS 1 /* 2 and 4 measured: no difference (config 2: 0.273 / 0.280 / 0.272 ms) */
#endif
        constexpr int PAIRS = ((RB - QN) % (2 * MM_SPLIT_FILL_PAIRS * NN) == 0 && RB > QN) ? MM_SPLIT_FILL_PAIRS : 1;
        static_assert((RB - QN) % 2 == 0, "whole noise pairs");
        auto fill = [&](unsigned int nb) __attribute__((always_inline)) {
            T *dst = ring + ((size_t)half * RB + QN + 2 * PAIRS * noise_rank) * 64 * NW + (size_t)lane * NW;
            for (unsigned int j = QN + 2u * PAIRS * (unsigned int)noise_rank; j < nb; j += 2 * PAIRS * NN) {
                T rows[2 * PAIRS][NW];
                MM_UNROLL
                for (int q = 0; q < PAIRS; ++q)
                    mm_draw_noise_pair<D, Tab, !UFILT, MHP>(a.seed, chain, it + j + 2 * q, rows[2 * q], &rows[2 * q][D], rows[2 * q + 1], &rows[2 * q + 1][D], tab);
                MM_UNROLL
                for (int r = 0; r < 2 * PAIRS; ++r) {
                    MM_UNROLL
                    for (int k = D + 1; k < NW; ++k)
                        rows[r][k] = T(0);
                    MM_UNROLL
                    for (int k = 0; k < NW; k += EPL) {
                        mm_vrow v;
                        MM_UNROLL
                        for (int e = 0; e < EPL; ++e)
                            v[e] = rows[r][k + e];
                        /* rows past nb (an odd or short batch) land inside the half: RB - QN is a multiple of 2 PAIRS */
                        *reinterpret_cast<mm_vrow *>(dst + (size_t)r * 64 * NW + k) = v;
                    }
                }
                dst += (size_t)2 * PAIRS * NN * 64 * NW;
            }
            it += nb;
            half ^= 1;
        };
        for (unsigned int done = 0; done < n_silent; done += RB) {
            fill(min((unsigned int)RB, n_silent - done));
            MM_SPLIT_SYNC();
        }
        /* collected transitions: after the barrier that hands batch b over, the partner has finished batch b - 1; a tile
         * that batch completed is written out while the partner stages into the other one */
        unsigned int tcol = 0, rows_out = 0, pend = 0; /* pend: rows of the completed tile waiting to be flushed */
        int tb = 0, pend_tb = 0;
        for (unsigned int done = 0; done < n_loud; done += RB) {
            const unsigned int nb = min((unsigned int)RB, n_loud - done);
            fill(nb);
            MM_SPLIT_SYNC();
            if (PFLUSH && pend && noise_rank == MM_SPLIT_FLUSH_RANK) {
                mm_flush_tile_raw<T, D, Tile>(a.out, a.n_total, a.n_chains, tiles + (size_t)pend_tb * TILE_ELEMS, lane, wave_c0,
                                              (unsigned long long)a.out_t0 + rows_out, pend);
                rows_out += pend;
                pend = 0;
            }
            tcol += nb; /* what the partner is staging now */
            if (tcol == (unsigned int)TILE_T || done + nb >= n_loud) {
                pend = tcol;
                pend_tb = tb;
                tb ^= 1;
                tcol = 0;
            }
        }
        MM_SPLIT_SYNC(); /* the partner has finished its last batch */
        if (PFLUSH && pend && noise_rank == MM_SPLIT_FLUSH_RANK)
            mm_flush_tile_raw<T, D, Tile>(a.out, a.n_total, a.n_chains, tiles + (size_t)pend_tb * TILE_ELEMS, lane, wave_c0,
                                          (unsigned long long)a.out_t0 + rows_out, pend);
        MM_SPLIT_PROF_END(0);
        return;
    }

    /* ---- transition wave ---- */
    const bool active = c < a.n_chains;
    T x[D], g[D], lp;
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        x[i] = active ? a.state[c * D + i] : T(0);
    if (SAMPLER == MM_SAMPLER_HMC)
        lp = Tgt::logp_grad(a.P, x, g);
    else
        lp = Tgt::logp(a.P, x);
    __syncthreads(); /* the table is complete */
    MM_SPLIT_PROF_BEGIN();

    unsigned int n_acc32 = 0;
    unsigned int it = a.iter0;
    int half = 0;
    auto load_row = [&](const T *src, T *z, T *ln_u) __attribute__((always_inline)) {
        T row[NW];
        MM_UNROLL
        for (int k = 0; k < NW; k += EPL) {
            const mm_vrow v = *reinterpret_cast<const mm_vrow *>(src + k);
            MM_UNROLL
            for (int e = 0; e < EPL; ++e)
                row[k + e] = v[e];
        }
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            z[k] = row[k];
        *ln_u = row[D];
    };
    auto transition = [&](T *z, T ln_u, unsigned int iter_t) __attribute__((always_inline)) {
        if (SAMPLER == MM_SAMPLER_HMC)
            (void)mm_hmc_step_noise<T, Tgt, LCT>(a.P, a.scale, a.n_leapfrog, x, &lp, g, z, ln_u, mm_no_hook(), &n_acc32);
        else if constexpr (UFILT && MHP)
            (void)mm_mh_step_accept<T, Tgt>(a.P, a.scale, x, &lp, z, mm_accept_mhp_f32<Tab>{ln_u, a.seed, chain, iter_t, tab}, &n_acc32); /* `ln_u` holds u_hi */
        else if constexpr (UFILT)
            (void)mm_mh_step_accept<T, Tgt>(a.P, a.scale, x, &lp, z, mm_accept_u_f32<Tab>{ln_u, tab}, &n_acc32); /* `ln_u` holds u */
        else
            (void)mm_mh_step_noise<T, Tgt>(a.P, a.scale, x, &lp, z, ln_u, &n_acc32);
        /* accepts are counted inside the step, one add under the accept mask; lanes past n_chains count too (their counters
         * are dropped below) and the wave's total is summed once at the end -- a select, a ballot + s_bcnt1 and a 64-bit
         * scalar add per transition were 6 of the transition wave's ~160 dependent-issue slots */
    };
    /* one batch: nb transitions; `stage` (null: silent) receives the states.  The first QN take noise drawn here, in
     * registers; the others the partner's rows, each requested one transition ahead */
    T zq[QN > 0 ? QN : 1][D], lnq[QN > 0 ? QN : 1]; /* own noise of the NEXT batch, drawn after this batch's barrier */
    auto draw_own = [&](unsigned int it_batch) __attribute__((always_inline)) {
        MM_UNROLL
        for (int q = 0; q < QN; q += 2)
            mm_draw_noise_pair<D, Tab, !UFILT, MHP>(a.seed, chain, it_batch + q, zq[q], &lnq[q], zq[q + 1], &lnq[q + 1], tab);
    };
    /* Collected states are staged two transitions at a time where that makes wider LDS writes (MM_SPLIT_STAGE_PAIRS): the
     * state after the first transition of a pair waits in registers and both leave in 2 D sizeof(T) bytes cut into
     * 16- or 8-byte pieces.  At D = 2 (f32) that is one ds_write_b128 per two transitions, conflict free at the tile's
     * row stride, where the per-transition ds_write_b64 met a 2-way bank conflict that delayed the ring reads queued
     * behind it.  Batches and tiles hold an even number of transitions, so a pair never straddles a tile. */
#ifndef MM_SPLIT_STAGE_PAIRS
#define MM_SPLIT_STAGE_PAIRS 1
#endif
    constexpr int PAIR_BYTES = 2 * D * (int)sizeof(T);
    constexpr int PIECE_BYTES = (PAIR_BYTES % 16 == 0) ? 16 : ((PAIR_BYTES % 8 == 0) ? 8 : 4);
    /* only where a pair is whole 16-byte pieces and a single state is not: D = 2 in f32 (config 2: 0.250 -> 0.236 ms), D = 1 in
     * f64; at D = 3 (three 8-byte pieces per pair) it measured 1 % slower than ds_write2_b32 + ds_write_b32 per transition */
    constexpr bool STAGE_PAIRS = MM_SPLIT_STAGE_PAIRS && (D * (int)sizeof(T)) % 16 != 0 && PIECE_BYTES == 16 &&
                                 (STRIDE * (int)sizeof(T)) % PIECE_BYTES == 0 && RB % 2 == 0 && QN % 2 == 0;
    constexpr int PE = PIECE_BYTES / (int)sizeof(T); /* elements per piece */
    typedef T mm_piece __attribute__((ext_vector_type(PE > 1 ? PE : 2)));
    auto put_one = [&](T *dst) __attribute__((always_inline)) {
        MM_UNROLL
        for (int k = 0; k < D; ++k)
            dst[k] = x[k];
    };
    auto put_pair = [&](T *dst, const T *first) __attribute__((always_inline)) {
        T both[2 * D];
        MM_UNROLL
        for (int k = 0; k < D; ++k) {
            both[k] = first[k];
            both[D + k] = x[k];
        }
        MM_UNROLL
        for (int k = 0; k < 2 * D; k += PE) {
            mm_piece v;
            MM_UNROLL
            for (int e = 0; e < PE; ++e)
                v[e] = both[k + e];
            *reinterpret_cast<mm_piece *>(dst + k) = v;
        }
    };
    auto batch = [&](unsigned int nb, T *stage) __attribute__((always_inline)) {
        const T *src = ring + (size_t)half * RB * 64 * NW + (size_t)lane * NW;
        T zc[D], lnc;
        if (QN < RB)
            load_row(src + (size_t)QN * 64 * NW, zc, &lnc);
        if constexpr (STAGE_PAIRS) {
            T held[D];
            MM_UNROLL
            for (int q = 0; q < QN; q += 2) {
                if ((unsigned int)q < nb) {
                    transition(zq[q], lnq[q], it + q);
                    if ((unsigned int)q + 1u < nb) {
                        MM_UNROLL
                        for (int k = 0; k < D; ++k)
                            held[k] = x[k];
                        transition(zq[q + 1], lnq[q + 1], it + q + 1);
                        if (stage)
                            put_pair(stage + q * D, held);
                    } else if (stage) {
                        put_one(stage + q * D);
                    }
                }
            }
            unsigned int j = QN;
            for (; j + 1u < nb; j += 2u) {
                T zn[D], lnn; /* rows past nb hold stale data: requested, never used */
                load_row(src + (size_t)(j + 1u) * 64 * NW, zn, &lnn);
                transition(zc, lnc, it + j);
                MM_UNROLL
                for (int k = 0; k < D; ++k)
                    held[k] = x[k];
                load_row(src + (size_t)min(j + 2u, (unsigned int)RB - 1u) * 64 * NW, zc, &lnc);
                transition(zn, lnn, it + j + 1u);
                if (stage)
                    put_pair(stage + j * D, held);
            }
            if (j < nb) {
                transition(zc, lnc, it + j);
                if (stage)
                    put_one(stage + j * D);
            }
        } else {
            MM_UNROLL
            for (int q = 0; q < QN; ++q) {
                if ((unsigned int)q < nb) {
                    transition(zq[q], lnq[q], it + q);
                    if (stage)
                        put_one(stage + q * D);
                }
            }
            for (unsigned int j = QN; j < nb; ++j) {
                T zn[D], lnn; /* rows past nb hold stale data: requested, never used */
                load_row(src + (size_t)min(j + 1u, (unsigned int)RB - 1u) * 64 * NW, zn, &lnn);
                transition(zc, lnc, it + j);
                if (stage)
                    put_one(stage + j * D);
                MM_UNROLL
                for (int k = 0; k < D; ++k)
                    zc[k] = zn[k];
                lnc = lnn;
            }
        }
        it += nb;
        half ^= 1;
    };

    if (QN)
        draw_own(it);
    for (unsigned int done = 0; done < n_silent; done += RB) {
        MM_SPLIT_SYNC(); /* ring half `half` is full; the other one is free again */
        batch(min((unsigned int)RB, n_silent - done), nullptr);
        if (QN)
            MM_SPLIT_PROF_DRAW(draw_own(it)); /* for the next batch (a batch past the end of the run: drawn, never used) */
    }
    unsigned int tcol = 0, rows_out = 0;
    int tb = 0;
    for (unsigned int done = 0; done < n_loud; done += RB) {
        const unsigned int nb = min((unsigned int)RB, n_loud - done);
        MM_SPLIT_SYNC(); /* ... and (PFLUSH) the tile this batch may be the first to write into has been flushed */
        batch(nb, tiles + (size_t)tb * TILE_ELEMS + lane * STRIDE + tcol * D);
        if (QN)
            MM_SPLIT_PROF_DRAW(draw_own(it));
        tcol += nb;
        if (PFLUSH) {
            if (tcol == (unsigned int)TILE_T) {
                tb ^= 1;
                tcol = 0;
            }
        } else if (tcol == (unsigned int)TILE_T || done + nb >= n_loud) {
            mm_flush_tile_raw<T, D, Tile>(a.out, a.n_total, a.n_chains, tiles, lane, wave_c0,
                                          (unsigned long long)a.out_t0 + rows_out, tcol);
            rows_out += tcol;
            tcol = 0;
        }
    }
    MM_SPLIT_SYNC(); /* hands the last tile over */
    MM_SPLIT_PROF_END(1);

    if (active) {
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = x[i];
        if (a.accept)
            a.accept[c] += (unsigned long long)n_acc32;
    }
    if (a.accept_total) {
        unsigned long long wave_acc = active ? (unsigned long long)n_acc32 : 0ull;
        MM_UNROLL
        for (int off = 32; off > 0; off >>= 1)
            wave_acc += __shfl_xor(wave_acc, off);
        if (lane == 0 && wave_acc)
            atomicAdd(a.accept_total, wave_acc);
    }
}

template <class T, class Tgt, int SAMPLER, int LCT = 0, int QP = 0, int NN = 1, int RBF = 0, int NT = MM_SPLIT_NTILE(SAMPLER == MM_SAMPLER_MH)>
__global__ __launch_bounds__(256 * (1 + NN)) void mm_run_split_kernel(const mm_run_args<T> a)
{
    mm_run_split_body<T, Tgt, SAMPLER, LCT, QP, NN, RBF, NT>(a);
}
/* workgroup = PW pairs (PW = 1, 2): 64 PW (1 + NN) threads, 1 + NN waves per SIMD as before (4 / PW workgroups per CU) */
template <class T, class Tgt, int SAMPLER, int PW, int LCT = 0, int QP = 0, int NN = 3, int RBF = 0, int NT = MM_SPLIT_NTILE(SAMPLER == MM_SAMPLER_MH)>
__global__ __launch_bounds__(64 * PW * (1 + NN)) __attribute__((amdgpu_waves_per_eu(1 + NN, 1 + NN))) void mm_run_splitw_kernel(const mm_run_args<T> a)
{
    mm_run_split_body<T, Tgt, SAMPLER, LCT, QP, NN, RBF, NT, PW>(a);
}

#if !defined(__HIPCC_RTC__)
#include <atomic>
template <class T, class Tgt, int SAMPLER, int LCT = 0, int QP = 0, int NN = 1, int RBF = 0, int NT = MM_SPLIT_NTILE(SAMPLER == MM_SAMPLER_MH)>
hipError_t mm_launch_run_split(const mm_run_args<T> &a, hipStream_t stream)
{
    using Plan = mm_split_plan<T, Tgt::dim, SAMPLER == MM_SAMPLER_MH, RBF, NT>;
    static std::atomic<unsigned long long> attr_set{0}; /* > 64 KB of dynamic LDS has to be allowed once per kernel and device */
    auto kern = mm_run_split_kernel<T, Tgt, SAMPLER, LCT, QP, NN, RBF, NT>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 64 || !((attr_set >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)Plan::lds_bytes);
        if (e != hipSuccess)
            return e;
        if (dev < 64)
            attr_set |= 1ull << dev;
    }
    const unsigned int grid = (unsigned int)((a.n_chains + 255ull) / 256ull);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * (1 + NN)), Plan::lds_bytes, stream, a);
    return hipGetLastError();
}
template <class T, class Tgt, int SAMPLER, int PW, int LCT = 0, int QP = 0, int NN = 3, int RBF = 0, int NT = MM_SPLIT_NTILE(SAMPLER == MM_SAMPLER_MH)>
hipError_t mm_launch_run_splitw(const mm_run_args<T> &a, hipStream_t stream)
{
    using Plan = mm_split_plan<T, Tgt::dim, SAMPLER == MM_SAMPLER_MH, RBF, NT, PW>;
    static std::atomic<unsigned long long> attr_set{0};
    auto kern = mm_run_splitw_kernel<T, Tgt, SAMPLER, PW, LCT, QP, NN, RBF, NT>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (Plan::lds_bytes > 64 * 1024 && (dev >= 64 || !((attr_set >> dev) & 1ull))) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)Plan::lds_bytes);
        if (e != hipSuccess)
            return e;
        if (dev < 64)
            attr_set |= 1ull << dev;
    }
    const unsigned int grid = (unsigned int)((a.n_chains + 64ull * PW - 1ull) / (64ull * PW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * PW * (1 + NN)), Plan::lds_bytes, stream, a);
    return hipGetLastError();
}
#endif /* !__HIPCC_RTC__ */

#endif /* MM_SPLIT_KERNELS_H */
