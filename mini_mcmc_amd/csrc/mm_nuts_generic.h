/*
 * mm_nuts_generic.h -- one NUTS transition of one chain at ANY dimension (run-time D), host + device, one definition.
 *
 * The reference's NUTS takes any target that implements GradientTarget at any dimension (nuts.rs:123-129,
 * distributions.rs:81-87); the fixed-D kernels (mm_nuts.h, mm_nuts_kernels.h, mm_nuts_dims.h) keep a transition's
 * vectors in registers and exist for a list of dimensions.  Everything else runs here: the thirteen vectors of a
 * transition and the pending-subtree stack live in a STORE (mm_generic.h: mm_gstore -- plain arrays on the host, a
 * lane-interleaved HBM area on the device) and the arithmetic walks them element by element.
 *
 * This is mm_nuts_tree (mm_nuts.h) operation for operation -- same leapfrog, same sequential dot products, same
 * draws at the same points (mm_rng.h, NUTS schedule), same merges in the same order -- so at a dimension both paths
 * support they give the same bits (tests/test_generic_dim.py), and the host build of this header
 * (oracle/engine_host.cpp, modes 4-6) is the device path's bit-exact twin at every dimension.
 *
 * Restates NUTSChain::step nuts.rs:550-691, build_tree :764-946, leapfrog :979-996, stop_criterion :963-977,
 * find_reasonable_epsilon :695-761, init_chain :528-545.
 */
#ifndef MM_NUTS_GENERIC_H
#define MM_NUTS_GENERIC_H

#include "mm_generic.h"
#include "mm_nuts.h"

/* vectors of a chain in the store; stack entry e keeps (first-leaf position, first-leaf momentum, proposal) */
enum {
    MM_NV_X = 0, MM_NV_XM, MM_NV_XP, MM_NV_PM, MM_NV_PP, MM_NV_GM, MM_NV_GP, MM_NV_CX, MM_NV_CP, MM_NV_CG, MM_NV_SFX,
    MM_NV_SFP, MM_NV_SPR, MM_NV_T0, MM_NV_STACK0
};
MM_HD int mm_gen_nuts_vectors(int max_depth) { return MM_NV_STACK0 + 3 * (max_depth > 1 ? max_depth : 1); }

/* z[0..D) of (chain, iteration) in the element type's schedule (mm_nuts_momentum at run-time D) */
template <class F> MM_HD void mm_gen_normals(uint64_t seed, uint64_t chain, uint32_t iter, int D, float, F &&f)
{
    const mm_icdf_global tab;
    for (int b = 0; 4 * b < D; ++b) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        for (int k = 0; k < 4 && 4 * b + k < D; ++k)
            f(4 * b + k, mm_icdf_f32(blk.w[k], tab));
    }
}
template <class F> MM_HD void mm_gen_normals(uint64_t seed, uint64_t chain, uint32_t iter, int D, double, F &&f)
{
    for (int b = 0; 2 * b < D; ++b) {
        const mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        double z0, z1;
        mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
        f(2 * b, z0);
        if (2 * b + 1 < D)
            f(2 * b + 1, z1);
    }
}

template <class TT, class S> MM_HD TT mm_gen_dot(const S &s, int va, int vb)
{
    TT acc = 0;
    for (int i = 0; i < s.dim; ++i)
        acc = mm_fma(s.ld(va, i), s.ld(vb, i), acc);
    return acc;
}
template <class S> MM_HD void mm_gen_copy(const S &s, int dst, int src)
{
    for (int i = 0; i < s.dim; ++i)
        s.st(dst, i, s.ld(src, i));
}

/* nuts.rs:979-996 on the store's vectors (vx, vp, vg), in place */
template <class TT, class S> MM_HD TT mm_gen_nuts_leapfrog(int kind, const mm_tparams<TT> &P, const S &s, int vx, int vp, int vg, TT eps)
{
    const TT h = eps * TT(0.5);
    for (int i = 0; i < s.dim; ++i) {
        const TT p = mm_fma(h, s.ld(vg, i), s.ld(vp, i));
        s.st(vp, i, p);
        s.st(vx, i, mm_fma(eps, p, s.ld(vx, i)));
    }
    const TT lp = mm_gen_logp_grad<TT, S>(kind, P, s, vx, vg);
    for (int i = 0; i < s.dim; ++i)
        s.st(vp, i, mm_fma(h, s.ld(vg, i), s.ld(vp, i)));
    return lp;
}

/* nuts.rs:963-977: (x+ - x-).p- >= 0 and (x+ - x-).p+ >= 0 */
template <class TT, class S> MM_HD bool mm_gen_stop_criterion(const S &s, int xm, int xp, int pm, int pp)
{
    TT a = 0;
    for (int i = 0; i < s.dim; ++i)
        a = mm_fma(s.ld(xp, i) - s.ld(xm, i), s.ld(pm, i), a);
    if (!(a >= TT(0)))
        return false;
    TT b = 0;
    for (int i = 0; i < s.dim; ++i)
        b = mm_fma(s.ld(xp, i) - s.ld(xm, i), s.ld(pp, i), b);
    return b >= TT(0);
}

/* nuts.rs:695-761; position in MM_NV_X, momentum in MM_NV_PM; uses T0 (gradient at the start) and CX / CP / CG */
template <class TT, class ST, class S> MM_HD ST mm_gen_find_reasonable_epsilon(int kind, const mm_tparams<TT> &P, const S &s)
{
    ST epsilon = 1;
    const ST half = ST(0.5);
    const TT ulogp = mm_gen_logp_grad<TT, S>(kind, P, s, MM_NV_X, MM_NV_T0);
    const TT mom_sq = mm_gen_dot<TT>(s, MM_NV_PM, MM_NV_PM);
    auto leap = [&](ST e) -> TT {
        mm_gen_copy(s, MM_NV_CX, MM_NV_X);
        mm_gen_copy(s, MM_NV_CP, MM_NV_PM);
        mm_gen_copy(s, MM_NV_CG, MM_NV_T0);
        return mm_gen_nuts_leapfrog<TT, S>(kind, P, s, MM_NV_CX, MM_NV_CP, MM_NV_CG, (TT)e);
    };
    TT ulogp_p = leap(epsilon);
    bool grad_real = true; /* grad_prime of the FIRST leapfrog only (Q11) */
    for (int i = 0; i < s.dim; ++i)
        grad_real = grad_real && mm_is_real(s.ld(MM_NV_CG, i));
    ST k = 1;
    while (!mm_is_real(ulogp_p) && !grad_real) {
        k = k * half;
        ulogp_p = leap(epsilon * k);
    }
    epsilon = half * k * epsilon;
    ST lap = (ST)(double)(ulogp_p - ulogp - (mm_gen_dot<TT>(s, MM_NV_CP, MM_NV_CP) - mom_sq) * TT(0.5));
    const ST a = (lap > mm_logT(half)) ? ST(1) : ST(-1);
    const ST ln2 = mm_logT(ST(2));
    while (a * lap > -a * ln2) {
        epsilon = (a > ST(0)) ? epsilon * ST(2) : epsilon * half;
        ulogp_p = leap(epsilon);
        lap = (ST)(double)(ulogp_p - ulogp - (mm_gen_dot<TT>(s, MM_NV_CP, MM_NV_CP) - mom_sq) * TT(0.5));
    }
    return epsilon;
}

/* nuts.rs:528-545 (mm_nuts_init_chain): position in MM_NV_X */
template <class TT, class ST, class S>
MM_HD void mm_gen_nuts_init_chain(int kind, const mm_tparams<TT> &P, const S &s, mm_nuts_adapt<ST> *ad, ST eps_tol, uint64_t seed,
                                  uint64_t chain)
{
    mm_gen_normals(seed, chain, 0u, s.dim, TT(0), [&](int i, TT z) { s.st(MM_NV_PM, i, z); });
    ST d = ad->epsilon + ST(1);
    if ((d < 0 ? -d : d) <= eps_tol)
        ad->epsilon = mm_gen_find_reasonable_epsilon<TT, ST, S>(kind, P, s);
    ad->mu = mm_logT(ST(10) * ad->epsilon);
}

/* nuts.rs:550-691 (mm_nuts_step): the position in MM_NV_X is updated in place */
template <class TT, class ST, class S>
MM_HD mm_nuts_info mm_gen_nuts_step(int kind, const mm_tparams<TT> &P, const S &s, mm_nuts_adapt<ST> *ad, uint32_t m, uint32_t n_discard,
                                    ST target_accept_p, int max_depth, uint64_t seed, uint64_t chain)
{
    /* scalars of the pending-subtree stack */
    ST stk_alpha[MM_NUTS_JMAX];
    uint32_t stk_level[MM_NUTS_JMAX], stk_n[MM_NUTS_JMAX], stk_nalpha[MM_NUTS_JMAX];
    /* the auxiliary uniforms: draw k = a half of Philox block AUX + (k >> 1) (mm_rng.h) */
    uint32_t aux_k = 0, aux_have = 0xffffffffu;
    mm_u32x4 aux_blk;
    aux_blk.w[0] = aux_blk.w[1] = aux_blk.w[2] = aux_blk.w[3] = 0u;
    auto aux = [&]() -> double {
        const uint32_t k = aux_k++;
        if ((k >> 1) != aux_have) {
            aux_blk = mm_block(seed, chain, m, MM_AUX_BLOCK + (k >> 1));
            aux_have = k >> 1;
        }
        return (k & 1u) ? mm_u53(aux_blk.w[2], aux_blk.w[3]) : mm_u53(aux_blk.w[0], aux_blk.w[1]);
    };
    mm_nuts_info info;
    info.depth = 0;
    info.n_leapfrog = 0;

    /* ---- begin (nuts.rs:550-576) ---- */
    mm_gen_normals(seed, chain, m, s.dim, TT(0), [&](int i, TT z) { s.st(MM_NV_PM, i, z); });
    const TT ulogp = mm_gen_logp_grad<TT, S>(kind, P, s, MM_NV_X, MM_NV_GM);
    const ST joint = (ST)(double)(ulogp - mm_gen_dot<TT>(s, MM_NV_PM, MM_NV_PM) * TT(0.5));
    const ST exp1_obs = (ST)(-mm_log(aux()));
    const ST logu = joint - exp1_obs;
    for (int i = 0; i < s.dim; ++i) {
        const TT x = s.ld(MM_NV_X, i);
        s.st(MM_NV_XM, i, x);
        s.st(MM_NV_XP, i, x);
        s.st(MM_NV_PP, i, s.ld(MM_NV_PM, i));
        s.st(MM_NV_GP, i, s.ld(MM_NV_GM, i));
    }
    int j = 0;
    uint32_t n = 1, n_alpha = 0;
    bool sflag = true;
    ST alpha = 0;

    while (sflag) {
        /* ---- double_begin ---- */
        const ST u_run_1 = (ST)aux();
        const int v = (u_run_1 < ST(0.5)) ? 1 : -1;
        const bool neg = v == -1;
        mm_gen_copy(s, MM_NV_CX, neg ? MM_NV_XM : MM_NV_XP);
        mm_gen_copy(s, MM_NV_CP, neg ? MM_NV_PM : MM_NV_PP);
        mm_gen_copy(s, MM_NV_CG, neg ? MM_NV_GM : MM_NV_GP);
        const TT eps_signed = (TT)((ST)v * ad->epsilon);
        int sp = 0;
        uint32_t S_level = 0, S_n = 0, S_nalpha = 0;
        bool S_s = true;
        ST S_alpha = 0;

        for (bool done = false; !done;) {
            /* ---- one leaf (build_tree base case, nuts.rs:782-856) ---- */
            const TT lp = mm_gen_nuts_leapfrog<TT, S>(kind, P, s, MM_NV_CX, MM_NV_CP, MM_NV_CG, eps_signed);
            info.n_leapfrog += 1;
            const ST jointp = (ST)(double)(lp - mm_gen_dot<TT>(s, MM_NV_CP, MM_NV_CP) * TT(0.5));
            S_level = 0;
            S_n = (logu < jointp) ? 1u : 0u;
            S_s = (logu - ST(1000)) < jointp;
            S_alpha = mm_accept_stat<ST>(jointp - joint);
            S_nalpha = 1;
            mm_gen_copy(s, MM_NV_SFX, MM_NV_CX);
            mm_gen_copy(s, MM_NV_SFP, MM_NV_CP);
            mm_gen_copy(s, MM_NV_SPR, MM_NV_CX);
            /* ---- hand S up the implicit recursion (mm_nuts_tree::hand_up) ---- */
            for (;;) {
                if (S_level == (uint32_t)j) {
                    done = true;
                    break;
                }
                if (sp > 0 && stk_level[sp - 1] == S_level) {
                    /* S is the second child: merge with its sibling (nuts.rs:900-928) */
                    const int e = sp - 1;
                    const int vfx = MM_NV_STACK0 + 3 * e, vfp = vfx + 1, vpr = vfx + 2;
                    const uint32_t n1 = stk_n[e];
                    const double u = aux(); /* always f64 (nuts.rs:910) */
                    uint32_t den = n1 + S_n;
                    if (den < 1)
                        den = 1;
                    const bool take2 = u < ((double)S_n / (double)den);
                    if (!take2)
                        mm_gen_copy(s, MM_NV_SPR, vpr);
                    S_n += n1;
                    const bool crit = neg ? mm_gen_stop_criterion<TT>(s, MM_NV_CX, vfx, MM_NV_CP, vfp)
                                          : mm_gen_stop_criterion<TT>(s, vfx, MM_NV_CX, vfp, MM_NV_CP);
                    S_s = S_s && crit;
                    S_alpha = stk_alpha[e] + S_alpha;
                    S_nalpha += stk_nalpha[e];
                    mm_gen_copy(s, MM_NV_SFX, vfx);
                    mm_gen_copy(s, MM_NV_SFP, vfp);
                    S_level += 1;
                    sp -= 1;
                } else if (S_s) {
                    /* first child, still valid: wait for the sibling */
                    const int vfx = MM_NV_STACK0 + 3 * sp;
                    mm_gen_copy(s, vfx, MM_NV_SFX);
                    mm_gen_copy(s, vfx + 1, MM_NV_SFP);
                    mm_gen_copy(s, vfx + 2, MM_NV_SPR);
                    stk_alpha[sp] = S_alpha;
                    stk_level[sp] = S_level;
                    stk_n[sp] = S_n;
                    stk_nalpha[sp] = S_nalpha;
                    sp += 1;
                    break; /* next leaf */
                } else {
                    /* first child with s' = 0: the parent returns it as it is (nuts.rs:858 not taken) */
                    S_level += 1;
                }
            }
        }

        /* ---- double_end (nuts.rs:581-671) ---- */
        mm_gen_copy(s, neg ? MM_NV_XM : MM_NV_XP, MM_NV_CX);
        mm_gen_copy(s, neg ? MM_NV_PM : MM_NV_PP, MM_NV_CP);
        mm_gen_copy(s, neg ? MM_NV_GM : MM_NV_GP, MM_NV_CG);
        alpha = S_alpha; /* from the LAST doubling only (nuts.rs:614-615, 649-650) */
        n_alpha = S_nalpha;
        const ST tmp = mm_minT(ST(1), (ST)S_n / (ST)n);
        const ST u_run_2 = (ST)aux();
        if (S_s && (u_run_2 < tmp))
            mm_gen_copy(s, MM_NV_X, MM_NV_SPR);
        n += S_n;
        sflag = S_s && mm_gen_stop_criterion<TT>(s, MM_NV_XM, MM_NV_XP, MM_NV_PM, MM_NV_PP);
        j += 1;
        if (j >= max_depth)
            sflag = false; /* depth cap: not in the reference */
        info.depth = j;
    }

    /* ---- dual averaging (nuts.rs:676-690; mm_nuts_tree::finish) ---- */
    ST eta = ST(1) / (ST)(m + MM_NUTS_T0);
    ad->h_bar = (ST(1) - eta) * ad->h_bar + eta * (target_accept_p - alpha / (ST)n_alpha);
    if (m <= n_discard) {
        const ST _m = (ST)m;
        ad->epsilon = mm_expT(ad->mu - mm_sqrtT(_m) / ST(MM_NUTS_GAMMA) * ad->h_bar);
        eta = mm_expT(-ST(MM_NUTS_KAPPA) * mm_logT(_m));
        ad->epsilon_bar = mm_expT((ST(1) - eta) * mm_logT(ad->epsilon_bar) + eta * mm_logT(ad->epsilon));
    } else {
        ad->epsilon = ad->epsilon_bar;
    }
    return info;
}

#if defined(__HIPCC__)
/* ---------------------------------------------------------------- device side */
template <class TT, class ST> struct mm_gen_nuts_args {
    mm_tparams<TT> P;
    int kind, dim, max_depth;
    TT *state;                      /* [C, D] */
    mm_nuts_adapt<ST> *adapt;       /* [C] */
    TT *out;                        /* [C, n_total, D] or NULL */
    unsigned long long *n_leapfrog; /* [C] running totals, or NULL */
    unsigned int *depth_hist;       /* [MM_NUTS_JMAX + 1] or NULL */
    TT *store;                      /* [mm_gen_nuts_vectors(max_depth) * dim][c_pad] */
    unsigned long long n_chains, c_pad, seed, chain_offset, n_total;
    unsigned int m0, n_pre, n_rec, write_initial, out_t0, n_discard;
    ST target_accept_p, eps_tol;
};

/* init_chain for every chain (mm_nuts_init_kernel at run-time D) */
template <class TT, class ST> __global__ __launch_bounds__(64) void mm_nuts_generic_init_kernel(const mm_gen_nuts_args<TT, ST> a)
{
    const unsigned long long c = (unsigned long long)blockIdx.x * 64ull + (threadIdx.x & 63);
    if (c >= a.n_chains)
        return;
    mm_gstore<TT, TT *> s;
    s.base = a.store + c;
    s.stride = (size_t)a.c_pad;
    s.dim = a.dim;
    for (int i = 0; i < a.dim; ++i)
        s.st(MM_NV_X, i, a.state[c * a.dim + i]);
    mm_nuts_adapt<ST> ad = a.adapt[c];
    mm_gen_nuts_init_chain<TT, ST>(a.kind, a.P, s, &ad, a.eps_tol, a.seed, a.chain_offset + c);
    a.adapt[c] = ad;
}

/* one chain per lane, the lanes of a wave take their transitions in step (mm_nuts_run_kernel at run-time D); rows
 * leave as per-lane stores */
template <class TT, class ST> __global__ __launch_bounds__(64) void mm_nuts_generic_run_kernel(const mm_gen_nuts_args<TT, ST> a)
{
    const int lane = threadIdx.x & 63;
    __shared__ unsigned int hist_lds[MM_NUTS_JMAX + 1];
    if (lane <= MM_NUTS_JMAX)
        hist_lds[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long c = (unsigned long long)blockIdx.x * 64ull + lane;
    const bool active = c < a.n_chains;
    const unsigned long long chain = a.chain_offset + c;
    const int D = a.dim;
    mm_gstore<TT, TT *> s;
    s.base = a.store + (active ? c : 0);
    s.stride = (size_t)a.c_pad;
    s.dim = D;
    unsigned long long n_lf = 0;
    if (active) {
        for (int i = 0; i < D; ++i)
            s.st(MM_NV_X, i, a.state[c * D + i]);
        mm_nuts_adapt<ST> ad = a.adapt[c];
        TT *const rows = a.out ? a.out + (c * a.n_total + a.out_t0) * D : nullptr;
        unsigned int row = 0, m = a.m0;
        auto record = [&]() {
            if (rows)
                for (int i = 0; i < D; ++i)
                    rows[(size_t)row * D + i] = s.ld(MM_NV_X, i);
            ++row;
        };
        if (a.write_initial)
            record();
        const unsigned int total = a.n_pre + a.n_rec;
        for (unsigned int t = 0; t < total; ++t) {
            ++m;
            const mm_nuts_info inf = mm_gen_nuts_step<TT, ST>(a.kind, a.P, s, &ad, m, a.n_discard, a.target_accept_p, a.max_depth,
                                                              a.seed, chain);
            n_lf += inf.n_leapfrog;
            if (a.depth_hist)
                atomicAdd(&hist_lds[inf.depth < MM_NUTS_JMAX ? inf.depth : MM_NUTS_JMAX], 1u);
            if (t >= a.n_pre)
                record();
        }
        for (int i = 0; i < D; ++i)
            a.state[c * D + i] = s.ld(MM_NV_X, i);
        a.adapt[c] = ad;
        if (a.n_leapfrog)
            a.n_leapfrog[c] += n_lf;
    }
    __builtin_amdgcn_wave_barrier();
    if (a.depth_hist && lane <= MM_NUTS_JMAX && hist_lds[lane] != 0u)
        atomicAdd(&a.depth_hist[lane], hist_lds[lane]);
}

hipError_t mm_launch_nuts_generic_m0(const mm_gen_nuts_args<float, double> &a, int init, hipStream_t stream);
hipError_t mm_launch_nuts_generic_m1(const mm_gen_nuts_args<float, float> &a, int init, hipStream_t stream);
hipError_t mm_launch_nuts_generic_m2(const mm_gen_nuts_args<double, double> &a, int init, hipStream_t stream);
#endif /* __HIPCC__ */

#endif /* MM_NUTS_GENERIC_H */
