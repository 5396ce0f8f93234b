/*
 * mm_nuts.h -- one NUTS transition of one chain, iterative (host + device, one definition).
 *
 * Restates the reference's per-chain sampler
 *     NUTSChain::step          nuts.rs:550-691      build_tree (recursive)   nuts.rs:764-946
 *     leapfrog                 nuts.rs:979-996      stop_criterion           nuts.rs:963-977
 *     find_reasonable_epsilon  nuts.rs:695-761      init_chain               nuts.rs:528-545
 * (Hoffman & Gelman Alg. 6, slice variant, dual averaging) without recursion: the 2^j leaves of a doubling are
 * produced one after the other by leapfrogging from the outer edge, and completed subtrees are merged on an explicit
 * stack in exactly the order the recursion returns them, so every uniform is drawn at the same point of the
 * algorithm and early termination (a first child with s' = 0 is handed up unmerged; a second child is merged, draw
 * included) behaves as in nuts.rs:858-929.  Kept quirks: alpha / n_alpha from the last
 * doubling only, the inner-merge uniform is always f64, direction v = +1 iff u < 0.5, biased progressive sampling,
 * tensors of type TT with scalars of type ST (the reference runs f32 tensors with T scalars), the epsilon search's
 * `&&` and stale gradient.  New: a depth cap (`max_depth`; the reference's `while s` is unbounded).
 * Arithmetic: the engine's (explicit fma, analytic gradients: mm_targets.h); stream: mm_rng.h (NUTS schedule).
 */
#ifndef MM_NUTS_H
#define MM_NUTS_H

#include "mm_samplers.h"

#define MM_NUTS_JMAX 12 /* storage bound for the subtree stack; run-time max_depth <= MM_NUTS_JMAX */

/* per-chain adaptation state (NUTSChain fields epsilon, epsilon_bar, h_bar, mu; nuts.rs:374-386) */
template <class ST> struct mm_nuts_adapt {
    ST epsilon, epsilon_bar, h_bar, mu;
};

/* constants of NUTSChain::new (nuts.rs:425-427) */
#define MM_NUTS_GAMMA 0.05
#define MM_NUTS_T0 10
#define MM_NUTS_KAPPA 0.75

/* Pending-subtree stack of one chain.  Storage is strided so that on the device consecutive lanes interleave
 * (element (slot) of lane l at base[slot * stride + l]); on the host stride = 1. */
template <class TT, class ST, int D> struct mm_nuts_stack {
    TT *vec;        /* [JMAX][3][D]: first-leaf position, first-leaf momentum, proposal */
    ST *alpha;      /* [JMAX] */
    uint32_t *cnt;  /* [JMAX][3]: level, n', n_alpha' */
    int stride;
    MM_HD TT &v(int e, int which, int d) const { return vec[(size_t)((e * 3 + which) * D + d) * stride]; }
    MM_HD ST &a(int e) const { return alpha[(size_t)e * stride]; }
    MM_HD uint32_t &c(int e, int which) const { return cnt[(size_t)(e * 3 + which) * stride]; }
    static constexpr int vec_slots = MM_NUTS_JMAX * 3 * D;
    static constexpr int alpha_slots = MM_NUTS_JMAX;
    static constexpr int cnt_slots = MM_NUTS_JMAX * 3;
};

MM_HD float mm_sqrtT(float x) { return sqrtf(x); }
MM_HD double mm_sqrtT(double x) { return sqrt(x); }
MM_HD float mm_minT(float a, float b) { return fminf(a, b); }
MM_HD double mm_minT(double a, double b) { return fmin(a, b); }

/* min(1, exp(d)), the acceptance statistic of a NUTS leaf (nuts.rs:823-826), without the exponential where it cannot matter:
 * for d >= 0 (and for a NaN, which fmin drops) the minimum is 1 whatever exp returns, for d < 0 exp(d) <= 1 is the minimum
 * itself -- the same bits as mm_minT(1, mm_exp_hotT(d)) for every d.  Written as a branch so that a wave none of whose
 * working lanes has d < 0 skips the exponential altogether (round 6: the tail of a one-chain-per-lane NUTS launch is ONE
 * live lane, half of whose leaves have d >= 0; in mode 0 the f64 exponential is ~30 two-slot instructions of its ~470 per leaf). */
#ifndef MM_NUTS_ACCEPT_BRANCH
#define MM_NUTS_ACCEPT_BRANCH 1
#endif
template <class ST> MM_HD ST mm_accept_stat(ST d)
{
#if MM_NUTS_ACCEPT_BRANCH
    ST a = ST(1);
    if (d < ST(0))
        a = mm_exp_hotT(d);
    return a;
#else
    return mm_minT(ST(1), mm_exp_hotT(d));
#endif
}

MM_HD bool mm_is_real(float x) { return x == x && x != MM_INFINITY_F && x != -MM_INFINITY_F; }
MM_HD bool mm_is_real(double x) { return x == x && x != (double)MM_INFINITY_F && x != -(double)MM_INFINITY_F; }

/* momentum draw: z[0..D) of (chain, iteration) in the element type's schedule (mm_rng.h) */
template <int D> MM_HD void mm_nuts_momentum(uint64_t seed, uint64_t chain, uint32_t iter, float *z)
{
    const mm_icdf_global tab;
    MM_UNROLL
    for (int b = 0; b < (D + 3) / 4; ++b) {
        mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        MM_UNROLL
        for (int i = 0; i < 4; ++i)
            if (4 * b + i < D)
                z[4 * b + i] = mm_icdf_f32(blk.w[i], tab);
    }
}
template <int D> MM_HD void mm_nuts_momentum(uint64_t seed, uint64_t chain, uint32_t iter, double *z)
{
    MM_UNROLL
    for (int b = 0; b < (D + 1) / 2; ++b) {
        mm_u32x4 blk = mm_block(seed, chain, iter, (uint32_t)b);
        double z0, z1;
        mm_box_muller_f64(mm_u53(blk.w[0], blk.w[1]), mm_u53(blk.w[2], blk.w[3]), &z0, &z1);
        z[2 * b] = z0;
        if (2 * b + 1 < D)
            z[2 * b + 1] = z1;
    }
}

/* nuts.rs:979-996, in place: (x, p, g) -> one leapfrog step of signed size eps; returns logp(x') */
template <class TT, class Tgt> MM_HD TT mm_nuts_leapfrog(const mm_tparams<TT> &P, TT *x, TT *p, TT *g, TT eps)
{
    constexpr int D = Tgt::dim;
    const TT h = eps * TT(0.5);
    MM_UNROLL
    for (int i = 0; i < D; ++i) {
        p[i] = mm_fma(h, g[i], p[i]);
        x[i] = mm_fma(eps, p[i], x[i]);
    }
    TT lp = Tgt::logp_grad(P, x, g);
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        p[i] = mm_fma(h, g[i], p[i]);
    return lp;
}

/* nuts.rs:963-977 */
template <class TT, int D, class Red>
MM_HD bool mm_stop_criterion(const TT *x_minus, const TT *x_plus, const TT *p_minus, const TT *p_plus)
{
    TT diff[D];
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        diff[i] = x_plus[i] - x_minus[i];
    return Red::dot(diff, p_minus) >= TT(0) && Red::dot(diff, p_plus) >= TT(0);
}

/* nuts.rs:695-761 */
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
MM_HD ST mm_find_reasonable_epsilon(const mm_tparams<TT> &P, const TT *position, const TT *mom)
{
    constexpr int D = Tgt::dim;
    ST epsilon = 1;
    const ST half = ST(0.5);
    TT g0[D], x[D], p[D], g[D];
    const TT ulogp = Tgt::logp_grad(P, position, g0);
    const TT mom_sq = Red::dot(mom, mom);
    auto leap = [&](ST e) -> TT {
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            x[i] = position[i];
            p[i] = mom[i];
            g[i] = g0[i];
        }
        return mm_nuts_leapfrog<TT, Tgt>(P, x, p, g, (TT)e);
    };
    TT ulogp_p = leap(epsilon);
    bool grad_real = true; /* grad_prime of the FIRST leapfrog only: the loop below never refreshes it (Q11) */
    MM_UNROLL
    for (int i = 0; i < D; ++i)
        grad_real = grad_real && mm_is_real(g[i]);
    ST k = 1;
    while (!mm_is_real(ulogp_p) && !grad_real) {
        k = k * half;
        ulogp_p = leap(epsilon * k);
    }
    epsilon = half * k * epsilon;
    ST lap = (ST)(double)(ulogp_p - ulogp - (Red::dot(p, p) - mom_sq) * TT(0.5));
    const ST a = (lap > mm_logT(half)) ? ST(1) : ST(-1);
    const ST ln2 = mm_logT(ST(2));
    while (a * lap > -a * ln2) {
        epsilon = (a > ST(0)) ? epsilon * ST(2) : epsilon * half; /* epsilon * 2^a */
        ulogp_p = leap(epsilon);
        lap = (ST)(double)(ulogp_p - ulogp - (Red::dot(p, p) - mom_sq) * TT(0.5));
    }
    return epsilon;
}

/* nuts.rs:528-545 init_chain, without the sample bookkeeping: always draws D normals (iteration 0 of the stream);
 * searches epsilon iff it is still the sentinel -1; mu = ln(10 epsilon). */
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
MM_HD void mm_nuts_init_chain(const mm_tparams<TT> &P, const TT *position, mm_nuts_adapt<ST> *ad, ST eps_tol,
                              uint64_t seed, uint64_t chain)
{
    constexpr int D = Tgt::dim;
    TT mom0[D];
    mm_nuts_momentum<D>(seed, chain, 0u, mom0);
    ST d = ad->epsilon + ST(1);
    if ((d < 0 ? -d : d) <= eps_tol)
        ad->epsilon = mm_find_reasonable_epsilon<TT, ST, Tgt, Red>(P, position, mom0);
    ad->mu = mm_logT(ST(10) * ad->epsilon);
}

/* diagnostics of a transition (not in the reference) */
struct mm_nuts_info {
    int depth;          /* doublings performed */
    uint32_t n_leapfrog; /* gradient evaluations inside the tree */
};

/* One transition (nuts.rs:550-691) as a resumable object: begin() | { double_begin() | leaf_step() until it
 * returns true | double_end() } while `s` | finish().  mm_nuts_step below runs these back to back (the host build, the
 * lane-synchronous kernel); the lane-asynchronous kernel (mm_nuts_kernels.h) calls them under masks, one leaf per
 * tick and lane, so that a lane whose tree is finished starts its next transition instead of idling.  Either way a
 * chain goes through exactly the same operations in the same order.
 * The outer edge in direction v is advanced on a copy (cx, cp, cg) and written back when the doubling ends. */
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>> struct mm_nuts_tree {
    static constexpr int D = Tgt::dim;
    /* of the transition */
    TT xm[D], xp[D], pm[D], pp[D], gm[D], gp[D];
    ST joint, logu, alpha;
    uint32_t n, n_alpha, aux_k, m;
    int j;
    bool s;
    mm_nuts_info info;
    /* of the doubling in progress */
    int v, sp;
    TT cx[D], cp[D], cg[D], eps_signed;
    /* the most recently completed subtree S */
    uint32_t S_level, S_n, S_nalpha;
    bool S_s;
    ST S_alpha;
    TT S_first_x[D], S_first_p[D], S_prime[D];

    /* The auxiliary uniforms (mm_rng.h: draw k = one half of Philox block AUX + (k >> 1)) come from a window of two
     * consecutive blocks, aux_q[0..3] = draws 2 aux_b .. 2 aux_b + 3, so that every block is evaluated once, not once
     * per half.  PRE = false: the window follows the draws on demand.  PRE = true: the caller has advanced it
     * (aux_advance) at a point where all lanes do it together -- the asynchronous-lane kernel, where a Philox
     * evaluation inside each of the five sections that draw would run five times per tick for a few lanes each. */
    double aux_q[4];
    uint32_t aux_b;
    MM_HD void aux_load(uint64_t seed, uint64_t chain, uint32_t b, double *q) const
    {
        const mm_u32x4 blk = mm_block(seed, chain, m, MM_AUX_BLOCK + b);
        q[0] = mm_u53(blk.w[0], blk.w[1]);
        q[1] = mm_u53(blk.w[2], blk.w[3]);
    }
    MM_HD void aux_advance(uint64_t seed, uint64_t chain)
    {
        aux_q[0] = aux_q[2];
        aux_q[1] = aux_q[3];
        aux_b += 1;
        aux_load(seed, chain, aux_b + 1, aux_q + 2);
    }
    /* true when the window's first block is used up: at most three draws can follow before the next check */
    MM_HD bool aux_stale() const { return (aux_k >> 1) > aux_b; }
    template <bool PRE = false> MM_HD double aux(uint64_t seed, uint64_t chain)
    {
        const uint32_t k = aux_k++;
        if (!PRE) {
            while ((k >> 1) > aux_b + 1)
                aux_advance(seed, chain);
        }
        const uint32_t r = k - 2u * aux_b;
        const double lo = (r & 1u) ? aux_q[1] : aux_q[0], hi = (r & 1u) ? aux_q[3] : aux_q[2];
        return (r & 2u) ? hi : lo;
    }

    /* m_: the 1-based global step count (self.m after the increment) */
    MM_HD void begin(const mm_tparams<TT> &P, const TT *x, uint32_t m_, uint64_t seed, uint64_t chain)
    {
        m = m_;
        aux_k = 0;
        aux_b = 0;
        aux_load(seed, chain, 0, aux_q);
        aux_load(seed, chain, 1, aux_q + 2);
        begin_with(P, x, seed, chain, aux(seed, chain));
    }
    /* the same with the transition's first auxiliary uniform handed in (m already set): kernels that keep the uniforms
     * somewhere else than in aux_q (mm_nuts_pair_kernel: a ring in LDS) */
    MM_HD void begin_with(const mm_tparams<TT> &P, const TT *x, uint64_t seed, uint64_t chain, double u0)
    {
        TT mom0[D], grad[D];
        mm_nuts_momentum<D>(seed, chain, m, mom0);
        const TT ulogp = Tgt::logp_grad(P, x, grad);
        joint = (ST)(double)(ulogp - Red::dot(mom0, mom0) * TT(0.5));
        const ST exp1_obs = (ST)(-mm_log(u0)); /* Exp(1) by inversion of the first auxiliary uniform */
        logu = joint - exp1_obs;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            xm[i] = xp[i] = x[i];
            pm[i] = pp[i] = mom0[i];
            gm[i] = gp[i] = grad[i];
        }
        j = 0;
        n = 1;
        s = true;
        alpha = 0;
        n_alpha = 0;
        info.depth = 0;
        info.n_leapfrog = 0;
    }

    template <bool PRE = false> MM_HD void double_begin(const mm_nuts_adapt<ST> &ad, uint64_t seed, uint64_t chain)
    {
        double_begin_with(ad.epsilon, aux<PRE>(seed, chain));
    }
    MM_HD void double_begin_with(ST epsilon, double u)
    {
        const ST u_run_1 = (ST)u;
        v = (u_run_1 < ST(0.5)) ? 1 : -1;
        const bool neg = v == -1;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            cx[i] = neg ? xm[i] : xp[i];
            cp[i] = neg ? pm[i] : pp[i];
            cg[i] = neg ? gm[i] : gp[i];
        }
        eps_signed = (TT)((ST)v * epsilon);
        sp = 0; /* pending first children on the stack */
        S_level = 0;
        S_n = 0;
        S_nalpha = 0;
        S_s = true;
        S_alpha = 0;
    }

    /* one leaf: the leapfrog from the outer edge and the one-leaf subtree S it makes */
    MM_HD void leaf(const mm_tparams<TT> &P)
    {
        const TT lp = mm_nuts_leapfrog<TT, Tgt>(P, cx, cp, cg, eps_signed);
        info.n_leapfrog += 1;
        const ST jointp = (ST)(double)(lp - Red::dot(cp, cp) * TT(0.5));
        S_level = 0;
        S_n = (logu < jointp) ? 1u : 0u;
        S_s = (logu - ST(1000)) < jointp;
        S_alpha = mm_accept_stat<ST>(jointp - joint);
        S_nalpha = 1;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            S_first_x[i] = cx[i];
            S_first_p[i] = cp[i];
            S_prime[i] = cx[i];
        }
    }

    /* one step of handing S up the (implicit) recursion */
    enum { HAND_MORE = 0, HAND_NEXT_LEAF = 1, HAND_DONE = 2 };
    template <bool PRE = false>
    MM_HD int hand_up(uint64_t seed, uint64_t chain, const mm_nuts_stack<TT, ST, Tgt::dim> &stk)
    {
        return hand_up_with(stk, [&]() -> double { return this->template aux<PRE>(seed, chain); });
    }
    /* is S the second child of a pair whose first child waits on the stack? (then the next hand_up step is a merge) */
    MM_HD bool sibling_waits(const mm_nuts_stack<TT, ST, Tgt::dim> &stk) const
    {
        return sp > 0 && stk.c(sp - 1, 0) == S_level;
    }
    /* draw(): the next auxiliary uniform, called only when a merge takes place */
    template <class Draw>
    MM_HD int hand_up_with(const mm_nuts_stack<TT, ST, Tgt::dim> &stk, Draw &&draw)
    {
        if (S_level == (uint32_t)j)
            return HAND_DONE;
        if (sp > 0 && stk.c(sp - 1, 0) == S_level) {
            /* S is the second child: merge with its sibling T1 (nuts.rs:900-928) */
            const int e = sp - 1;
            const uint32_t n1 = stk.c(e, 1);
            const double u = draw(); /* always f64 (nuts.rs:910) */
            uint32_t den = n1 + S_n;
            if (den < 1)
                den = 1;
            const bool take2 = u < ((double)S_n / (double)den);
            TT fx[D], fp[D];
            MM_UNROLL
            for (int i = 0; i < D; ++i) {
                fx[i] = stk.v(e, 0, i);
                fp[i] = stk.v(e, 1, i);
                if (!take2)
                    S_prime[i] = stk.v(e, 2, i);
            }
            S_n += n1;
            const bool crit = (v == -1) ? mm_stop_criterion<TT, D, Red>(cx, fx, cp, fp)
                                        : mm_stop_criterion<TT, D, Red>(fx, cx, fp, cp);
            S_s = S_s && crit; /* the sibling's s' is 1, or it would not be waiting */
            S_alpha = stk.a(e) + S_alpha;
            S_nalpha += stk.c(e, 2);
            MM_UNROLL
            for (int i = 0; i < D; ++i) {
                S_first_x[i] = fx[i];
                S_first_p[i] = fp[i];
            }
            S_level += 1;
            sp -= 1;
        } else if (S_s) {
            /* first child, still valid: wait for the sibling */
            MM_UNROLL
            for (int i = 0; i < D; ++i) {
                stk.v(sp, 0, i) = S_first_x[i];
                stk.v(sp, 1, i) = S_first_p[i];
                stk.v(sp, 2, i) = S_prime[i];
            }
            stk.a(sp) = S_alpha;
            stk.c(sp, 0) = S_level;
            stk.c(sp, 1) = S_n;
            stk.c(sp, 2) = S_nalpha;
            sp += 1;
            return HAND_NEXT_LEAF;
        } else {
            /* first child with s' = 0: the parent returns it as it is (nuts.rs:858 not taken) */
            S_level += 1;
        }
        return (S_level == (uint32_t)j) ? HAND_DONE : HAND_MORE;
    }

    /* ---- leaves taken in pairs (mm_nuts_pair_kernel): level 0 of the pending-subtree stack in registers ----
     * A first child at level 0 is a single leaf: its first-leaf state and its proposal are the leaf itself, so what waits
     * for the sibling is (x, p) of that leaf, its alpha and its n' -- no trip to the stack.  pair_first() is hand_up's
     * decision for a level-0 subtree, pair_second() its merge branch with the sibling read from L0_* instead of the stack:
     * the same operations in the same order, so the chain's numbers do not change. */
    TT L0_x[D], L0_p[D];
    ST L0_alpha;
    uint32_t L0_n;
    MM_HD int pair_first()
    {
        if (j == 0)
            return HAND_DONE; /* a doubling of one leaf */
        if (S_s) {
            MM_UNROLL
            for (int i = 0; i < D; ++i) {
                L0_x[i] = cx[i];
                L0_p[i] = cp[i];
            }
            L0_alpha = S_alpha;
            L0_n = S_n;
            return HAND_NEXT_LEAF;
        }
        S_level = 1; /* first child with s' = 0: handed up unmerged, its sibling is never built */
        return (S_level == (uint32_t)j) ? HAND_DONE : HAND_MORE;
    }
    MM_HD int pair_second(double u)
    {
        const uint32_t n1 = L0_n;
        uint32_t den = n1 + S_n;
        if (den < 1)
            den = 1;
        const bool take2 = u < ((double)S_n / (double)den);
        MM_UNROLL
        for (int i = 0; i < D; ++i)
            if (!take2)
                S_prime[i] = L0_x[i];
        S_n += n1;
        const bool crit = (v == -1) ? mm_stop_criterion<TT, D, Red>(cx, L0_x, cp, L0_p)
                                    : mm_stop_criterion<TT, D, Red>(L0_x, cx, L0_p, cp);
        S_s = S_s && crit;
        S_alpha = L0_alpha + S_alpha;
        S_nalpha += 1u;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            S_first_x[i] = L0_x[i];
            S_first_p[i] = L0_p[i];
        }
        S_level = 1;
        return (S_level == (uint32_t)j) ? HAND_DONE : HAND_MORE;
    }
    /* the hand_up steps that need no uniform (a push, or a first child with s' = 0 moving up): false when the next step
     * is a merge */
    MM_HD int hand_up_free(const mm_nuts_stack<TT, ST, Tgt::dim> &stk)
    {
        if (S_level == (uint32_t)j)
            return HAND_DONE;
        if (S_s) {
            MM_UNROLL
            for (int i = 0; i < D; ++i) {
                stk.v(sp, 0, i) = S_first_x[i];
                stk.v(sp, 1, i) = S_first_p[i];
                stk.v(sp, 2, i) = S_prime[i];
            }
            stk.a(sp) = S_alpha;
            stk.c(sp, 0) = S_level;
            stk.c(sp, 1) = S_n;
            stk.c(sp, 2) = S_nalpha;
            sp += 1;
            return HAND_NEXT_LEAF;
        }
        S_level += 1;
        return (S_level == (uint32_t)j) ? HAND_DONE : HAND_MORE;
    }

    /* one leaf and the merges it completes; true when the doubling is complete (or was cut short) */
    MM_HD bool leaf_step(const mm_tparams<TT> &P, uint64_t seed, uint64_t chain,
                         const mm_nuts_stack<TT, ST, Tgt::dim> &stk)
    {
        leaf(P);
        int r;
        do
            r = hand_up(seed, chain, stk);
        while (r == HAND_MORE);
        return r == HAND_DONE;
    }

    /* x[D]: the chain's position, replaced by the doubling's proposal when it is accepted */
    template <bool PRE = false> MM_HD void double_end(TT *x, uint64_t seed, uint64_t chain, int max_depth)
    {
        double_end_with(x, max_depth, aux<PRE>(seed, chain));
    }
    MM_HD void double_end_with(TT *x, int max_depth, double u)
    {
        const bool neg = v == -1;
        MM_UNROLL
        for (int i = 0; i < D; ++i) {
            if (neg) {
                xm[i] = cx[i];
                pm[i] = cp[i];
                gm[i] = cg[i];
            } else {
                xp[i] = cx[i];
                pp[i] = cp[i];
                gp[i] = cg[i];
            }
        }
        alpha = S_alpha; /* from the LAST doubling only (nuts.rs:614-615, 649-650) */
        n_alpha = S_nalpha;
        const ST tmp = mm_minT(ST(1), (ST)S_n / (ST)n);
        const ST u_run_2 = (ST)u;
        if (S_s && (u_run_2 < tmp)) {
            MM_UNROLL
            for (int i = 0; i < D; ++i)
                x[i] = S_prime[i];
        }
        n += S_n;
        s = S_s && mm_stop_criterion<TT, D, Red>(xm, xp, pm, pp);
        j += 1;
        if (j >= max_depth)
            s = false; /* depth cap: not in the reference */
        info.depth = j;
    }

    /* dual averaging (nuts.rs:676-690) */
    MM_HD void finish(mm_nuts_adapt<ST> *ad, uint32_t n_discard, ST target_accept_p) const
    {
        ST eta = ST(1) / (ST)(m + MM_NUTS_T0);
        ad->h_bar = (ST(1) - eta) * ad->h_bar + eta * (target_accept_p - alpha / (ST)n_alpha);
        if (m <= n_discard) {
            const ST _m = (ST)m;
            ad->epsilon = mm_expT(ad->mu - mm_sqrtT(_m) / ST(MM_NUTS_GAMMA) * ad->h_bar);
            eta = mm_expT(-ST(MM_NUTS_KAPPA) * mm_logT(_m)); /* m^-kappa */
            ad->epsilon_bar = mm_expT((ST(1) - eta) * mm_logT(ad->epsilon_bar) + eta * mm_logT(ad->epsilon));
        } else {
            ad->epsilon = ad->epsilon_bar;
        }
    }
};

/* nuts.rs:550-691.  x[D] is updated in place; m is the 1-based global step count (self.m after the increment). */
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
MM_HD mm_nuts_info mm_nuts_step(const mm_tparams<TT> &P, TT *x, mm_nuts_adapt<ST> *ad, uint32_t m, uint32_t n_discard,
                                ST target_accept_p, int max_depth, uint64_t seed, uint64_t chain,
                                const mm_nuts_stack<TT, ST, Tgt::dim> &stk)
{
    mm_nuts_tree<TT, ST, Tgt, Red> t;
    t.begin(P, x, m, seed, chain);
    while (t.s) {
        t.double_begin(*ad, seed, chain);
        while (!t.leaf_step(P, seed, chain, stk)) {
        }
        t.double_end(x, seed, chain, max_depth);
    }
    t.finish(ad, n_discard, target_accept_p);
    return t.info;
}

/* The same transition with the leaves taken in pairs (pair_first / pair_second / hand_up_free): the order in which
 * mm_nuts_pair_kernel walks a tree, run here back to back.  Bit-identical to mm_nuts_step by construction; the host
 * build compares the two (tests/test_nuts_parity_cpu.py). */
template <class TT, class ST, class Tgt, class Red = mm_red_seq<TT, Tgt::dim>>
MM_HD mm_nuts_info mm_nuts_step_pairs(const mm_tparams<TT> &P, TT *x, mm_nuts_adapt<ST> *ad, uint32_t m, uint32_t n_discard,
                                      ST target_accept_p, int max_depth, uint64_t seed, uint64_t chain,
                                      const mm_nuts_stack<TT, ST, Tgt::dim> &stk)
{
    using Tree = mm_nuts_tree<TT, ST, Tgt, Red>;
    Tree t;
    t.begin(P, x, m, seed, chain);
    auto draw = [&]() -> double { return t.aux(seed, chain); };
    while (t.s) {
        t.double_begin(*ad, seed, chain);
        for (;;) {
            t.leaf(P);
            int r = t.pair_first();
            if (r == Tree::HAND_NEXT_LEAF) {
                t.leaf(P);
                r = t.pair_second(draw());
            }
            while (r == Tree::HAND_MORE)
                r = t.sibling_waits(stk) ? t.hand_up_with(stk, draw) : t.hand_up_free(stk);
            if (r == Tree::HAND_DONE)
                break;
        }
        t.double_end(x, seed, chain, max_depth);
    }
    t.finish(ad, n_discard, target_accept_p);
    return t.info;
}

#endif /* MM_NUTS_H */
