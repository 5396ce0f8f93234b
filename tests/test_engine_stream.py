"""The engine's counter-based stream: Philox4x32-10 known answers, the draw schedule's distribution, and the
accuracy of the shared elementary functions (mini_mcmc_amd/csrc/mm_math.h).  CPU only.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_philox_random123_known_answers(O):
    # Random123 kat_vectors, philox4x32-10
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_engine_block_is_philox_of_chain_iteration_block(O):
    out = (C.c_uint32 * 4)()
    chain = (7 << 32) | 123456
    O.lib().o_engine_block(0x1122334455667788, chain, 99, 3, out)
    assert list(out) == O.philox4x32_10([123456, 7, 99, 3], [0x55667788, 0x11223344])


def test_engine_normals_and_uniforms_distribution(O):
    z = np.concatenate([O.engine_normals_f32(42, c, 5, 8) for c in range(20000)]).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.02 and abs((z**4).mean() - 3) < 0.1
    assert np.all(np.isfinite(z))
    u = np.array([O.engine_accept_f32(42, c, 5) for c in range(50000)], dtype=np.float64)
    assert u.min() > 0 and u.max() <= 1 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # the spare uniform is independent of the normals made from the same block
    z0 = np.array([O.engine_normals_f32(42, c, 5, 1)[0] for c in range(50000)], dtype=np.float64)
    assert abs(np.corrcoef(z0, u)[0, 1]) < 0.02
    z64 = np.concatenate([O.engine_normals_f64(42, c, 5, 4) for c in range(20000)])
    assert abs(z64.mean()) < 0.015 and abs(z64.var() - 1) < 0.03
    a = np.array([O.engine_aux_u53(42, 3, 5, k) for k in range(20000)])
    assert a.min() > 0 and a.max() <= 1 and abs(a.mean() - 0.5) < 0.01


def test_mh_paired_stream_restatement_is_two_transitions_per_block(O):
    """The MH sampler's f32 stream at dim <= 2 (csrc/mm_rng.h, round 5) as oracle/orng.c restates it, against its TEXT written out
    here on the oracle's Philox: iteration t takes words 2h, 2h + 1 (h = t & 1) of block (chain, t >> 1, 0x20000000) -- the normals
    from their top 24 bits (the same inverse-CDF map as everywhere), the accept uniform's high 16 bits from their low bytes --
    and the uniform's low 8 bits from byte h of word 0 of block (chain, t >> 1, 0x20000001); u = (s16 256 + s8 + 1) 2^-24.
    (The GPU is compared with an independent numpy statement of the same text in tests/test_gpu_stream_independent.py; the host
    twin of the engine's MH step with this restatement, reference order, in tests/test_step_parity.py.)"""
    seed = 0x1122334455667788
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for chain in (0, 5, (7 << 32) | 123456):
        for t in (0, 1, 76, 77, 1001):
            w = O.philox4x32_10([chain & 0xFFFFFFFF, chain >> 32, t >> 1, 0x20000000], key)
            a = O.philox4x32_10([chain & 0xFFFFFFFF, chain >> 32, t >> 1, 0x20000001], key)
            h = t & 1
            wa, wb = w[2 * h], w[2 * h + 1]
            z, u = O.engine_mhp_noise_f32(seed, chain, t, 2)
            want = np.zeros(2, dtype=np.float32)
            O.lib().o_engine_icdf24_words((C.c_uint32 * 2)(wa, wb), 2, want.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(z, want)
            s16 = (wa & 255) | ((wb & 255) << 8)
            s8 = (a[0] >> (8 * h)) & 255
            assert u == np.float32((s16 * 256 + s8 + 1) * 2.0**-24)
            z1, u1 = O.engine_mhp_noise_f32(seed, chain, t, 1)
            assert z1[0] == z[0] and u1 == u
    # two iterations of a pair share the block, neighbouring pairs do not: the four normals of (2k, 2k + 1) are distinct words
    zs = np.array([O.engine_mhp_noise_f32(42, c, t, 2)[0] for c in range(4000) for t in (10, 11)], dtype=np.float64).reshape(-1)
    us = np.array([O.engine_mhp_noise_f32(42, c, t, 2)[1] for c in range(4000) for t in (10, 11)], dtype=np.float64)
    assert abs(zs.mean()) < 0.03 and abs(zs.var() - 1) < 0.05 and abs(us.mean() - 0.5) < 0.02
    assert abs(np.corrcoef(us[0::2], us[1::2])[0, 1]) < 0.05  # the two uniforms of a pair: disjoint bits


def test_f32_normal_is_the_inverse_cdf_on_the_whole_24_bit_lattice(O):
    """mm_rng.h "icdf24": z = sign * -Phi^-1(n 2^-25), n = (w >> 8) | 1.  EVERY one of the 2^23 magnitudes: the
    product's evaluation (host build of mm_icdf_f32) equals the oracle's independent restatement bit for bit and is
    within 1.5e-7 max(1, |z|) of scipy's ndtri; sign = bit 8 gives the exact mirror image; the lattice has the
    moments of N(0, 1)."""
    from scipy.special import ndtri

    worst, s2, s4, prev_last = 0.0, 0.0, 0.0, None
    for k0 in range(0, 1 << 23, 1 << 21):
        k = np.arange(k0, k0 + (1 << 21), dtype=np.uint32)
        w = (k << np.uint32(9)) | np.uint32(0xAB)  # low byte is not part of the normal
        got = O.engine_host_icdf24(w)
        assert np.array_equal(got.view(np.uint32), O.engine_icdf24(w).view(np.uint32))
        neg = O.engine_host_icdf24(w | np.uint32(0x100))
        assert np.array_equal(neg.view(np.uint32), got.view(np.uint32) | np.uint32(0x80000000))
        want = -ndtri((2.0 * k.astype(np.float64) + 1.0) * 2.0**-25)
        g = got.astype(np.float64)
        worst = max(worst, float(np.max(np.abs(g - want) / np.maximum(1.0, want))))
        assert np.all(g > 0)
        # decreasing in n up to the approximation error where two cubics meet
        assert np.all(np.diff(g) <= 2.5e-7 * np.maximum(1.0, g[:-1]))
        if prev_last is not None:
            assert g[0] <= prev_last + 1e-6
        prev_last = g[-1]
        s2 += float(np.sum(g * g))
        s4 += float(np.sum(g**4))
    assert worst < 1.5e-7, worst
    assert abs(s2 / (1 << 23) - 1.0) < 1e-5 and abs(s4 / (1 << 23) - 3.0) < 1e-3
    assert O.engine_host_icdf24(np.array([0], dtype=np.uint32))[0] == pytest.approx(5.4201, abs=1e-3)  # the tail ends at Phi^-1(2^-25)


def test_f32_accept_log_is_ln_u_on_all_2_24_uniforms(O):
    """mm_lnu_f32 (exponent * ln 2 + a cubic in the mantissa on 32 segments) on EVERY accept uniform u = (s + 1) 2^-24:
    within 1.2e-7 max(1, |ln u|) of ln u, never positive, non-decreasing up to that error, exactly 0 at u = 1."""
    worst, prev_last = 0.0, None
    for s0 in range(0, 1 << 24, 1 << 22):
        s = np.arange(s0, s0 + (1 << 22), dtype=np.float64)
        u = ((s + 1.0) * 2.0**-24).astype(np.float32)
        got = O.engine_host_lnu_f32(u).astype(np.float64)
        want = np.log(u.astype(np.float64))
        worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want)))))
        assert np.all(got <= 0.0)
        assert np.all(np.diff(got) >= -2.5e-7 * np.maximum(1.0, np.abs(got[:-1])))
        if prev_last is not None:
            assert got[0] >= prev_last - 1e-6
        prev_last = got[-1]
    assert worst < 1.2e-7, worst
    assert O.engine_host_lnu_f32(np.array([1.0], dtype=np.float32))[0] == 0.0


def test_filtered_accept_test_equals_the_plain_one(O):
    """mm_ratio_exceeds_ln_u decides `ratio > ln u` from the f32 table logarithm when the ratio is outside a band around
    it and from mm_log otherwise.  (a) the band is wide enough: over 53-bit uniforms of every magnitude the table value
    of (float)u is within a QUARTER of the band of ln u; (b) the decisions agree with the plain comparison on ratios
    drawn at, just inside, just outside and far from the band, and on NaN / inf."""
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.random(2_000_000), np.ldexp(rng.random(2_000_000), -rng.integers(0, 53, 2_000_000)),
                        (rng.integers(0, 1 << 53, 1_000_000, dtype=np.uint64).astype(np.float64) + 1.0) * 2.0**-53,
                        np.array([2.0**-53, 1.0, 0.5, 1.0 - 2.0**-53])])
    u = np.clip(u, 2.0**-53, 1.0)
    lf = O.engine_host_lnu_f32(u.astype(np.float32)).astype(np.float64)
    lnu = np.log(u)
    band = 1e-6 + 1e-6 * np.abs(lf)
    assert np.max(np.abs(lf - lnu) / band) < 0.25
    for scale in (0.0, 0.2, 0.9, 1.0, 1.1, 3.0, 1e3, 1e7):
        for sign in (-1.0, 1.0):
            ratio = lnu + sign * scale * band * rng.random(u.size)
            assert O.engine_host_ratio_filter_disagreements(ratio, u) == 0
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0])
    assert O.engine_host_ratio_filter_disagreements(np.repeat(special, 4), np.tile([2.0**-53, 0.3, 1.0, 0.999], 5)) == 0


@pytest.fixture(scope="module")
def mmath(tmp_path_factory):
    """Host build of mm_math.h alone (the functions that DEFINE the engine's log/exp/sincos)."""
    d = tmp_path_factory.mktemp("mmath")
    src = d / "w.c"
    src.write_text(
        '#include "mm_math.h"\n'
        "float w_logf(float x){return mm_logf(x);} float w_expf(float x){return mm_expf(x);}\n"
        "void w_sc(float u,float*s,float*c){mm_sincos2pif(u,s,c);}\n"
        "double w_log(double x){return mm_log(x);} double w_exp(double x){return mm_exp(x);}\n"
        "void w_scd(double u,double*s,double*c){mm_sincos2pi(u,s,c);}\n"
    )
    so = d / "w.so"
    subprocess.run(
        ["gcc", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-shared", "-fPIC", "-I",
         os.path.join(ROOT, "mini_mcmc_amd", "csrc"), str(src), "-o", str(so), "-lm"], check=True)
    L = C.CDLL(str(so))
    L.w_logf.restype = C.c_float
    L.w_logf.argtypes = [C.c_float]
    L.w_expf.restype = C.c_float
    L.w_expf.argtypes = [C.c_float]
    L.w_sc.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.w_log.restype = C.c_double
    L.w_log.argtypes = [C.c_double]
    L.w_exp.restype = C.c_double
    L.w_exp.argtypes = [C.c_double]
    L.w_scd.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


def test_mm_math_f32_accuracy(mmath):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(20000), 2.0 ** rng.uniform(-24, 0, 20000), [1.0, 2.0**-24, 0.5, 0.70710678]])
    xs = xs.astype(np.float32)
    got = np.array([mmath.w_logf(float(x)) for x in xs], dtype=np.float64)
    ref = np.log(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)) < 3e-7
    assert mmath.w_logf(1.0) == 0.0
    ys = rng.uniform(-90, 88, 20000).astype(np.float32)
    got = np.array([mmath.w_expf(float(y)) for y in ys], dtype=np.float64)
    ref = np.exp(ys.astype(np.float64))
    ok = ref > 1e-37
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 3e-7
    assert mmath.w_expf(0.0) == 1.0 and mmath.w_expf(-200.0) == 0.0 and math.isinf(mmath.w_expf(100.0))
    s, c = C.c_float(), C.c_float()
    us = np.concatenate([rng.random(20000), [0.0, 0.25, 0.5, 0.75, 1.0, 0.125, 0.375]]).astype(np.float32)
    err = 0.0
    for u in us:
        mmath.w_sc(float(u), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - math.sin(2 * math.pi * float(u))), abs(c.value - math.cos(2 * math.pi * float(u))))
    assert err < 3e-7
    mmath.w_sc(0.25, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (1.0, 0.0)
    mmath.w_sc(0.5, C.byref(s), C.byref(c))
    assert (abs(s.value), c.value) == (0.0, -1.0)


def test_mm_math_f64_accuracy(mmath):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), 2.0 ** rng.uniform(-53, 3, 20000), [1.0, 2.0**-53]])
    got = np.array([mmath.w_log(float(x)) for x in xs])
    ref = np.log(xs)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)) < 5e-16
    ys = rng.uniform(-700, 700, 20000)
    got = np.array([mmath.w_exp(float(y)) for y in ys])
    ref = np.exp(ys)
    assert np.max(np.abs(got - ref) / ref) < 5e-16
    s, c = C.c_double(), C.c_double()
    err = 0.0
    for u in rng.random(20000):
        mmath.w_scd(float(u), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - math.sin(2 * math.pi * u)), abs(c.value - math.cos(2 * math.pi * u)))
    assert err < 2e-15  # the libm reference itself rounds 2*pi*u


def test_paired_noise_equals_scalar_noise_bit_for_bit(tmp_path):
    """mm_draw_noise_pair (two iterations in the two lanes of packed arithmetic, Philox counters interleaved) must
    return exactly mm_draw_noise + mm_lnu_f32 (f32) / mm_log (f64) of each iteration -- compiled here with g++ like the host
    build."""
    src = tmp_path / "pair.cpp"
    src.write_text(r'''
#include "mm_samplers.h"
#include <cstdio>
template <int D> int check(uint64_t seed) {
  int bad = 0;
  for (uint64_t chain = 0; chain < 2000; ++chain) for (uint32_t it = 0; it < 8; it += 2) {
    float za[D], zb[D], la, lb, z1[D], z2[D], u1, u2;
    mm_draw_noise_pair<D>(seed, chain * 7919u, it, za, &la, zb, &lb);
    mm_draw_noise<D>(seed, chain * 7919u, it, z1, &u1);
    mm_draw_noise<D>(seed, chain * 7919u, it + 1, z2, &u2);
    for (int i = 0; i < D; ++i) bad += (za[i] != z1[i]) + (zb[i] != z2[i]);
    bad += (la != mm_lnu_f32(u1, mm_icdf_global())) + (lb != mm_lnu_f32(u2, mm_icdf_global()));
  }
  return bad;
}
// the MH sampler's paired stream (f32, D <= 2): the pair call from an EVEN iteration (one shared block) and from an ODD one
// (two blocks) against the per-iteration call; with LN = false the pair hands out u_hi = the upper end of the uniform's
// interval, from which the filtered accept test must decide as the plain comparison does on the exact u
template <int D> int check_mhp(uint64_t seed) {
  int bad = 0;
  const mm_icdf_global tab;
  for (uint64_t chain = 0; chain < 2000; ++chain) for (uint32_t it = 0; it < 9; ++it) {
    float za[D], zb[D], la, lb, ha, hb, z1[D], z2[D], u1, u2;
    mm_draw_noise_pair<D, mm_icdf_global, true, true>(seed, chain * 7919u, it, za, &la, zb, &lb);
    mm_draw_noise<D, mm_icdf_global, true>(seed, chain * 7919u, it, z1, &u1);
    mm_draw_noise<D, mm_icdf_global, true>(seed, chain * 7919u, it + 1, z2, &u2);
    for (int i = 0; i < D; ++i) bad += (za[i] != z1[i]) + (zb[i] != z2[i]);
    bad += (la != mm_lnu_f32(u1, tab)) + (lb != mm_lnu_f32(u2, tab));
    mm_draw_noise_pair<D, mm_icdf_global, false, true>(seed, chain * 7919u, it, za, &ha, zb, &hb);
    bad += !(u1 <= ha && ha - u1 < 0x1.0p-16f) + !(u2 <= hb && hb - u2 < 0x1.0p-16f);
    const float probes[] = {la, la + 1e-6f, la - 1e-6f, 0.0f, -1.0f, -20.0f};
    for (float r : probes) {
      const uint64_t sd = seed, ch = chain * 7919u; const uint32_t t = it;
      bad += mm_ratio_exceeds_lnu_mhp(r, ha, [sd, ch, t]() { return mm_mhp_low_byte(sd, ch, t); }, tab) != (r > la);
    }
  }
  return bad;
}
int main() {
  int bad = check<1>(1) + check<2>(42) + check<3>(42) + check<4>(5) + check<5>(6) + check<8>(7) + check<16>(8) + check<32>(9);
  bad += check_mhp<1>(3) + check_mhp<2>(42);
  double za[3], zb[3], la, lb, z1[3], z2[3], u1, u2;
  mm_draw_noise_pair<3>(42, 5, 10, za, &la, zb, &lb);
  mm_draw_noise<3>(42, 5, 10, z1, &u1); mm_draw_noise<3>(42, 5, 11, z2, &u2);
  for (int i = 0; i < 3; ++i) bad += (za[i] != z1[i]) + (zb[i] != z2[i]);
  bad += (la != mm_log(u1)) + (lb != mm_log(u2));
  std::printf("%d\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "pair"
    subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-I",
                    os.path.join(ROOT, "mini_mcmc_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout


def test_engine_noise_against_statements_that_share_nothing_with_the_product(O):
    """oracle/orng.c states the engine's normals a second time WITHOUT the product's table or elementary functions:
    o_ndtri (Halley's iteration on libm's erfc; here also checked against scipy) rounded to f32, and Box-Muller from libm's
    log / sin / cos.  Data against data: over ALL 2^23 magnitudes of the f32 lattice the product's table-driven normal is
    the correctly rounded inverse CDF or its neighbour (most of them exact); a million f64 normals agree to 6e-15."""
    from scipy.special import ndtri

    p = np.concatenate([2.0 ** -np.arange(1, 26), 1.0 - 2.0 ** -np.arange(2, 30), np.linspace(1e-9, 1 - 1e-9, 2001)])
    mine = np.array([O.lib().o_ndtri(float(q)) for q in p])
    assert np.max(np.abs(mine - ndtri(p)) / np.maximum(1.0, np.abs(mine))) < 4e-15
    exact, total, worst_ulp, worst_abs = 0, 0, 0, 0.0
    for k0 in range(0, 1 << 23, 1 << 21):
        k = np.arange(k0, k0 + (1 << 21), dtype=np.uint32)
        w = (k << np.uint32(9)) | np.uint32(0x3C)
        table = O.engine_host_icdf24(w)                      # the PRODUCT's evaluation (host build of mm_rng.h)
        own = O.engine_icdf24_independent(w)                 # f64 inverse CDF, rounded
        d = np.abs(table.view(np.int32).astype(np.int64) - own.view(np.int32).astype(np.int64))
        exact += int(np.sum(d == 0))
        total += d.size
        big = np.abs(own) >= 0.25                            # where an ulp is a meaningful unit; towards 0 the bound is absolute
        worst_ulp = max(worst_ulp, int(d[big].max()) if big.any() else 0)
        if (~big).any():
            worst_abs = max(worst_abs, float(np.abs(table[~big].astype(np.float64) - own[~big].astype(np.float64)).max()))
    # measured: 72.6 % of the lattice exact, the rest one ulp; below 0.25 at most 1.5e-8 (half an ulp of 0.25)
    assert worst_ulp <= 1 and worst_abs <= 1.5e-8 and exact / total > 0.7, (worst_ulp, worst_abs, exact / total)
    z_engine = np.concatenate([O.engine_normals_f64(42, c, 5, 8) for c in range(125000)])
    L = O.lib()
    z_libm = np.array([L.o_engine_normal_f64_libm(42, c, 5, i) for c in range(125000) for i in range(8)])
    # the libm statement rounds the angle 2 pi u before sin / cos (the engine reduces in turns): the bound is absolute,
    # a few 1e-16 per unit of radius (measured: 2.7e-15 at most)
    assert np.max(np.abs(z_engine - z_libm)) < 6e-15
