"""The engine's random stream ON THE DEVICE against a derivation that shares nothing with the product: Philox4x32-10
written out in numpy here, the normal's inverse CDF from scipy (`ndtri`), logarithm / cosine / sine from numpy.  The
oracle (`oracle/orng.c`) and the host build include `csrc/mm_math.h` / `csrc/mm_icdf_table.h`, so a wrong table row or a
wrong `mm_log` would be invisible to a GPU-vs-oracle comparison; `tests/test_engine_stream.py` guards them exhaustively
on the CPU, this file does it through the C ABI on the GPU box (`mmcmc_draw_noise`, then one MH step whose accept
decisions follow from these draws).  Draw schedule: `csrc/mm_rng.h` header (normative text), restated below."""
import numpy as np
import pytest

M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: [n, 4] uint32, key: (k0, k1).  Random123's philox4x32 with ten rounds (Salmon et al., SC'11)."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & M32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & M32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return np.stack(c, axis=1).astype(np.uint32)


def block(seed, chains, iteration, b):
    ctr = np.empty((chains.size, 4), dtype=np.uint32)
    ctr[:, 0] = (chains & M32).astype(np.uint32)
    ctr[:, 1] = (chains >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(iteration)
    ctr[:, 3] = np.uint32(b)
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))


def u53(hi, lo):
    return (((hi.astype(np.uint64) << np.uint64(21)) | (lo.astype(np.uint64) >> np.uint64(11))).astype(np.float64) + 1.0) * 2.0**-53


def test_the_numpy_philox_reproduces_random123s_known_answers():
    kat = [([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
           ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
           ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])]
    for c, k, want in kat:
        assert philox4x32_10(np.array([c], dtype=np.uint32), k)[0].tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2, 3, 8, 32])
def test_device_f32_noise_is_the_inverse_cdf_of_philox_words(dim):
    from scipy.special import ndtri

    from mini_mcmc_amd.core import draw_noise

    seed, off, it, n = 0x1234567ABCDEF01, (5 << 32) + 17, 77, 20000
    z, u = draw_noise(seed, off, it, n, dim, np.float32)
    chains = np.uint64(off) + np.arange(n, dtype=np.uint64)
    for b in range((dim + 3) // 4):
        w = block(seed, chains, it, b)
        if b == 0:
            s = (w[:, 0] & 255) | ((w[:, 1] & 255) << 8) | ((w[:, 2] & 255) << 16)
            assert np.array_equal(u, ((s.astype(np.float64) + 1.0) * 2.0**-24).astype(np.float32))  # exact
        for i in range(4):
            if 4 * b + i >= dim:
                break
            nn = ((w[:, i] >> 8) | 1).astype(np.float64)
            mag = -ndtri(nn * 2.0**-25)
            got = z[:, 4 * b + i].astype(np.float64)
            # which value of bit 8 means "negative" is the product's to choose; the magnitude and the symmetry are not
            assert np.max(np.abs(np.abs(got) - mag) / np.maximum(1.0, mag)) < 1.6e-7
            sgn_bit = ((w[:, i] >> 8) & 1).astype(bool)
            assert np.all(np.signbit(got) == sgn_bit) or np.all(np.signbit(got) == ~sgn_bit)
            assert np.max(np.abs(got)) <= 5.4202


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2])
def test_device_mh_paired_f32_noise_is_two_transitions_per_philox_block(dim):
    """The Metropolis-Hastings sampler's f32 stream at dim <= 2 (csrc/mm_rng.h, round 5), restated here from its text: iteration
    t takes words 2 (t & 1), 2 (t & 1) + 1 of block (chain, t >> 1, 0x20000000): normals from their top 24 bits, the accept
    uniform's high 16 bits from their low bytes, its low 8 bits from byte (t & 1) of word 0 of block (chain, t >> 1,
    0x20000001).  Both iterations of a pair, an odd and an even start; at dim 3 the MH noise is the common stream."""
    from scipy.special import ndtri

    from mini_mcmc_amd.core import draw_noise

    seed, off, n = 0x1234567ABCDEF01, (5 << 32) + 17, 20000
    chains = np.uint64(off) + np.arange(n, dtype=np.uint64)
    for it in (76, 77, 1):
        z, u = draw_noise(seed, off, it, n, dim, np.float32, sampler="mh")
        w = block(seed, chains, it >> 1, 0x20000000)
        a = block(seed, chains, it >> 1, 0x20000001)
        h = it & 1
        wa, wb = w[:, 2 * h], w[:, 2 * h + 1]
        s16 = (wa & 255) | ((wb & 255) << 8)
        s8 = (a[:, 0] >> (8 * h)) & 255
        assert np.array_equal(u, ((s16.astype(np.float64) * 256 + s8 + 1.0) * 2.0**-24).astype(np.float32))  # exact
        for i, wi in enumerate((wa, wb)[:dim]):
            nn = ((wi >> 8) | 1).astype(np.float64)
            mag = -ndtri(nn * 2.0**-25)
            got = z[:, i].astype(np.float64)
            assert np.max(np.abs(np.abs(got) - mag) / np.maximum(1.0, mag)) < 1.6e-7
            sgn_bit = ((wi >> 8) & 1).astype(bool)
            assert np.all(np.signbit(got) == sgn_bit) or np.all(np.signbit(got) == ~sgn_bit)
    z3, u3 = draw_noise(seed, off, 77, n, 3, np.float32, sampler="mh")
    z0, u0 = draw_noise(seed, off, 77, n, 3, np.float32)
    assert np.array_equal(z3, z0) and np.array_equal(u3, u0)


@pytest.mark.gpu
def test_device_mh_f32_run_follows_from_the_paired_draws():
    """Five MH transitions of 20 000 chains on the standard 2-D Gaussian in f32 through the SAMPLING kernel (whose accept test
    sees only the high 16 bits of the uniform unless it needs more): proposals and accept decisions recomputed from the
    independent restatement of the paired stream agree on every chain not within rounding of a threshold."""
    from mini_mcmc_amd.core import draw_noise, init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    n, seed, steps = 20000, 99, 5
    init = init_with_seed(n, 2, 7, np.float32)
    mh = MetropolisHastings(Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), IsotropicGaussian(0.7), init).seed(seed)
    out = mh.run(steps, 0)
    x = init.astype(np.float64)
    clear_all = np.ones(n, dtype=bool)
    for t in range(steps):
        z, u = draw_noise(seed, 0, t, n, 2, np.float32, sampler="mh")
        prop = x + 0.7 * z.astype(np.float64)
        ratio = -0.5 * (prop**2).sum(1) + 0.5 * (x**2).sum(1)
        lnu = np.log(u.astype(np.float64))
        clear_all &= np.abs(ratio - lnu) > 1e-5
        acc = ratio > lnu
        x = np.where(acc[:, None], prop, x)
        ok = clear_all
        assert np.allclose(out[ok, t, :], x[ok], rtol=0, atol=2e-5 * (t + 1))
    assert clear_all.mean() > 0.995


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2, 3, 8, 32])
def test_device_f64_noise_is_box_muller_of_philox_words(dim):
    from mini_mcmc_amd.core import draw_noise

    seed, off, it, n = 0xFEDCBA9876543210, (1 << 40) + 3, 123456, 20000
    z, u = draw_noise(seed, off, it, n, dim, np.float64)
    chains = np.uint64(off) + np.arange(n, dtype=np.uint64)
    for b in range((dim + 1) // 2):
        w = block(seed, chains, it, b)
        u1, u2 = u53(w[:, 0], w[:, 1]), u53(w[:, 2], w[:, 3])
        r = np.sqrt(-2.0 * np.log(u1))
        # 2 pi u2 in double-double so that the comparison's own argument reduction is not the error being measured
        ang = 2.0 * np.pi * (u2 - np.floor(u2 * 4.0) / 4.0)  # u2 = k/4 + f: rotate by quarter turns exactly
        q = (np.floor(u2 * 4.0).astype(np.int64)) & 3
        c0, s0 = np.cos(ang), np.sin(ang)
        cosv = np.select([q == 0, q == 1, q == 2, q == 3], [c0, -s0, -c0, s0])
        sinv = np.select([q == 0, q == 1, q == 2, q == 3], [s0, c0, -s0, -c0])
        tol = 4e-15 * np.maximum(1.0, r)
        assert np.all(np.abs(z[:, 2 * b] - r * cosv) <= tol)
        if 2 * b + 1 < dim:
            assert np.all(np.abs(z[:, 2 * b + 1] - r * sinv) <= tol)
    aux = block(seed, chains, it, 0x40000000)
    assert np.array_equal(u, u53(aux[:, 0], aux[:, 1]))  # the accept uniform: words 0, 1 of the auxiliary block, exact


@pytest.mark.gpu
def test_device_mh_step_follows_from_the_independent_draws():
    """One MH transition of 20 000 chains on the standard 2-D Gaussian (f64): proposal = x + sigma z and the accept
    decision ln u < logp(x') - logp(x) recomputed from the numpy stream above agree with the device on every chain
    that is not within rounding of the threshold (metropolis_hastings.rs:303-315)."""
    from mini_mcmc_amd.core import init_with_seed
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings

    n, seed = 20000, 99
    init = init_with_seed(n, 2, 7, np.float64)
    mh = MetropolisHastings(Gaussian2D([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]]), IsotropicGaussian(0.7), init).seed(seed)
    out = mh.run(1, 0)[:, 0, :]
    acc = mh.accept_counts.astype(bool)
    chains = np.arange(n, dtype=np.uint64)
    from mini_mcmc_amd.core import draw_noise

    z, u = draw_noise(seed, 0, 0, n, 2, np.float64)  # the handle's seed is the Philox key (mmcmc_mh_seed)
    w = block(seed, chains, 0, 0)
    r = np.sqrt(-2.0 * np.log(u53(w[:, 0], w[:, 1])))
    assert np.max(np.abs(np.hypot(z[:, 0], z[:, 1]) - r)) < 1e-13  # same stream as the run's
    prop = init + 0.7 * z
    ratio = -0.5 * (prop**2).sum(1) + 0.5 * (init**2).sum(1)
    lnu = np.log(u)
    clear = np.abs(ratio - lnu) > 1e-9
    assert clear.mean() > 0.999
    assert np.array_equal(acc[clear], (ratio > lnu)[clear])
    assert np.allclose(out[acc], prop[acc], rtol=0, atol=1e-14) and np.array_equal(out[~acc], init[~acc])
