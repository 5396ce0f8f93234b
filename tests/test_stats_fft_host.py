"""The statistics kernel's wave-level FFT (mini_mcmc_amd/csrc/mm_stats_fft.h) on the CPU: oracle/engine_host.cpp runs the
kernel's own per-lane code for the 64 lanes of a wave, phase by phase (pass 1 -> LDS -> pass 2 -> LDS -> pass 3), and
the result must be |FFT_N(a + i b)|^2 in the bin order the kernel's fold assumes -- the index maps, twiddles and butterfly
networks are what this pins before any GPU sees them (the GPU tests then compare R-hat / ESS with oracle/stats.c,
stats.rs:416-620)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

_fp = C.POINTER(C.c_float)


def _power(r1, a, b):
    E = O.engine_host_lib()
    n = 64 * r1
    out = np.zeros(n, dtype=np.float32)
    rc = E.eh_fft_power(r1, a.ctypes.data_as(_fp), b.ctypes.data_as(_fp), len(a), out.ctypes.data_as(_fp))
    assert rc == 0
    return out


@pytest.mark.parametrize("r1,m", [(8, 200), (8, 256), (8, 101), (8, 2), (16, 500), (16, 512), (16, 257), (32, 1000),
                                  (32, 1024), (32, 513)])
def test_wave_fft_power_spectrum_and_lag_sums(r1, m):
    rng = np.random.default_rng(1000 * r1 + m)
    a = rng.standard_normal(m).astype(np.float32)
    b = (3.0 * rng.standard_normal(m)).astype(np.float32)
    n = 64 * r1
    s = _power(r1, a, b)
    z = np.zeros(n, dtype=np.complex128)
    z[:m] = a.astype(np.float64) + 1j * b.astype(np.float64)
    ref = np.abs(np.fft.fft(z)) ** 2
    assert np.max(np.abs(s - ref)) <= 5e-7 * ref.max()
    # one inverse for both half-chains: c_k = (1 / N) sum_f S(f) cos(2 pi f k / N) = sum_t a_t a_(t+k) + sum_t b_t b_(t+k);
    # the cross term of the packed transform is odd in f and drops out of the cosine sum (mm_stats_fft.h)
    k = np.arange(m)
    ck = (np.cos(2 * np.pi * np.outer(k, np.arange(n)) / n) @ s.astype(np.float64)) / n
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    direct = np.array([a64[:m - j] @ a64[j:] + b64[:m - j] @ b64[j:] for j in k])
    assert np.max(np.abs(ck - direct)) <= 3e-7 * direct[0]


def test_wave_fft_rejects_lengths_it_cannot_pad():
    E = O.engine_host_lib()
    a = np.zeros(300, dtype=np.float32)
    out = np.zeros(512, dtype=np.float32)
    assert E.eh_fft_power(8, a.ctypes.data_as(_fp), a.ctypes.data_as(_fp), 300, out.ctypes.data_as(_fp)) != 0
