"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/mmcmc.h
declares, fails loudly without a GPU (no CPU fallback), and its host-side pieces (init stream, statistics finish,
summaries) agree with the oracle.  No compute entry point is exercised here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import mini_mcmc_amd

    return mini_mcmc_amd.lib()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mmcmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmcmc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mini_mcmc_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmcmc.h but not exported by libmmcmc.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.mmcmc_version() == 102


def test_header_is_plain_c(tmp_path):
    """include/mmcmc.h is the drop-in boundary a `-sys` crate / cgo / JNI stub binds: it must compile as C99, pedantic."""
    import subprocess

    src = tmp_path / "c99.c"
    src.write_text('#include "mmcmc.h"\nint main(void) { return MMCMC_OK; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "c99.o")], check=True)


def test_product_does_not_link_or_reference_the_oracle():
    import subprocess

    so = os.path.join(ROOT, "mini_mcmc_amd", "libmmcmc.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "oracle" not in needed and "engine_host" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mini_mcmc_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert '#include "../../oracle' not in text and "liboracle" not in text, f


def test_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mini_mcmc_amd import MmcmcError
    from mini_mcmc_amd.core import init_det
    from mini_mcmc_amd.distributions import Gaussian2D, IsotropicGaussian, RosenbrockND
    from mini_mcmc_amd.hmc import HMC
    from mini_mcmc_amd.metropolis_hastings import MetropolisHastings
    from mini_mcmc_amd.stats import split_rhat_mean_ess

    with pytest.raises(MmcmcError) as e:
        MetropolisHastings(Gaussian2D([0, 0], [[1, 0], [0, 1]]), IsotropicGaussian(1.0), init_det(4, 2))
    assert e.value.status == -4
    with pytest.raises(MmcmcError):
        HMC(RosenbrockND(3), init_det(4, 3, np.float32), 0.032, 10)
    with pytest.raises(MmcmcError):
        split_rhat_mean_ess(np.zeros((2, 10, 2), dtype=np.float32))
    n = C.c_int(-1)
    assert lib.mmcmc_device_count(C.byref(n)) == -4 and n.value == 0


def test_argument_validation(lib):
    from mini_mcmc_amd import _lib as L

    h = C.c_void_p()
    t = L.TargetDesc()
    t.kind, t.dim = 99, 2
    p = L.ProposalDesc()
    p.std = 1.0
    init = np.zeros((4, 2))
    assert lib.mmcmc_mh_create(C.byref(h), C.byref(t), C.byref(p), init.ctypes.data, 4, 1, 0) == -2  # unknown kind
    t.kind, t.dim = L.GAUSSIAN2D, 3
    assert lib.mmcmc_mh_create(C.byref(h), C.byref(t), C.byref(p), init.ctypes.data, 4, 1, 0) == -3  # shape
    t.dim = 2
    assert lib.mmcmc_mh_create(C.byref(h), C.byref(t), C.byref(p), init.ctypes.data, 0, 1, 0) == -1  # zero chains
    assert lib.mmcmc_mh_create(C.byref(h), C.byref(t), C.byref(p), None, 4, 1, 0) == -1
    p.std = -1.0
    assert lib.mmcmc_mh_create(C.byref(h), C.byref(t), C.byref(p), init.ctypes.data, 4, 1, 0) == -1
    assert lib.mmcmc_mh_run(None, 1, 1, None, 0, None, None) == -1
    assert lib.mmcmc_status_string(-2).decode().startswith("unsupported")


def test_init_with_seed_is_the_reference_stream(lib, O):
    from mini_mcmc_amd.core import init_det, init_with_seed

    # two independent restatements (C++ in the product, C in the oracle) of rand 0.9 / rand_distr 0.5
    for n, d, seed in [(4, 2, 42), (65, 3, 42), (1000, 5, 7), (3, 32, 123456789)]:
        assert np.array_equal(init_with_seed(n, d, seed), O.init_with_seed(n, d, seed))
    exp = np.array([[0.83439755, -0.51496293], [1.40772757, 0.46445486], [0.95366687, 0.27411556],
                    [-1.37731726, 0.41445339]], dtype=np.float32)
    np.testing.assert_allclose(init_det(4, 2, np.float32), exp, atol=2e-7)


def test_stats_finish_matches_oracle(lib, O):
    # feed the host finish exact sufficient statistics computed in f64 and compare with the oracle's full path
    from mini_mcmc_amd.stats import stats_finish

    rng = np.random.default_rng(0)
    for (c, n, p) in [(4, 60, 2), (3, 200, 3), (8, 501, 1)]:
        x = np.zeros((c, n, p), dtype=np.float32)
        e = rng.standard_normal((c, n, p))
        for t in range(1, n):
            x[:, t] = 0.6 * x[:, t - 1] + e[:, t]
        x[:, :, 0] += np.arange(c)[:, None] * 0.3
        m = n // 2
        sp = np.concatenate([x[:, :m], x[:, n - m:]], axis=0).astype(np.float64)
        mu = sp.mean(axis=1)
        y = sp - mu[:, None, :]
        ssq = (y**2).sum(axis=1)
        ac = np.stack([(y[:, : m - lag] * y[:, lag:]).sum(axis=1).sum(axis=0) for lag in range(m)])
        rhat, ess = stats_finish(mu, ssq, ac)
        r0, e0 = O.split_rhat_mean_ess(x)
        np.testing.assert_allclose(rhat, r0, rtol=5e-5)
        np.testing.assert_allclose(ess, e0, rtol=2e-3)


def test_basic_stats_matches_oracle(lib, O, kats):
    from mini_mcmc_amd.stats import basic_stats

    rng = np.random.default_rng(1)
    for n in (1 + 1, 5, 8, 101):
        d = rng.standard_normal(n).astype(np.float32) * 10
        a, b = basic_stats("x", d), O.basic_stats(d)
        assert (a.min, a.median, a.max) == (b["min"], b["median"], b["max"])
        assert a.mean == pytest.approx(b["mean"], rel=1e-6) and a.std == pytest.approx(b["std"], rel=1e-5)
    s = basic_stats("ESS", np.array([3.0, 1.0, 2.0, 5.0], dtype=np.float32))
    assert str(s) == "ESS in [1.00, 5.00], median: 2.00, mean: 2.75 ± 1.71"


def test_rtc_compiler_info_needs_no_gpu():
    """mmcmc_rtc_compiler_info (ABI 102): which copy of hipRTC the process would compile units with, the HIP runtime it bound
    and the compiler the library was built with -- answered without a device (round 6: the NUTS kernel that "hipRTC"
    miscompiled in rounds 3-4 was miscompiled by the 7.0.2 compiler PyTorch bundles, not by the system's 7.2)."""
    from mini_mcmc_amd import _lib as L
    from mini_mcmc_amd.distributions import rtc_compiler_info

    v = rtc_compiler_info()
    assert v["built_with"][0] >= 6
    assert v["hiprtc_path"] is None or "hiprtc" in v["hiprtc_path"]
    assert v["process_hip"][0] in (0, 6, 7, 8)
    lib = L.lib()
    assert lib.mmcmc_rtc_compiler_info(None, 0, None, None) in (0, -2)  # every pointer may be NULL
