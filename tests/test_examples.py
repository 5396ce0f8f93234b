"""The reference's examples (examples/*.rs and their `test_main`s), run through the GPU engine: examples/*.py."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location("example_" + name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_minimal_mh():
    s = _load("minimal_mh").main()
    assert s.shape == (4, 1000, 2) and np.isfinite(s).all()


def test_minimal_hmc():
    s = _load("minimal_hmc").main()
    assert s.shape == (4, 400, 3) and np.isfinite(s).all()


def test_minimal_nuts():
    s, stats = _load("minimal_nuts").main()
    assert s.shape == (4, 400, 2) and np.isfinite(s).all()
    assert "ESS" in str(stats) or "ess" in str(stats).lower()


def test_gauss_mh(tmp_path):
    s, stats, path = _load("gauss_mh").main(seed=7, out_dir=str(tmp_path))
    assert s.shape == (4, 1250, 2)
    pooled = s.reshape(-1, 2)
    assert np.all(np.abs(pooled.mean(axis=0)) < 0.5)
    with open(path) as f:
        header = f.readline().strip()
        rows = sum(1 for _ in f)
    assert header == "chain,observation,dim_0,dim_1" and rows == 4 * 1250  # io/csv.rs:47-69


def test_rosenbrock_mh():
    s, _ = _load("rosenbrock_mh").main()
    assert s.shape == (4, 1250, 2) and np.isfinite(s).all()


def test_rosenbrock3d_hmc():
    s, _ = _load("rosenbrock3d_hmc").main()
    assert s.shape == (4, 400, 3)


def test_mixture_gibbs():
    s = _load("mixture_gibbs").main(seed=3)
    assert s.shape == (4, 1100, 2)
    z = s[..., 1]
    assert set(np.unique(z)).issubset({0.0, 1.0})
    # gibbs.rs:389-418: mixture mean within 10 % (here: four chains x 1100, a looser band)
    assert abs(s[..., 0].mean() - (0.25 * -2.0 + 0.75 * 3.0)) < 0.5


def test_poisson_mh():
    chain, freqs, pmf = _load("poisson_mh").main()
    assert chain.shape == (10_000,) and chain.min() >= 0
    assert np.all(np.abs(freqs[:11] - pmf[:11]) < 0.05)  # tests/metrohast_poisson_test.rs:90-249's tolerance
