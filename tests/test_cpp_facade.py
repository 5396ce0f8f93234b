"""include/mmcmc.hpp (C++ facade mirroring the crate's surface over the C ABI): compile with g++ against libmmcmc.so
and run.  On the CPU it must fail loudly with MMCMC_ERR_NO_DEVICE; the gpu-marked variant runs the examples' shapes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = tmp_path / "facade_test"
    libdir = os.path.join(ROOT, "mini_mcmc_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", str(exe), "-L", libdir, "-lmmcmc",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return str(exe)


def test_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([_build(tmp_path), "0"], capture_output=True, text=True)
    assert out.returncode == 0 and "facade ok (no gpu" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_facade_runs_reference_examples_on_gpu(tmp_path):
    out = subprocess.run([_build(tmp_path), "1"], capture_output=True, text=True)
    assert out.returncode == 0 and "facade ok (gpu)" in out.stdout, out.stdout + out.stderr
