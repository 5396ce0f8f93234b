"""The Rust wrappers (rust/mini-mcmc-hip) cannot be compiled in this image (no cargo / rustc); the calls they make are
performed by a plain-C program instead (tests/c/abi_call_sequence.c), built with gcc against include/mmcmc.h and
libmmcmc.so.  CPU box: it compiles as C99, links, loads and reports "no device" (exit 77: no CPU fallback).  GPU box:
it runs the whole sequence (MH, HMC, step, stats, device group, NUTS off-by-one) and checks the results."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_call_sequence")
    libdir = os.path.join(ROOT, "mini_mcmc_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_call_sequence.c"), "-o", exe, "-L" + libdir, "-lmmcmc",
           "-Wl,-rpath," + libdir, "-lm"]
    # libmmcmc.so's own dependencies (HIP runtime) are resolved at load time from the ROCm installation
    r = subprocess.run(cmd + ["-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def test_c_sequence_builds_loads_and_refuses_without_device(tmp_path):
    import torch

    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, env=_env(), timeout=600)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "abi sequence ok" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 77 and "no device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_sequence_runs_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0 and "abi sequence ok" in r.stdout, r.stdout + r.stderr


def test_rust_sources_declare_what_the_header_exports():
    """Every `pub fn mmcmc_*` of the -sys crate is an entry point of the header (a typo would only show at link time on
    a machine with a Rust toolchain)."""
    import re

    sys_rs = open(os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", "src", "lib.rs")).read()
    header = open(os.path.join(ROOT, "include", "mmcmc.h")).read()
    declared = set(re.findall(r"pub fn (mmcmc_\w+)\s*\(", sys_rs))
    exported = set(re.findall(r"\b(mmcmc_\w+)\s*\(", header))
    assert len(declared) > 30 and declared <= exported, sorted(declared - exported)
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", f))
    assert os.path.exists(os.path.join(ROOT, "rust", "mini-mcmc-hip", "src", "lib.rs"))
