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


def test_rust_sys_crate_is_the_header_in_both_directions():
    """rust/mini-mcmc-hip-sys/src/lib.rs is GENERATED from include/mmcmc.h (tools/gen_rust_sys.py): the committed file must
    equal a fresh generation, every exported function of the header must be declared, and nothing else; every `sys::`
    item the safe wrappers use must exist in it."""
    import re

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_sys

    path = os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", "src", "lib.rs")
    sys_rs = open(path).read()
    assert sys_rs == gen_rust_sys.generate(), "run python tools/gen_rust_sys.py"
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "mmcmc.h")).read(), flags=re.S)
    declared = set(re.findall(r"pub fn (mmcmc_\w+)\s*\(", sys_rs))
    exported = set(re.findall(r"^(?:int|const char \*|void)\s*(mmcmc_\w+)\s*\(", header, flags=re.M))
    assert len(exported) >= 100 and declared == exported, (sorted(declared - exported), sorted(exported - declared))
    wrappers = open(os.path.join(ROOT, "rust", "mini-mcmc-hip", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(\w+)", wrappers))
    known = declared | set(re.findall(r"pub (?:const|struct) (\w+)", sys_rs))
    assert used <= known, sorted(used - known)
    for name in ("run_progress", "GpuHmcGroup", "GpuMhGroup", "GpuNutsGroup", "GpuTracker", "RunStats"):
        assert name in wrappers, name
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", f))
