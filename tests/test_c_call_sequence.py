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


def test_rust_sys_crate_asserts_the_header_layouts(tmp_path):
    """No Rust compiler here, so layouts are pinned from both sides by what IS here (tools/gen_rust_sys.py): the C compiler's
    sizeof / _Alignof / offsetof of every struct of include/mmcmc.h (a) equal what `#[repr(C)]` gives the generated Rust
    declaration (the generator computes it and refuses otherwise), (b) stand in the crate as compile-time assertions --
    `cargo build` on an ABI that disagrees fails instead of corrupting memory -- and (c) stand in layout_check.c as
    `_Static_assert`s that gcc checks against the header here (and the crate's build.rs there).  A header that moves a field
    without the crate being regenerated fails (c)."""
    import re

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_sys

    crate = os.path.join(ROOT, "rust", "mini-mcmc-hip-sys")
    sys_rs = open(os.path.join(crate, "src", "lib.rs")).read()
    check_c = open(os.path.join(crate, "layout_check.c")).read()
    assert check_c == gen_rust_sys.generate_c(), "run python tools/gen_rust_sys.py"
    _, structs, _, _, _ = gen_rust_sys.parse(open(os.path.join(ROOT, "include", "mmcmc.h")).read())
    assert {n for n, _ in structs} >= {"mmcmc_target_desc", "mmcmc_proposal_desc", "mmcmc_basic_stats", "mmcmc_run_stats", "mmcmc_timing"}
    by_name = dict(structs)
    c = gen_rust_sys.c_layouts(structs)
    for name, fields in structs:
        size, align, offs = c[name]
        assert gen_rust_sys.rust_layout(name, by_name) == (size, align, offs)
        assert f"assert!(core::mem::size_of::<{name}>() == {size});" in sys_rs
        assert f"assert!(core::mem::align_of::<{name}>() == {align});" in sys_rs
        assert f'_Static_assert(sizeof({name}) == {size},' in check_c
        assert len(offs) == len(fields)
        for fname, off in offs:
            assert f"assert!(core::mem::offset_of!({name}, {fname}) == {off});" in sys_rs
            assert f"_Static_assert(offsetof({name}, {fname}) == {off}," in check_c
    assert c["mmcmc_target_desc"][0] == 80 and dict(c["mmcmc_target_desc"][2])["matrix"] == 72  # i32, i32, [f64; 8], *const f64
    # the committed file compiles against the header ...
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", os.path.join(crate, "layout_check.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # ... and stops compiling when the header drifts (a field inserted into mmcmc_target_desc)
    inc = tmp_path / "include"
    inc.mkdir()
    hdr = open(os.path.join(ROOT, "include", "mmcmc.h")).read()
    drifted = hdr.replace("    int32_t dim;\n    double params[8];", "    int32_t dim;\n    int32_t flags;\n    double params[8];", 1)
    assert drifted != hdr
    (inc / "mmcmc.h").write_text(drifted)
    d2 = tmp_path / "rust" / "sys"
    d2.mkdir(parents=True)
    (d2 / "layout_check.c").write_text(check_c)
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", str(d2 / "layout_check.c")], capture_output=True, text=True)
    assert r.returncode != 0 and "mmcmc_target_desc" in r.stderr
