"""Generate rust/mini-mcmc-hip-sys/src/lib.rs from include/mmcmc.h: every `#define MMCMC_*` integer constant, every
struct, every opaque handle and EVERY exported function, so the -sys crate cannot fall behind the header
(tests/test_c_call_sequence.py compares the committed file with a fresh generation, in both directions).

    python tools/gen_rust_sys.py            # rewrite the file
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file differs

Layouts (round 5): no Rust compiler exists in this image, so the crate's `#[repr(C)]` structs cannot be compared with the
header's by compiling both.  Instead the C compiler here is asked for `sizeof` / `_Alignof` / `offsetof` of every struct and
field of the header (`c_layouts`), the generator computes what `#[repr(C)]` gives the Rust declaration it emits
(`rust_layout`: the platform's size and alignment of every scalar, fields in order, each at the next multiple of its
alignment, the size rounded up to the struct's alignment) and refuses to generate when the two differ; the C numbers go
into the crate as `const _: () = assert!(size_of::<T>() == N)` / `offset_of!` lines -- the first `cargo build` on a machine
whose ABI differs fails to compile instead of corrupting memory -- and into `rust/mini-mcmc-hip-sys/layout_check.c` as
`_Static_assert`s against the header, which a CPU test compiles (tests/test_c_call_sequence.py), so header, generator and
crate cannot drift apart.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmcmc.h")
OUT = os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", "src", "lib.rs")
OUT_C = os.path.join(ROOT, "rust", "mini-mcmc-hip-sys", "layout_check.c")

# (size, alignment) of the Rust scalar types on the targets the library exists for (x86-64 / aarch64 Linux, LP64)
RUST_SCALARS = {"c_int": (4, 4), "c_uint": (4, 4), "c_double": (8, 8), "f32": (4, 4), "f64": (8, 8), "usize": (8, 8), "u64": (8, 8),
                "u32": (4, 4), "i32": (4, 4), "c_char": (1, 1), "u8": (1, 1)}


def rust_layout(name, structs_by_name):
    """(size, align, [(field, offset)]) of the `#[repr(C)]` struct the generator emits for `name`."""
    def size_align(t):
        t = t.strip()
        if t.startswith("*") or t.startswith("Option<"):
            return 8, 8
        m = re.match(r"\[(.+); (\d+)\]$", t)
        if m:
            sz, al = size_align(m.group(1))
            return sz * int(m.group(2)), al
        if t in RUST_SCALARS:
            return RUST_SCALARS[t]
        sz, al, _ = rust_layout(t, structs_by_name)
        return sz, al

    off, align, fields = 0, 1, []
    for fname, ftype in structs_by_name[name]:
        sz, al = size_align(ftype)
        off = (off + al - 1) // al * al
        fields.append((fname, off))
        off += sz
        align = max(align, al)
    return (off + align - 1) // align * align, align, fields


def c_layouts(structs):
    """{struct: (size, align, [(field, offset)])} as THIS machine's C compiler lays the header's structs out."""
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HEADER}"', "int main(void) {"]
    for name, fields in structs:
        lines.append(f'    printf("S {name} %zu %zu\\n", sizeof({name}), (size_t)_Alignof({name}));')
        for fname, _ in fields:
            lines.append(f'    printf("F {name} {fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["    return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(src, "w").write("\n".join(lines) + "\n")
        subprocess.run([os.environ.get("CC", "gcc"), "-std=c11", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    res = {}
    for ln in out.splitlines():
        q = ln.split()
        if q[0] == "S":
            res[q[1]] = (int(q[2]), int(q[3]), [])
        else:
            res[q[1]][2].append((q[2], int(q[3])))
    return res


def layouts(structs):
    by_name = dict(structs)
    c = c_layouts(structs)
    for name, _ in structs:
        r = rust_layout(name, by_name)
        if r != c[name]:
            raise SystemExit(f"{name}: #[repr(C)] of the generated declaration gives {r}, the C compiler {c[name]}: fix the type map")
    return c

SCALARS = {"int": "c_int", "double": "c_double", "float": "f32", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
           "int32_t": "i32", "unsigned int": "c_uint", "char": "c_char", "void": "c_void"}


def rust_type(ctype: str) -> str:
    t = " ".join(ctype.replace("*", " * ").split())
    const = t.startswith("const ")
    if const:
        t = t[len("const "):]
    stars = t.count("*")
    base = t.replace("*", "").strip()
    rt = SCALARS.get(base, base)  # structs / handles keep their C names
    if stars == 0:
        return rt
    out = rt
    for level in range(stars):
        out = ("*const " if (const and level == 0) else "*mut ") + out
    return out


def strip_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse(header_text: str):
    defines = []
    for m in re.finditer(r"^#define (MMCMC_\w+) \(?(-?\d+)\)?", header_text, flags=re.M):
        if m.group(1) != "MMCMC_H":
            defines.append((m.group(1), int(m.group(2))))
    body = strip_comments(header_text)
    structs, opaque = [], []
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", body, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            f = " ".join(f.split())
            if not f:
                continue
            arr = re.search(r"\[(\d+)\]$", f)
            if arr:
                f = f[:arr.start()].strip()
            parts = [q.strip() for q in f.split(",")]
            name1 = re.search(r"(\w+)$", parts[0]).group(1)
            ctype = parts[0][:-len(name1)].strip()
            for name in [name1] + parts[1:]:
                rt = rust_type(ctype)
                if arr:
                    rt = f"[{rt}; {arr.group(1)}]"
                fields.append((name, rt))
        structs.append((m.group(3), fields))
    for m in re.finditer(r"typedef struct (\w+) (\w+);", body):
        opaque.append(m.group(2))
    fnptrs = []
    for m in re.finditer(r"typedef void \(\*(mmcmc_\w+)\)\(([^;]*?)\);", body, flags=re.S):
        args = []
        for p in " ".join(m.group(2).split()).split(","):
            am = re.match(r"(.+?)(\w+)$", p.strip())
            args.append((am.group(2), rust_type(am.group(1).strip())))
        fnptrs.append((m.group(1), args))
    funcs = []
    for m in re.finditer(r"^(int|const char \*|void)\s*(mmcmc_\w+)\s*\(([^;{]*?)\)\s*;", body, flags=re.M | re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for i, p in enumerate(params.split(",")):
                p = p.strip()
                am = re.match(r"(.+?)(\w+)(\[\d*\])?$", p)
                ctype, pname = am.group(1).strip(), am.group(2)
                if am.group(3):
                    ctype += " *"
                if pname in ("type", "in", "fn", "mod", "ref", "self", "move", "box", "loop", "match", "use"):
                    pname += "_"
                args.append((pname, rust_type(ctype)))
        rret = {"int": "c_int", "const char *": "*const c_char", "void": None}[ret]
        funcs.append((name, args, rret))
    return defines, structs, opaque, funcs, fnptrs


def generate() -> str:
    defines, structs, opaque, funcs, fnptrs = parse(open(HEADER).read())
    L = []
    L.append("//! Raw declarations of the C ABI in `include/mmcmc.h`: every constant, struct, handle and exported function.")
    L.append("//! GENERATED by tools/gen_rust_sys.py from the header -- do not edit; the header's comments name the reference")
    L.append("//! item each entry point stands in for (file:line of mini-mcmc v0.8.3).  Status convention: 0 ok, < 0")
    L.append("//! `MMCMC_ERR_*`, > 0 a HIP error code; no panics cross the boundary.")
    L.append("#![allow(non_camel_case_types)]")
    L.append("")
    L.append("use std::os::raw::{c_char, c_double, c_int, c_uint, c_void};")
    L.append("")
    for name, val in defines:
        L.append(f"pub const {name}: c_int = {val};")
    L.append("")
    for name, fields in structs:
        L.append("#[repr(C)]")
        L.append("#[derive(Clone, Copy, Debug" + ("" if any("*" in t for _, t in fields) else ", Default") + ")]")
        L.append(f"pub struct {name} {{")
        for fname, ftype in fields:
            L.append(f"    pub {fname}: {ftype},")
        L.append("}")
    L.append("")
    L.append("// Layout of every struct as the C compiler lays out include/mmcmc.h (tools/gen_rust_sys.py: sizeof / _Alignof /")
    L.append("// offsetof, also asserted on the C side by layout_check.c): a build whose ABI disagrees does not compile.")
    lay = layouts(structs)
    for name, _ in structs:
        size, align, offs = lay[name]
        L.append(f"const _: () = assert!(core::mem::size_of::<{name}>() == {size});")
        L.append(f"const _: () = assert!(core::mem::align_of::<{name}>() == {align});")
        for fname, off in offs:
            L.append(f"const _: () = assert!(core::mem::offset_of!({name}, {fname}) == {off});")
    L.append("")
    for name in opaque:
        L.append("#[repr(C)]")
        L.append(f"pub struct {name} {{")
        L.append("    _p: [u8; 0],")
        L.append("}")
    L.append("")
    for name, args in fnptrs:  # C callbacks: nullable function pointers
        a = ", ".join(f"{n}: {t}" for n, t in args)
        L.append(f'pub type {name} = Option<unsafe extern "C" fn({a})>;')
    L.append("")
    L.append('extern "C" {')
    for name, args, ret in funcs:
        a = ", ".join(f"{n}: {t}" for n, t in args)
        L.append(f"    pub fn {name}({a})" + (f" -> {ret};" if ret else ";"))
    L.append("}")
    L.append("")
    L.append("#[allow(dead_code)]")
    L.append("fn _unused(_: c_uint) {}")
    return "\n".join(L) + "\n"


def generate_c() -> str:
    """layout_check.c: the numbers the crate asserts, asserted against the header by the C compiler."""
    _, structs, _, _, _ = parse(open(HEADER).read())
    lay = layouts(structs)
    L = ["/* GENERATED by tools/gen_rust_sys.py -- do not edit.  The struct layouts mini-mcmc-hip-sys/src/lib.rs asserts on the Rust",
         " * side (size_of / align_of / offset_of!), asserted here against include/mmcmc.h by the C compiler: compiled by",
         " * tests/test_c_call_sequence.py and by the crate's build.rs, so header, generator and crate cannot drift apart. */",
         "#include <stddef.h>", '#include "../../include/mmcmc.h"', ""]
    for name, _ in structs:
        size, align, offs = lay[name]
        L.append(f'_Static_assert(sizeof({name}) == {size}, "{name}: size");')
        L.append(f'_Static_assert(_Alignof({name}) == {align}, "{name}: alignment");')
        for fname, off in offs:
            L.append(f'_Static_assert(offsetof({name}, {fname}) == {off}, "{name}.{fname}: offset");')
    L.append("")
    L.append("int mmcmc_layout_check(void) { return 0; }")
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    text, ctext = generate(), generate_c()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text and open(OUT_C).read() == ctext else 1)
    open(OUT, "w").write(text)
    open(OUT_C, "w").write(ctext)
    print(f"{OUT}: {text.count('pub fn mmcmc_')} functions, {text.count('const _: () = assert!')} layout assertions")
