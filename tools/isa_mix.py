"""Static instruction mix of one kernel of libmmcmc.so (no GPU needed):
    python3 tools/isa_mix.py <mangled-name substring> [--so path] [--json]
Extracts every gfx950 code object from the library's clang offload bundles, disassembles the one that defines the
kernel (llvm-objdump --symbolize-operands) and counts vector instructions by issue class, over the whole kernel and
over its LOOP BODIES (address ranges closed by a backward branch) -- the part that runs per transition.

Two-slot classes (tools/issue_rate.hip; MI355X_MICROARCH.md 'vector-instruction ISSUE cost'): packed
v_pk_*, 32x32 integer multiplies (v_mul_lo_u32, v_mul_hi_u32 / _i32, v_mad_u64_u32 / _i64_i32), every *_f64, and the
transcendentals (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos).  `double_slot_share` = two-slot vector
instructions / all vector instructions inside loop bodies.  It is a STATIC share (loop trip counts are not known to a
disassembly); tools/summarize_pmc.py stores it beside the SQ counters with that label."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

TWO_SLOT = re.compile(r"^v_(pk_|mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|exp_|log_|rcp_|rsq_|sqrt_|sin_|cos_)|^v_\w+_f64")


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        end = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                yield data[i + off:i + off + size]
            end = max(end, i + off + size)
        pos = max(end, i + len(MAGIC))


def kernel_text(so, want):
    for co in code_objects(so):
        if want.encode() not in co:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", f.name], capture_output=True,
                                 text=True, check=True).stdout
        lines, name, take = [], None, False
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
            if m and not m.group(1).startswith("L"):
                if take:
                    break
                if want in m.group(1):
                    take, name = True, m.group(1)
                continue
            if take:
                lines.append(ln)
        if take:
            return name, lines
    raise SystemExit(f"no gfx950 kernel matching {want!r} in {so}")


def analyse(lines):
    insts = []  # (index, mnemonic, branch target label or None)
    label_at = {}
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(L\d+)>:", ln)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+(\w+)\s*(.*?)\s*//", ln)
        if not m:
            continue
        mn, ops = m.group(1), m.group(2)
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = re.search(r"(L\d+)", ops)
            tgt = t.group(1) if t else None
        insts.append((mn, tgt))
    in_loop = [False] * len(insts)
    for i, (mn, tgt) in enumerate(insts):
        if tgt is not None and tgt in label_at and label_at[tgt] <= i:
            for k in range(label_at[tgt], i + 1):
                in_loop[k] = True

    def count(sel):
        v = [mn for k, (mn, _) in enumerate(insts) if sel(k) and mn.startswith("v_")]
        two = [mn for mn in v if TWO_SLOT.match(mn)]
        return {"vector": len(v), "two_slot": len(two),
                "two_slot_share": (len(two) / len(v)) if v else None,
                "scalar": sum(1 for k, (mn, _) in enumerate(insts) if sel(k) and mn.startswith("s_")),
                "lds": sum(1 for k, (mn, _) in enumerate(insts) if sel(k) and mn.startswith("ds_")),
                "vmem": sum(1 for k, (mn, _) in enumerate(insts) if sel(k) and mn.startswith(("global_", "buffer_", "flat_", "scratch_")))}

    return {"whole_kernel": count(lambda k: True), "loop_bodies": count(lambda k: in_loop[k])}


def mix(want, so=None):
    so = so or os.path.join(ROOT, "mini_mcmc_amd", "libmmcmc.so")
    name, lines = kernel_text(so, want)
    res = analyse(lines)
    res["symbol"] = name
    res["double_slot_share"] = res["loop_bodies"]["two_slot_share"]
    res["how"] = ("static: two-slot vector instructions / vector instructions inside the kernel's loop bodies, from the disassembly of the "
                  "code object in libmmcmc.so (tools/isa_mix.py); trip counts are not weighted")
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = None
    if "--so" in sys.argv:
        so = sys.argv[sys.argv.index("--so") + 1]
        args.remove(so)
    r = mix(args[0], so)
    print(json.dumps(r, indent=1))
