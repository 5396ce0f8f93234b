"""Every gfx950 kernel of libmmcmc.so that uses private (scratch) memory, from the code objects' metadata notes:
    python tools/scratch_scan.py [min bytes, default 1]
A kernel with scratch either indexes a local array at run time or spills registers; the hot ones must have none
(tests/test_codegen.py guards three of them)."""
import os, re, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix

READELF = os.path.join(os.path.dirname(isa_mix.OBJDUMP), "llvm-readelf")
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mini_mcmc_amd", "libmmcmc.so")
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = []
for co in isa_mix.code_objects(so):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    cur_sz = 0
    for ln in notes.splitlines():  # per kernel the metadata lists .private_segment_fixed_size before .symbol
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
        if m:
            cur_sz = int(m.group(1))
        m = re.search(r"\.symbol:\s+(\S+)\.kd", ln)
        if m:
            rows.append((cur_sz, m.group(1)))
names = [r[1] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for (sz, raw), d in sorted(zip(rows, dem), reverse=True):
    if sz >= lo:
        print(sz, re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", ""))[:150])
