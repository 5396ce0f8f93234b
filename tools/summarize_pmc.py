"""Summarise the rocprofv3 --pmc passes of a sampling kernel into the JSON files bench.py quotes:
    profiles/<what>_kernel_traffic.json   (tools/pmc_hmc_traffic.sh: WRITE_SIZE, FETCH_SIZE; HBM bytes per launch)
    profiles/<what>_kernel_counters.json  (tools/pmc_sq.sh: SQ instruction / cycle counters; vector instructions per transition)
usage: python tools/summarize_pmc.py <tag> <variant> [hmc|mh]
reads gpurun_out/<tag>_traffic_<what>/{w,f} and gpurun_out/<tag>_sq_<what>/{p1,p2}; copies the CSVs to profiles/<tag>_*.csv.
Every summary carries the round tag, the kernel's name and the sha256 of the kernel's sources (tools/kernel_fingerprint.py):
bench.py quotes a summary only while all three still match what it is timing."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import mix  # noqa: E402
from kernel_fingerprint import sampling_kernel_sources_sha256  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, variant = sys.argv[1], int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "hmc"
C = 65536
NC, ND, D = (400, 50, 3) if what == "hmc" else (1000, 100, 2)
KNAME = "mm_run_split_kernel" if variant == 5 else "mm_run_kernel"
# the mangled prefix of the instance the probe launches (mm_target<float, kind, D>, sampler, L)
MANGLED = {"hmc": "_Z19mm_run_split_kernelIf9mm_targetIfLi4ELi3EELi1ELi10E", "mh": "_Z19mm_run_split_kernelIf9mm_targetIfLi0ELi2EELi0ELi0E"}[what]
FP = sampling_kernel_sources_sha256()


def read(path):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(path)):
        if KNAME in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    return {k: sum(v) / len(v) for k, v in agg.items()}, meta


w = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_traffic_{what}", "w", "**", "*counter_collection.csv"), recursive=True)
f = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_traffic_{what}", "f", "**", "*counter_collection.csv"), recursive=True)
if w and f:
    wv, meta = read(w[0])
    fv, _ = read(f[0])
    shutil.copy(w[0], os.path.join(ROOT, "profiles", f"{tag}_{what}_pmc_write_size.csv"))
    shutil.copy(f[0], os.path.join(ROOT, "profiles", f"{tag}_{what}_pmc_fetch_size.csv"))
    alg = C * D * 4 * (NC + 2)
    hbm = wv["WRITE_SIZE"] * 1024 + 2 * fv["FETCH_SIZE"] * 1024
    json.dump({
        "kernel": meta["Kernel_Name"], "variant": variant, "round": tag, "sources_sha256": FP,
        "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/pmc_hmc_traffic.sh) on `python3 tools/pmc_probe.py {what} collect` "
                  f"= run({NC}, {ND}) of 65536 chains, no accept counters; profiles/{tag}_{what}_pmc_write_size.csv, {tag}_{what}_pmc_fetch_size.csv",
        "WRITE_SIZE_KB": wv["WRITE_SIZE"], "FETCH_SIZE_KB_raw": fv["FETCH_SIZE"],
        "corrections": "MI355X_MICROARCH.md HBM section: counters are KiB; FETCH_SIZE reads 1/2 of the bytes of a wide coalesced read on gfx950 -> doubled; "
                       "WRITE_SIZE is exact for 16-byte-per-lane streaming stores (the tile flush uses global_store_dwordx4)",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
    }, open(os.path.join(ROOT, "profiles", f"{what}_kernel_traffic.json"), "w"), indent=1)
    print("traffic", hbm / alg)
p1 = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_{what}", "p1", "**", "*counter_collection.csv"), recursive=True)
p2 = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_{what}", "p2", "**", "*counter_collection.csv"), recursive=True)
if p1 and p2:
    a, meta = read(p1[0])
    b, _ = read(p2[0])
    a.update(b)
    shutil.copy(p1[0], os.path.join(ROOT, "profiles", f"{tag}_sq_{what}_p1.csv"))
    shutil.copy(p2[0], os.path.join(ROOT, "profiles", f"{tag}_sq_{what}_p2.csv"))
    transitions = 1024 * (NC + ND)  # per 64 chains
    waves = a["SQ_WAVES"]
    isa = mix(MANGLED)
    json.dump({
        "kernel": meta["Kernel_Name"], "variant": variant, "round": tag, "sources_sha256": FP,
        "source": f"rocprofv3 --pmc SQ_* (two passes, tools/pmc_sq.sh) on `python3 tools/pmc_probe.py {what} collect`; profiles/{tag}_sq_{what}_p1.csv, _p2.csv",
        "waves": waves, "vgpr": int(meta["VGPR_Count"]), "lds_bytes_per_workgroup": int(meta["LDS_Block_Size"]),
        "valu_instructions_per_transition_of_64_chains": a["SQ_INSTS_VALU"] / transitions,
        "salu_instructions_per_transition_of_64_chains": a["SQ_INSTS_SALU"] / transitions,
        "lds_instructions_per_transition_of_64_chains": a["SQ_INSTS_LDS"] / transitions,
        "double_slot_share": isa["double_slot_share"], "double_slot_share_how": isa["how"], "isa_mix": isa,
        "wave_quad_cycles": a["SQ_WAVE_CYCLES"], "valu_active_quad_cycles": a["SQ_ACTIVE_INST_VALU"],
        "valu_active_over_wave_cycles": a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"],
        "wait_any_over_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
        "wait_inst_any_over_wave_cycles": a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
        "busy_cycles": a.get("SQ_BUSY_CYCLES"),
        "note": "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); one VALU "
                "instruction holds its wave's issue for one quad-cycle",
    }, open(os.path.join(ROOT, "profiles", f"{what}_kernel_counters.json"), "w"), indent=1)
    print("counters ok", a["SQ_INSTS_VALU"] / transitions, "double-slot share", isa["double_slot_share"])
