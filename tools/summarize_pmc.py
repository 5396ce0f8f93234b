"""Summarise the rocprofv3 --pmc passes of the headline kernel into the two JSON files bench.py quotes:
    profiles/hmc_kernel_traffic.json   (tools/pmc_hmc_traffic.sh: WRITE_SIZE, FETCH_SIZE; HBM bytes per launch)
    profiles/hmc_kernel_counters.json  (tools/pmc_sq.sh: SQ instruction / cycle counters; issue slots per transition)
usage: python tools/summarize_pmc.py <tag> <variant>   (reads gpurun_out/pmc_hmc and gpurun_out/<tag>_sq_hmc, copies the
CSVs to profiles/<tag>_pmc_*.csv)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, variant = sys.argv[1], int(sys.argv[2])
C, NC, ND, D = 65536, 400, 50, 3
KNAME = "mm_run_split_kernel" if variant == 5 else "mm_run_kernel"


def read(path):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(path)):
        if KNAME in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    return {k: sum(v) / len(v) for k, v in agg.items()}, meta


out = {}
w = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_hmc", "w", "*counter_collection.csv"))
f = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_hmc", "f", "*counter_collection.csv"))
if w and f:
    wv, meta = read(w[0])
    fv, _ = read(f[0])
    shutil.copy(w[0], os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.csv"))
    shutil.copy(f[0], os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.csv"))
    alg = C * D * 4 * (NC + 2)
    hbm = wv["WRITE_SIZE"] * 1024 + 2 * fv["FETCH_SIZE"] * 1024
    json.dump({
        "kernel": meta["Kernel_Name"], "variant": variant,
        "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, tools/pmc_hmc_traffic.sh) on `python3 tools/pmc_probe.py hmc collect` "
                  f"= HMC::run(400, 50) of 65536 chains, no accept counters; profiles/{tag}_pmc_write_size.csv, {tag}_pmc_fetch_size.csv",
        "WRITE_SIZE_KB": wv["WRITE_SIZE"], "FETCH_SIZE_KB_raw": fv["FETCH_SIZE"],
        "corrections": "MI355X_MICROARCH.md HBM section: counters are KiB; FETCH_SIZE reads 1/2 of the bytes of a wide coalesced read on gfx950 -> doubled; "
                       "WRITE_SIZE is exact for 16-byte-per-lane streaming stores (the tile flush uses global_store_dwordx4)",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
    }, open(os.path.join(ROOT, "profiles", "hmc_kernel_traffic.json"), "w"), indent=1)
    print("traffic", hbm / alg)
p1 = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_hmc", "p1", "*counter_collection.csv"))
p2 = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_hmc", "p2", "*counter_collection.csv"))
if p1 and p2:
    a, meta = read(p1[0])
    b, _ = read(p2[0])
    a.update(b)
    shutil.copy(p1[0], os.path.join(ROOT, "profiles", f"{tag}_sq_hmc_p1.csv"))
    shutil.copy(p2[0], os.path.join(ROOT, "profiles", f"{tag}_sq_hmc_p2.csv"))
    transitions = 1024 * (NC + ND)  # per 64 chains
    waves = a["SQ_WAVES"]
    json.dump({
        "kernel": meta["Kernel_Name"], "variant": variant,
        "source": f"rocprofv3 --pmc SQ_* (two passes, tools/pmc_sq.sh) on `python3 tools/pmc_probe.py hmc collect`; profiles/{tag}_sq_hmc_p1.csv, _p2.csv",
        "waves": waves, "vgpr": int(meta["VGPR_Count"]), "lds_bytes_per_workgroup": int(meta["LDS_Block_Size"]),
        "valu_instructions_per_transition_of_64_chains": a["SQ_INSTS_VALU"] / transitions,
        "salu_instructions_per_transition_of_64_chains": a["SQ_INSTS_SALU"] / transitions,
        "lds_instructions_per_transition_of_64_chains": a["SQ_INSTS_LDS"] / transitions,
        "wave_quad_cycles": a["SQ_WAVE_CYCLES"], "valu_active_quad_cycles": a["SQ_ACTIVE_INST_VALU"],
        "valu_active_over_wave_cycles": a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"],
        "wait_any_over_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
        "wait_inst_any_over_wave_cycles": a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
        "busy_cycles": a.get("SQ_BUSY_CYCLES"),
        "note": "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); one VALU "
                "instruction holds its wave's issue for one quad-cycle",
    }, open(os.path.join(ROOT, "profiles", "hmc_kernel_counters.json"), "w"), indent=1)
    print("counters ok", a["SQ_INSTS_VALU"] / transitions)
