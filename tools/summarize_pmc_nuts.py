"""Summarise the SQ counter passes of config 5's scheduler kernel (tools/pmc_nuts.sh <tag>, OCCS=1) into
profiles/nuts5_kernel_counters.json, which bench.py quotes beside side.config5_nuts.f64_mfma_frac (only while its round tag is
the bench's).   python tools/summarize_pmc_nuts.py <tag>
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
agg = collections.defaultdict(float)
meta = None
for i in (1, 2, 3):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_sq_nuts", "occ1", f"p{i}", "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        raise SystemExit(f"pass p{i} missing")
    shutil.copy(fs[0], os.path.join(ROOT, "profiles", f"{tag}_sq_nuts_occ1_p{i}.csv"))
    for r in csv.DictReader(open(fs[0])):
        if "mm_nuts_lgq_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            meta = r
wave_cycles = 4.0 * agg["SQ_WAVE_CYCLES"]
valu_active = 4.0 * agg["SQ_ACTIVE_INST_VALU"]
mfma_busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"]
# leaf iterations of a wave: every one issues D/16 * D/4 = 16 MFMAs at D = 32 (the begin kernel's logp adds 16 per transition and chain group)
leaf_iters = agg["SQ_INSTS_MFMA"] / 16.0
out = {
    "kernel": meta["Kernel_Name"], "round": tag,
    "source": f"rocprofv3 --pmc SQ_* (three passes, tools/pmc_nuts.sh {tag}, one wave per SIMD) on `python3 tools/pmc_probe.py nuts5 65536 200 100`; "
              f"profiles/{tag}_sq_nuts_occ1_p1.csv, _p2.csv, _p3.csv",
    "vgpr": int(meta["VGPR_Count"]), "agpr": int(meta["Accum_VGPR_Count"]),
    "wave_cycles": wave_cycles, "valu_active_cycles": valu_active, "mfma_busy_cycles": mfma_busy,
    "valu_active_over_wave_cycles": valu_active / wave_cycles,
    "mfma_busy_over_wave_cycles": mfma_busy / wave_cycles,
    "wait_any_over_wave_cycles": 4.0 * agg["SQ_WAIT_ANY"] / wave_cycles,
    "wait_inst_any_over_wave_cycles": 4.0 * agg["SQ_WAIT_INST_ANY"] / wave_cycles,
    "scalar_active_over_wave_cycles": 4.0 * agg["SQ_ACTIVE_INST_SCA"] / wave_cycles,
    "leaf_iterations": leaf_iters,
    "wave_cycles_per_leaf_iteration": wave_cycles / leaf_iters,
    "valu_instructions_per_leaf_iteration": (agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"]) / leaf_iters,
    "salu_instructions_per_leaf_iteration": agg["SQ_INSTS_SALU"] / leaf_iters,
    "lds_instructions_per_leaf_iteration": agg["SQ_INSTS_LDS"] / leaf_iters,
    "vmem_instructions_per_leaf_iteration": agg["SQ_INSTS_VMEM"] / leaf_iters,
    "branch_instructions_per_leaf_iteration": agg["SQ_INSTS_BRANCH"] / leaf_iters,
}
# issue: the share of the wave's cycles in which its SIMD issues or executes one of its vector instructions -- the matrix core's busy
# cycles plus the other vector instructions' active cycles (SQ_ACTIVE_INST_VALU covers both kinds: the MFMA's part is its busy
# time where that is the larger figure)
out["issue_frac"] = max(out["valu_active_over_wave_cycles"], out["mfma_busy_over_wave_cycles"] +
                        max(0.0, out["valu_active_over_wave_cycles"] - agg["SQ_INSTS_MFMA"] * 4.0 / wave_cycles))
out["issue_frac_how"] = ("max(valu_active, mfma_busy + (valu_active - 4 quad-cycles-of-issue per MFMA)) over wave cycles: an MFMA is ACTIVE for its issue "
                         "slot only but keeps the SIMD's vector pipe BUSY for 64 cycles (tools/f64_rate.hip)")
json.dump(out, open(os.path.join(ROOT, "profiles", "nuts5_kernel_counters.json"), "w"), indent=1)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items() if k not in ("source", "kernel", "issue_frac_how")})
