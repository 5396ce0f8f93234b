# Round 5: dynamic instruction counts of the floor probe's kernels (tools/nuts_lg_floor_probe.hip): vector / scalar / LDS / VMEM
# instructions and wait cycles per leaf iteration of a wave, floor against the product's doubling in its best case.
#   FLAGS="-D..." bash tools/experiments/nuts_lg_floor_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
O=$R/gpurun_out/r5_floor_pmc_$TAG
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 $FLAGS -I $R/mini_mcmc_amd/csrc $R/tools/nuts_lg_floor_probe.hip -o /tmp/lg_floor_pmc 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- /tmp/lg_floor_pmc > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH -d $O/p2 -o p2 --output-format csv -- /tmp/lg_floor_pmc > $O/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "floor%s" % k[k.find("ELi") + 3] if "floor_kernel" in k else ("product_doubling" if "product_doubling" in k else None)
        if name:
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES",):
                n[name] += 1
for name in sorted(agg):
    a = agg[name]
    launches = a["SQ_WAVES"] / 1024.0
    # leaf iterations per wave: floor 40000 per launch; product 38400 per launch
    it = (40000.0 if name.startswith("floor") else 38400.0)
    per = lambda c: a[c] / a["SQ_WAVES"] / it if a["SQ_WAVES"] else 0
    print(name, "launches %.0f" % launches, {c: round(per(c), 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH")},
          {c: round(a[c] / max(a["SQ_WAVES"], 1) / it * 4, 0) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA")})
PY
