# rocprofv3 PMC passes over the one-chain-per-lane NUTS kernel (RosenbrockND(3), 65 536 chains, scalar mode 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${NUTS_VARIANT:-4}
O=$R/gpurun_out/pmc_nuts_small_v$V
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts3 1 $V > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts3 1 $V > $O/p2.log 2>&1
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "async_kernel" in r["Kernel_Name"] or "nuts_run_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(f.split("/")[-2], {k: f"{v:.4g}" for k, v in acc.items()})
PY
