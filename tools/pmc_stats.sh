# rocprofv3 PMC passes over the split-R-hat / ESS kernels ([65536, 400, 3] f32 sample)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stats
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py stats 0 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py stats 0 > $O/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py stats 0 > $O/p3.log 2>&1
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    acc = {}
    n = 0
    for r in csv.DictReader(open(f)):
        if "half_chain" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n += 1
    print(f.split("/")[-2], n, {k: f"{v:.4g}" for k, v in acc.items()})
PY
tail -2 $O/p3.log
