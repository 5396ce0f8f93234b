# SQ counters of the persistent NUTS scheduler (config 5) at one and two waves per SIMD:
#   bash tools/pmc_nuts.sh <tag>  -> gpurun_out/<tag>_sq_nuts/occ{1,2}/{p1,p2,p3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
for OCC in ${OCCS:-1 2}; do
export MMCMC_LGQ_OCC=$OCC
O=$R/gpurun_out/${TAG}_sq_nuts/occ$OCC
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 65536 200 100 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 65536 200 100 > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 65536 200 100 > $O/p3.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for occ in "${OCCS:-1 2}".split():
    for f in sorted(glob.glob("$R/gpurun_out/${TAG}_sq_nuts/occ%s/p*/*counter_collection.csv" % occ)):
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "lgq" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
        print("occ", occ, {k: "%.4g" % v for k, v in agg.items()})
PY
