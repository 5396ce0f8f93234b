# dynamic VALU instruction mix of the persistent NUTS scheduler (config 5): bash tools/pmc_nuts_mix.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
O=$R/gpurun_out/${TAG}_mix_nuts
mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 65536 200 100 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 65536 200 100 > $O/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "lgq" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print({k: "%.4g" % v for k, v in agg.items()})
PY
tail -2 $O/p1.log $O/p2.log
