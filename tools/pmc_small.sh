# SQ counters of the small kernels (integer-state MH, Gibbs mixture, tracker) at 65 536 chains: what bounds them?
# usage (on the GPU box): bash tools/pmc_small.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r3}
O=$R/gpurun_out/pmc_small_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/small_kernels.py > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/small_kernels.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/small_kernels.py > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/small_kernels.py > $O/p3.log 2>&1
python3 $R/tools/pmc_sum.py $O discrete gibbs tracker
python3 - <<PY
import csv
csv.field_size_limit(1 << 30)
for r in csv.DictReader(open("$O/kt/kt_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("discrete", "gibbs", "tracker")):
        print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
