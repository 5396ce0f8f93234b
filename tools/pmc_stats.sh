# rocprofv3 passes over the split-R-hat / ESS kernels ([65536, N, 3] f32 sample; N = $1, default 400; kernel = $2, default auto):
# kernel trace + SQ counters + LDS counters + HBM traffic, each in its own run (gpurun refuses mixed --pmc / trace runs)
# usage (on the GPU box): bash tools/pmc_stats.sh [n] [kernel] [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-400}
K=${2:-auto}
T=${3:-r3}
O=$R/gpurun_out/pmc_stats_${T}_${N}_${K}
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/tools/pmc_probe.py stats $N $K > $O/w.log 2>&1
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "half_chain" in k or "fft" in k or "stats_tail" in k:
            a = acc.setdefault(k, {"launches": 0})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, a in acc.items():
        print(f.split("/")[-2], k, {c: f"{v:.5g}" for c, v in a.items()})
for f in sorted(glob.glob("$O/kt/*kernel_stats.csv")):
    print(open(f).read())
PY
