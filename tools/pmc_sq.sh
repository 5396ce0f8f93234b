# SQ instruction / cycle counters of the sampling kernels (separate rocprofv3 --pmc passes, no tracing domains):
#   bash tools/pmc_sq.sh <tag> hmc|mh   -> gpurun_out/<tag>_sq_<what>/{p1,p2}/..._counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
WHAT=${2:-hmc}
O=$R/gpurun_out/${TAG}_sq_${WHAT}
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py $WHAT collect > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py $WHAT collect > $O/p2.log 2>&1
find $O -name "*counter_collection.csv"
tail -2 $O/p1.log $O/p2.log
