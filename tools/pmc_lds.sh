# LDS conflict counters of the sampling kernels (separate rocprofv3 --pmc pass, no tracing domains):
#   bash tools/pmc_lds.sh <tag> hmc|mh  -> gpurun_out/<tag>_lds_<what>/p3/..._counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
WHAT=${2:-mh}
O=$R/gpurun_out/${TAG}_lds_${WHAT}
mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py $WHAT collect > $O/p3.log 2>&1
find $O -name "*counter_collection.csv"
tail -2 $O/p3.log
