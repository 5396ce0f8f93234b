# HBM traffic of a sampling kernel: separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section)
#   bash tools/pmc_hmc_traffic.sh <tag> hmc|mh  -> gpurun_out/<tag>_traffic_<what>/{w,f}/..._counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
WHAT=${2:-hmc}
O=$R/gpurun_out/${TAG}_traffic_${WHAT}
mkdir -p $O
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/tools/pmc_probe.py $WHAT collect > $O/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/tools/pmc_probe.py $WHAT collect > $O/f.log 2>&1
find $O -name "*counter_collection.csv"
