# HBM traffic of the HMC kernel: separate rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_hmc
mkdir -p $O
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/tools/pmc_probe.py hmc collect > $O/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/tools/pmc_probe.py hmc collect > $O/f.log 2>&1
find $O -name "*counter_collection.csv"
