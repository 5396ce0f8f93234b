# Kernel times of the tracker (rocprofv3 --kernel-trace) for every library build copied to tools/experiments/tk/lib_*.so
# (e.g. cp mini_mcmc_amd/libmmcmc.so tools/experiments/tk/lib_new.so), on an HMC sample [65536, 400, 3]:
#   bash tools/experiments/tk_prof.sh      (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in $R/tools/experiments/tk/lib_*.so; do
  n=$(basename $l .so)
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tk_$n -o kt --output-format csv -- python3 $R/tools/experiments/tk_child.py $l > /dev/null 2>&1
  echo "== $n"; grep -h "tracker" $R/gpurun_out/tk_$n/kt_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
