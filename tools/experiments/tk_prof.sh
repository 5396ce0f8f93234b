cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in $R/tools/experiments/tk/lib_*.so; do
  n=$(basename $l .so)
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tk_$n -o kt --output-format csv -- python3 $R/tools/experiments/tk_child.py $l > /dev/null 2>&1
  echo "== $n"; grep -h "tracker" $R/gpurun_out/tk_$n/kt_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
