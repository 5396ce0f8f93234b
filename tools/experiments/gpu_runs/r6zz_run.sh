cd $GRAFT_REPO_ROOT
ROUND_TAG=r6zz timeout 2700 bash tools/measure_round6.sh > gpurun_out/r6zz_measure.log 2>&1 < /dev/null
tail -40 gpurun_out/r6zz_measure.log
