cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "long_chains or stats or rhat or ess" > gpurun_out/r6zm_stats_tests.log 2>&1 < /dev/null; tail -3 gpurun_out/r6zm_stats_tests.log
timeout 600 python tools/stats_long_timing.py > gpurun_out/r6zm_stats_long_timing.log 2>&1 < /dev/null; cat gpurun_out/r6zm_stats_long_timing.log
