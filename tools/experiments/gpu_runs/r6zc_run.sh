cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q -k "long_chains" > gpurun_out/r6zc_stats_tests.log 2>&1 < /dev/null; tail -2 gpurun_out/r6zc_stats_tests.log
timeout 600 python tools/stats_long_timing.py > gpurun_out/r6zc_stats_long_timing.log 2>&1 < /dev/null; cat gpurun_out/r6zc_stats_long_timing.log
