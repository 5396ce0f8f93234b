cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "long_chains or refuses_shapes or split_rhat or stats" > gpurun_out/r6p_stats_tests.log 2>&1; tail -5 gpurun_out/r6p_stats_tests.log
python tools/stats_long_timing.py > gpurun_out/r6p_stats_long_timing.log 2>&1; cat gpurun_out/r6p_stats_long_timing.log
