cd $GRAFT_REPO_ROOT
VARIANTS_FILE=tools/experiments/nuts_lg_r6_variants.txt bash tools/experiments/nuts_lg_fast_variants.sh > gpurun_out/r6l_nuts_pingpong_probe.log 2>&1
cat gpurun_out/r6l_nuts_pingpong_probe.log
python -m pytest tests -m gpu -x -q -k "nuts_lane_group or config5 or nuts_lg" > gpurun_out/r6l_nuts_tests.log 2>&1; tail -3 gpurun_out/r6l_nuts_tests.log
python tools/nuts_cfg5_timing.py 200 100 3 > gpurun_out/r6l_nuts_cfg5.jsonl 2>&1; cat gpurun_out/r6l_nuts_cfg5.jsonl
