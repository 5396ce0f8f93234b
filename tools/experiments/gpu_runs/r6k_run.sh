cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "nuts" > gpurun_out/r6k_nuts_tests.log 2>&1; tail -3 gpurun_out/r6k_nuts_tests.log
python tools/fuzz_variants.py 120 l > gpurun_out/r6k_fuzz_lgq.log 2>&1; tail -3 gpurun_out/r6k_fuzz_lgq.log
python tools/nuts_cfg5_timing.py 200 100 3 > gpurun_out/r6k_nuts_cfg5.jsonl 2>&1; cat gpurun_out/r6k_nuts_cfg5.jsonl
python tools/nuts_cfg5_timing.py 500 500 1 > gpurun_out/r6k_nuts_cfg5_500.jsonl 2>&1; cat gpurun_out/r6k_nuts_cfg5_500.jsonl
