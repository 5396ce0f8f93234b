cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "nuts" > gpurun_out/r6zl_nuts_tests.log 2>&1 < /dev/null; tail -2 gpurun_out/r6zl_nuts_tests.log
timeout 400 python tools/fuzz_variants.py 120 l > gpurun_out/r6zl_fuzz_lgq.log 2>&1 < /dev/null; grep "lane-group" gpurun_out/r6zl_fuzz_lgq.log
timeout 300 python tools/nuts_cfg5_timing.py 200 100 4 2>/dev/null < /dev/null | cut -c1-130 > gpurun_out/r6zl_nuts_cfg5.jsonl; cat gpurun_out/r6zl_nuts_cfg5.jsonl
timeout 300 python tools/nuts_cfg5_timing.py 500 500 1 2>/dev/null < /dev/null | cut -c1-130 >> gpurun_out/r6zl_nuts_cfg5.jsonl; tail -1 gpurun_out/r6zl_nuts_cfg5.jsonl
