cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6zv_gpu_suite.log 2>&1 < /dev/null; grep -E "passed|failed|error" gpurun_out/r6zv_gpu_suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6zv_smoke.log 2>&1 < /dev/null; tail -4 gpurun_out/r6zv_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r6zv_bench_20.json 2>/dev/null < /dev/null; cut -c1-300 gpurun_out/r6zv_bench_20.json
