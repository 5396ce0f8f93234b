# bench.py at 300 and at the default 20 steps, one line each, then the HMC / MH parity tests that pin the kernels' bits:
#   bash tools/experiments/ab_bench_hmc.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-ab}
python bench.py --steps 300 --warmup 20 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py > gpurun_out/${T}_bench_20.json 2>/dev/null
python - <<PY
import json
for f in ["gpurun_out/${T}_bench.json", "gpurun_out/${T}_bench_20.json"]:
    d = json.load(open(f))
    print(f, "%.2f G samples/s" % (d["value"] / 1e9), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"],
          "hbm frac %.4f" % d["roofline"]["frac"], "mh %.4f" % d["side"]["config2_mh"]["kernel_ms"])
PY
timeout 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference_order.py tests/test_user_target.py -m gpu -x -q -k "hmc or config3 or variants or split or partition or user" > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
