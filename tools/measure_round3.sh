# Round-3 measurement set (on the GPU box): ROUND_TAG=r3a bash tools/measure_round3.sh
# -> gpurun_out/$ROUND_TAG/{bench.json, bench_20.json, bench_prof.json, prof_bench/*kernel_stats.csv, nuts_small_d.jsonl, wide_hmc_timing.jsonl,
#    stats_ab.log, pmc_stats_*}
cd $GRAFT_REPO_ROOT
T=${ROUND_TAG:-r3a}
O=gpurun_out/$T
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
python tools/stats_ab.py 400 1000 2>/dev/null | grep -v amdgpu > $O/stats_ab.log
python tools/nuts_small_d.py 2>/dev/null | grep -v amdgpu > $O/nuts_small_d.jsonl
python tools/wide_hmc_timing.py 2>/dev/null | grep -v amdgpu > $O/wide_hmc_timing.jsonl
python tools/small_kernels.py 2>/dev/null | grep -v amdgpu > $O/small_kernels.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
bash tools/pmc_stats.sh 400 auto $T > $O/pmc_stats_400.log 2>&1
bash tools/pmc_stats.sh 1000 auto $T > $O/pmc_stats_1000.log 2>&1
python3 - <<PY
import json
for f in ("bench.json", "bench_20.json", "bench_prof.json"):
    try:
        j = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, "value", j["value"] / 1e9, "ms", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "issue frac", j["roofline"]["frac"], "hbm frac", j["roofline"]["hbm"]["frac"], "stats_ms", j["stats_ms"], "ess/s", j["ess_per_s"] / 1e6)
    except Exception as e:
        print(f, "unreadable", e)
PY
