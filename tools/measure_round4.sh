# Round-4 measurement set (on the GPU box): ROUND_TAG=r4a bash tools/measure_round4.sh
# -> gpurun_out/$ROUND_TAG/{bench.json, bench_20.json, bench_prof.json, prof_bench/*kernel_stats.csv, ...} and the PMC passes
#    gpurun_out/${ROUND_TAG}_{sq,traffic}_{hmc,mh}; then HERE: python tools/summarize_pmc.py $ROUND_TAG 5 hmc|mh
cd $GRAFT_REPO_ROOT
T=${ROUND_TAG:-r4a}
O=gpurun_out/$T
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
for W in hmc mh; do
  bash tools/pmc_sq.sh $T $W > $O/pmc_sq_$W.log 2>&1
  bash tools/pmc_hmc_traffic.sh $T $W > $O/pmc_traffic_$W.log 2>&1
done
python3 - <<PY
import json
for f in ("bench.json", "bench_20.json", "bench_prof.json"):
    try:
        j = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, "value", j["value"] / 1e9, "ms", j["ms_per_step"], "kernel_ms", r["kernel_ms"], "hbm frac", r["frac"], "fp32", r["fp32"]["frac"],
              "issue", (r.get("issue") or {}).get("frac"), "stats_ms", j["stats_ms"], "cpu", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
