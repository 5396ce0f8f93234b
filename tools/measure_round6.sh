# Round-6 measurement set (on the GPU box): ROUND_TAG=r6x bash tools/measure_round6.sh
# -> gpurun_out/$ROUND_TAG/{bench.json, bench_20.json, bench_prof.json, prof_bench/*kernel_stats.csv, group lines, pcie, ...} and the PMC
#    passes gpurun_out/${ROUND_TAG}_{sq,traffic}_{hmc,mh}, ${ROUND_TAG}_sq_nuts; then HERE: python tools/summarize_pmc.py $ROUND_TAG 5 hmc|mh,
#    python tools/summarize_pmc_nuts.py $ROUND_TAG
cd $GRAFT_REPO_ROOT
T=${ROUND_TAG:-r6x}
O=gpurun_out/$T
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
python bench.py --gpus 1 --group --steps 300 --warmup 10 2>/dev/null | grep "^{" > $O/bench_group1.json
MMCMC_BENCH_GROUP_DEVICES=0,0 python bench.py --gpus 2 --group --steps 300 --warmup 10 2>/dev/null | grep "^{" > $O/bench_group_0_0.json
python tools/pcie_inclusive.py 2>/dev/null > $O/pcie_inclusive.json
python tools/nuts_cfg5_timing.py 200 100 2 2>/dev/null > $O/nuts_cfg5.jsonl
python tools/nuts_cfg5_timing.py 500 500 1 2>/dev/null >> $O/nuts_cfg5.jsonl
python tools/stats_timing.py 2>/dev/null | cut -c1-60 > $O/stats_timing.log
timeout 600 python tools/stats_long_timing.py 2>/dev/null < /dev/null | grep -v amdgpu > $O/stats_long_timing.log
for i in 1 2 3; do python tools/time_cfg23.py 2>/dev/null | tail -1; done > $O/time_cfg23.log
python tools/nuts_small_d.py 2>/dev/null | grep -v amdgpu > $O/nuts_small_d.jsonl
python tools/small_kernels.py 2>/dev/null | grep -v amdgpu > $O/small_kernels.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
for W in hmc mh; do
  bash tools/pmc_sq.sh $T $W > $O/pmc_sq_$W.log 2>&1
  bash tools/pmc_hmc_traffic.sh $T $W > $O/pmc_traffic_$W.log 2>&1
done
OCCS=1 bash tools/pmc_nuts.sh $T > $O/pmc_nuts.log 2>&1
python3 - <<PY
import json
for f in ("bench.json", "bench_20.json", "bench_prof.json", "bench_group1.json", "bench_group_0_0.json"):
    try:
        j = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, "value", j["value"] / 1e9, "ms", j["ms_per_step"], "kernel_ms", r.get("kernel_ms"), "hbm frac", r["frac"], "stats_ms", j.get("stats_ms"),
              "cpu", (j.get("cpu_baseline") or {}).get("value"))
        for k, v in (j.get("side") or {}).items():
            if isinstance(v, dict):
                print("   side", k, {kk: vv for kk, vv in v.items() if kk in ("kernel_ms", "hbm_frac", "f64_mfma_frac", "ess_per_s", "ess_per_s_unconverged", "split_rhat_max_conventional", "max_rel_moment_error", "stats_ms", "kernel_ms_back_to_back", "hbm_frac_back_to_back")})
    except Exception as e:
        print(f, "unreadable", e)
PY
