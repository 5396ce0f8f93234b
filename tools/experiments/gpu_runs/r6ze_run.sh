cd $GRAFT_REPO_ROOT
VARIANTS_FILE=tools/experiments/nuts_lg_r6_variants.txt timeout 600 bash tools/experiments/nuts_lg_fast_variants.sh > gpurun_out/r6ze_nuts_deep_early_probe.log 2>&1 < /dev/null
cat gpurun_out/r6ze_nuts_deep_early_probe.log
