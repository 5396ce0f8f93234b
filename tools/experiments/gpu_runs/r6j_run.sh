cd $GRAFT_REPO_ROOT
VARIANTS_FILE=tools/experiments/nuts_lg_r6_variants.txt bash tools/experiments/nuts_lg_fast_variants.sh > gpurun_out/r6j_nuts_check_form3_probe.log 2>&1
cat gpurun_out/r6j_nuts_check_form3_probe.log
