cd $GRAFT_REPO_ROOT
export MMCMC_CAPTURE_DIR=$GRAFT_REPO_ROOT/gpurun_out/r6zq_units
rm -rf $MMCMC_CAPTURE_DIR; mkdir -p $MMCMC_CAPTURE_DIR
PATH=$GRAFT_REPO_ROOT/tools/experiments/hipcc_capture:$PATH timeout 900 python tools/experiments/repro_hiprtc_miscompile.py > gpurun_out/r6zq_repro.jsonl 2>&1 < /dev/null
cat gpurun_out/r6zq_repro.jsonl
ls $MMCMC_CAPTURE_DIR
for d in $MMCMC_CAPTURE_DIR/*; do rm -f $d/*.hsaco; grep -o "MM_USER_DIM=[0-9]*" $d/cmdline.txt; done
du -sh $MMCMC_CAPTURE_DIR
