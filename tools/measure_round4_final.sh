# Final measurement set of round 4: ROUND_TAG=r4z bash tools/measure_round4_final.sh  (then HERE: summarize_pmc for hmc and mh)
cd $GRAFT_REPO_ROOT
T=${ROUND_TAG:-r4z}
ROUND_TAG=$T bash tools/measure_round4.sh > gpurun_out/${T}_measure.log 2>&1
tail -4 gpurun_out/${T}_measure.log
python tools/nuts_small_d.py 2>/dev/null | grep -v amdgpu > gpurun_out/$T/nuts_small_d.jsonl
python tools/small_kernels.py 2>/dev/null | grep -v amdgpu > gpurun_out/$T/small_kernels.jsonl
python tools/stats_ab.py 400 1000 2>/dev/null | grep -v amdgpu > gpurun_out/$T/stats_ab.log
python tools/experiments/repro_hiprtc_miscompile.py > gpurun_out/$T/repro_hiprtc_miscompile.jsonl 2>&1
cat gpurun_out/$T/repro_hiprtc_miscompile.jsonl
