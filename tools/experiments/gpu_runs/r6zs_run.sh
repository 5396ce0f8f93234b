cd $GRAFT_REPO_ROOT
timeout 2400 python tools/experiments/repro_hiprtc_old_tree.py _r4tree > gpurun_out/r6zs_repro_old_tree.jsonl 2>&1 < /dev/null
cut -c1-400 gpurun_out/r6zs_repro_old_tree.jsonl
