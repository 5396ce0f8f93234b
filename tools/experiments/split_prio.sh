# A/B of wave priorities in the split kernel (tools/split_probe.hip), on the GPU box: bash tools/experiments/split_prio.sh
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
$B -o /tmp/sp_base 2>/dev/null & 
$B -DMM_SPLIT_PRIO_N=1 -o /tmp/sp_n1 2>/dev/null &
$B -DMM_SPLIT_PRIO_N=3 -o /tmp/sp_n3 2>/dev/null &
$B -DMM_SPLIT_PRIO_N=4 -o /tmp/sp_n4 2>/dev/null &
$B -DMM_SPLIT_PRIO_N=5 -o /tmp/sp_n5 2>/dev/null &
$B -DMM_SPLIT_PRIO_N=1 -DMM_SPLIT_PRIO_Q=1 -o /tmp/sp_n1q1 2>/dev/null &
wait
for r in 1 2 3; do for v in base n1 n3 n4 n5 n1q1; do echo "== $v $(/tmp/sp_$v 2>&1 | grep -i "split" | grep "with output" | awk '{print $1, $6}' | tr '\n' ' ')"; done; done
