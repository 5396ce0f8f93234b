# Round 4: the MH split kernel (config 2) after the three-input xor -- role layouts again (QP: noise pairs the transition wave
# draws itself; RB: ring half), where the roles wait (MM_SPLIT_PROFILE), and the Philox4x32-7 A/B (timing only: the stream is
# ten rounds).  bash tools/experiments/split_mh_round4.sh
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_MH_RB=8 -DMM_PROBE_MH_QP=0" "-DMM_PROBE_MH_RB=8 -DMM_PROBE_MH_QP=1" "-DMM_PROBE_MH_RB=16 -DMM_PROBE_MH_QP=2" "-DMM_PROBE_MH_RB=16 -DMM_PROBE_MH_QP=0" "-DMM_PHILOX_ROUNDS=7" "-DMM_SPLIT_PROFILE" "-DMM_SPLIT_PROFILE -DMM_PROBE_MH_RB=8 -DMM_PROBE_MH_QP=0"; do
  $B $v -o /tmp/sp_r4_$i 2>/dev/null &
  i=$((i+1))
done
wait
for r in 1 2; do i=0; for v in "default (RB=8 QP=2)" "RB=8 QP=0" "RB=8 QP=1" "RB=16 QP=2" "RB=16 QP=0" "Philox4x32-7 (timing only)"; do echo "$v: $(/tmp/sp_r4_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7}' | tr '\n' ' ')"; i=$((i+1)); done; done
echo "--- role profile, default layout"; /tmp/sp_r4_6 2>&1 | grep -A1 "split"
echo "--- role profile, RB=8 QP=0"; /tmp/sp_r4_7 2>&1 | grep -A1 "split"
