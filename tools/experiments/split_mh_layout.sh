# role layouts of the MH split kernel (config 2) in tools/split_probe.hip: pairs drawn by the transition wave (QP), ring half (RB)
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_MH_RB=6 -DMM_PROBE_MH_QP=0" "-DMM_PROBE_MH_RB=12 -DMM_PROBE_MH_QP=0" "-DMM_PROBE_MH_RB=8 -DMM_PROBE_MH_QP=0" "-DMM_PROBE_MH_RB=8 -DMM_PROBE_MH_QP=1" "-DMM_PROBE_MH_RB=12 -DMM_PROBE_MH_QP=1" "-DMM_PROBE_MH_RB=16 -DMM_PROBE_MH_QP=2"; do
  $B $v -o /tmp/sp_l$i 2>/dev/null &
  i=$((i+1))
done
wait
for r in 1 2; do i=0; for v in "default" "RB=6 QP=0" "RB=12 QP=0" "RB=8 QP=0" "RB=8 QP=1" "RB=12 QP=1" "RB=16 QP=2"; do echo "$v: $(/tmp/sp_l$i 2>&1 | grep "mh cfg2 *split" | awk '{print $4, $5, $6, $7}' | tr '\n' ' ')"; i=$((i+1)); done; done
