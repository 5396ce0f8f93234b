# Round 6: ring halves of the MH split kernel under the PAIRED stream (one Philox block per two transitions: a batch of RB
# transitions is RB / 2 noise pairs dealt to 3 noise waves -- RB = 8: 2 + 1 + 1, RB = 6: 1 + 1 + 1, RB = 12: 2 + 2 + 2).
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_MH_RB=4" "-DMM_PROBE_MH_RB=6" "-DMM_PROBE_MH_RB=10" "-DMM_PROBE_MH_RB=12" "-DMM_PROBE_MH_RB=14"; do
  $B $v -o /tmp/sp_l$i 2>/tmp/sp_l$i.err &
  i=$((i+1))
done
wait
for r in 1 2 3; do i=0; for v in "default (RB=8)" "RB=4" "RB=6" "RB=10" "RB=12" "RB=14"; do echo "$v: $(/tmp/sp_l$i 2>&1 | grep "mh cfg2 *split" | awk '{print $4, $5, $6, $7, $9}' | tr '\n' ' ')"; i=$((i+1)); done; done
grep -l "error" /tmp/sp_l*.err | head
