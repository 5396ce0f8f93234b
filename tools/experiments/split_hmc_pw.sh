# Round 6: the HMC split kernel (config 3) in the PRODUCT's layout (tiles written out by a noise wave, ring half 8) with ONE or
# TWO pairs per workgroup and the transition wave's priority raised, against the product's four pairs.
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 -DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=8 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_PW=1 -DMM_PROBE_PW_HMC=1 -DMM_SPLIT_PRIO_Q=3" "-DMM_PROBE_PW=1 -DMM_PROBE_PW_HMC=1 -DMM_SPLIT_PRIO_Q=1" "-DMM_PROBE_PW=1 -DMM_PROBE_PW_HMC=1" "-DMM_PROBE_PW=2 -DMM_PROBE_PW_HMC=1 -DMM_SPLIT_PRIO_Q=3" "-DMM_SPLIT_PRIO_Q=3"; do
  $B $v -o /tmp/sp_h_$i 2>/tmp/sp_h_$i.err &
  i=$((i+1))
done
wait
for r in 1 2 3 4; do i=0; for v in "PW=4 (product layout)" "PW=1 prio 3" "PW=1 prio 1" "PW=1" "PW=2 prio 3" "PW=4 prio 3"; do echo "$v: $(timeout 60 /tmp/sp_h_$i 2>&1 < /dev/null | grep "hmc cfg3 *split" | awk '{print $4, $5, $6, $7, $9}' | tr '\n' ' ')"; i=$((i+1)); done; done
