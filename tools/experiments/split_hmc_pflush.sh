# HMC split kernel: who writes the tiles out?  transition wave (default) vs the least-loaded noise wave (-DMM_SPLIT_HMC_PFLUSH=1),
# with the flushing rank 2 (fewest pairs) or 1.  bash tools/experiments/split_hmc_pflush.sh
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_SPLIT_HMC_PFLUSH=1" "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_SPLIT_FLUSH_RANK=1" "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=4" "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=8"; do
  $B $v -o /tmp/sp_pf_$i 2>/tmp/sp_pf_$i.err &
  i=$((i+1))
done
wait
for r in 1 2 3; do i=0; for v in "default" "PFLUSH rank 2" "PFLUSH rank 1" "PFLUSH RB=4" "PFLUSH RB=8"; do echo "$v: $(/tmp/sp_pf_$i 2>&1 | grep "hmc cfg3 *split" | awk '{print $4, $5, $6, $7, $8, $9, $10}' | tr '\n' ' ')"; i=$((i+1)); done; done
tail -3 /tmp/sp_pf_1.err
