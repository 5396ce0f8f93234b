# HMC split kernel with the noise wave writing the tiles out: ring halves 6 / 8 (three / four noise pairs per batch over three noise waves)
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=8" "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=6" "-DMM_SPLIT_HMC_PFLUSH=1 -DMM_PROBE_HMC_RB=8 -DMM_SPLIT_FLUSH_RANK=0" "-DMM_PROBE_HMC_RB=6"; do
  $B $v -o /tmp/sp_pg_$i 2>/tmp/sp_pg_$i.err &
  i=$((i+1))
done
wait
for r in 1 2 3; do i=0; for v in "PFLUSH RB=8" "PFLUSH RB=6" "PFLUSH RB=8 rank 0" "own flush RB=6"; do echo "$v: $(/tmp/sp_pg_$i 2>&1 | grep "hmc cfg3 *split" | awk '{print $4, $5, $6, $7, $8, $9, $10}' | tr '\n' ' ')"; i=$((i+1)); done; done
grep -h error /tmp/sp_pg_*.err | head -5
