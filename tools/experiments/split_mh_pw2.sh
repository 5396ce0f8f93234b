# Round 6, second pass of split_mh_pw.sh: with ONE pair per workgroup the transition wave is no longer the oldest wave of its SIMD
# (the four workgroups of a CU are of four ages and the arbiter serves the oldest first): s_setprio for the transition wave.
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_PW=1 -DMM_SPLIT_PRIO_Q=1" "-DMM_PROBE_PW=1 -DMM_SPLIT_PRIO_Q=3" "-DMM_PROBE_PW=2 -DMM_SPLIT_PRIO_Q=3" "-DMM_PROBE_PW=1 -DMM_SPLIT_PRIO_Q=3 -DMM_PROBE_PW_HMC=1 -DMM_PROBE_HMC_RB=8" "-DMM_SPLIT_PROFILE -DMM_PROBE_PW=1 -DMM_SPLIT_PRIO_Q=3"; do
  $B $v -o /tmp/sp_pw_$i 2>/dev/null &
  i=$((i+1))
done
wait
for r in 1 2 3; do i=0; for v in "PW=4 (product)" "PW=1 prio 1" "PW=1 prio 3" "PW=2 prio 3" "PW=1 prio 3 (MH + HMC, ring half 8)"; do echo "$v: $(/tmp/sp_pw_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7, $9}' | tr '\n' ' ')"; i=$((i+1)); done; done
echo "--- role profile, PW=1 prio 3"; /tmp/sp_pw_5 2>&1 | grep -A1 "mh cfg2 *split"
