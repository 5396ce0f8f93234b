# Round 6: the MH split kernel with the workgroup cut down to PW pairs -- PW = 1: ONE pair (64 chains, 1 transition +
# 3 noise waves, 256 threads, four workgroups per CU, the role of a wave rotated by blockIdx.x & 3; s_barrier joins 4 waves instead
# of 16, the table replicated per workgroup); PW = 2: two pairs (512 threads, two workgroups per CU, barrier of 8 waves) --
# against the product's PW = 4 (1024 threads, barrier of 16).  Timing + checksum (must be equal: same bits), then where the roles wait.
#   bash tools/experiments/split_mh_pw.sh            (on the GPU box)
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
i=0
for v in "" "-DMM_PROBE_PW=1" "-DMM_PROBE_PW=2" "-DMM_PROBE_PW=1 -DMM_PROBE_PW_HMC=1 -DMM_PROBE_HMC_RB=8" "-DMM_PROBE_PW=2 -DMM_PROBE_PW_HMC=1 -DMM_PROBE_HMC_RB=8" "-DMM_SPLIT_PROFILE" "-DMM_SPLIT_PROFILE -DMM_PROBE_PW=1" "-DMM_SPLIT_PROFILE -DMM_PROBE_PW=2"; do
  $B $v -o /tmp/sp_pw_$i 2>/dev/null &
  i=$((i+1))
done
wait
for r in 1 2 3; do i=0; for v in "PW=4 (product)" "PW=1 (MH)" "PW=2 (MH)" "PW=1 (MH + HMC, ring half 8)" "PW=2 (MH + HMC, ring half 8)"; do echo "$v: $(/tmp/sp_pw_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7, $9}' | tr '\n' ' ')"; i=$((i+1)); done; done
echo "--- role profile, PW=4"; /tmp/sp_pw_5 2>&1 | grep -A1 "mh cfg2 *split"
echo "--- role profile, PW=1"; /tmp/sp_pw_6 2>&1 | grep -A1 "mh cfg2 *split"
echo "--- role profile, PW=2"; /tmp/sp_pw_7 2>&1 | grep -A1 "mh cfg2 *split"
