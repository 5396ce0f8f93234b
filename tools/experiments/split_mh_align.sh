# Round 6: config 2's output costs 14 % of the MH kernel (0.175 ms with, 0.150 without).  Is that the 128-byte pieces of a chain's
# row straddling cache lines (rows of 1000 x 8 B start at multiples of 64 B)?  The same 1100 transitions with rows of 1008 and
# 1024 draws (whole lines).
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3 -DMM_PROBE_MH_ALIGNED tools/split_probe.hip -o /tmp/sp_al 2>/dev/null
for r in 1 2 3; do timeout 120 /tmp/sp_al 2>&1 < /dev/null | grep "split" | grep "mh" | awk '{print $1, $2, $4, $5, $6, $7}'; done
