# Round 4, third table probe (timing only): only ln u's table row broadcast (row 0 for every lane), the normals' gathers as
# they are -- the upper bound of what deciding the accept test without the exact logarithm could buy.
cd $GRAFT_REPO_ROOT
P=/tmp/csrc_probe3
rm -rf $P; mkdir -p $P/f/mini_mcmc_amd $P/f/tools
cp -r mini_mcmc_amd/csrc $P/f/mini_mcmc_amd/; cp tools/split_probe.hip $P/f/tools/
sed -i 's/const mm_v4f v = tab\[MM_ICDF_ROWS + r\];/const mm_v4f v = tab[MM_ICDF_ROWS + (r \& 0u)];/' $P/f/mini_mcmc_amd/csrc/mm_rng.h
grep -c "(r & 0u)" $P/f/mini_mcmc_amd/csrc/mm_rng.h
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3"
hipcc $F tools/split_probe.hip -o /tmp/sp_tc_0 2>/dev/null &
(cd $P/f && hipcc $F tools/split_probe.hip -o /tmp/sp_tc_1 2>/dev/null) &
wait
for r in 1 2 3; do i=0; for v in "product" "ln u row broadcast"; do echo "$v: $(/tmp/sp_tc_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7}' | tr '\n' ' ')"; i=$((i+1)); done; done
