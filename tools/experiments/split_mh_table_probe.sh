# Round 4: does the inverse-CDF table's gather (random 16-byte rows from LDS, half of the LDS's busy time bank conflicts:
# profiles/r2q_lds_mh.csv) bound the MH split kernel?  TIMING ONLY, wrong numbers: the product's sources are copied to /tmp and
# the row address is masked there so that (a) all rows lie in one 256-byte line = distinct banks or the same address (no
# conflict possible), (b) every lane reads row 0 (one broadcast).  bash tools/experiments/split_mh_table_probe.sh
cd $GRAFT_REPO_ROOT
P=/tmp/csrc_probe
rm -rf $P; mkdir -p $P/a/mini_mcmc_amd $P/b/mini_mcmc_amd $P/a/tools $P/b/tools
for v in a b; do cp -r mini_mcmc_amd/csrc $P/$v/mini_mcmc_amd/; cp tools/split_probe.hip $P/$v/tools/; done
sed -i 's/const unsigned int off = (b >> 15) \& 0x1ff0u;/const unsigned int off = (b >> 15) \& 0x00f0u;/; s/const mm_v4f v = tab\[MM_ICDF_ROWS + r\];/const mm_v4f v = tab[MM_ICDF_ROWS + (r \& 15u)];/' $P/a/mini_mcmc_amd/csrc/mm_rng.h
sed -i 's/const unsigned int off = (b >> 15) \& 0x1ff0u;/const unsigned int off = (b >> 15) \& 0x0000u;/; s/const mm_v4f v = tab\[MM_ICDF_ROWS + r\];/const mm_v4f v = tab[MM_ICDF_ROWS + (r \& 0u)];/' $P/b/mini_mcmc_amd/csrc/mm_rng.h
grep -c "0x00f0u" $P/a/mini_mcmc_amd/csrc/mm_rng.h; grep -c "0x0000u" $P/b/mini_mcmc_amd/csrc/mm_rng.h
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3"
hipcc $F tools/split_probe.hip -o /tmp/sp_tab_0 2>/dev/null &
(cd $P/a && hipcc $F tools/split_probe.hip -o /tmp/sp_tab_1 2>/dev/null) &
(cd $P/b && hipcc $F tools/split_probe.hip -o /tmp/sp_tab_2 2>/dev/null) &
wait
for r in 1 2 3; do i=0; for v in "product" "rows of one line (no bank conflict possible)" "row 0 for every lane (one broadcast)"; do echo "$v: $(/tmp/sp_tab_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7}' | tr '\n' ' ')"; i=$((i+1)); done; done
