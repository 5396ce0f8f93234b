# Round 4, second table probe for the MH split kernel: the same table DATA read through the vector-memory path (global
# loads of the __device__ copies, L1-resident: 8 KB + 512 B) instead of LDS -- bit-identical numbers if adopted; here as
# patched copies of the sources in /tmp.  c: ln u's row from global; d: the normals' rows from global; e: both.
cd $GRAFT_REPO_ROOT
P=/tmp/csrc_probe2
rm -rf $P
for v in c d e; do mkdir -p $P/$v/mini_mcmc_amd $P/$v/tools; cp -r mini_mcmc_amd/csrc $P/$v/mini_mcmc_amd/; cp tools/split_probe.hip $P/$v/tools/; done
LN='s/const mm_v4f v = tab\[MM_ICDF_ROWS + r\];/const mm_v4f v = *reinterpret_cast<const mm_v4f *>(mm_lnm_tab_d + 4 * r);/'
NR='s/const mm_v4f v = \*(mm_lds_f4_ptr)((unsigned int)(unsigned long long)tab + off);/const mm_v4f v = *reinterpret_cast<const mm_v4f *>(reinterpret_cast<const char *>(mm_icdf_tab_d) + off);/'
sed -i "$LN" $P/c/mini_mcmc_amd/csrc/mm_rng.h
sed -i "$NR" $P/d/mini_mcmc_amd/csrc/mm_rng.h
sed -i "$LN; $NR" $P/e/mini_mcmc_amd/csrc/mm_rng.h
for v in c d e; do grep -c "reinterpret_cast<const mm_v4f \*>(mm_lnm_tab_d + 4 \* r)\|reinterpret_cast<const char \*>(mm_icdf_tab_d) + off" $P/$v/mini_mcmc_amd/csrc/mm_rng.h; done
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3"
hipcc $F tools/split_probe.hip -o /tmp/sp_tb_0 2>/dev/null &
i=1; for v in c d e; do (cd $P/$v && hipcc $F tools/split_probe.hip -o /tmp/sp_tb_$i 2>/tmp/sp_tb_$i.err) & i=$((i+1)); done
wait
for r in 1 2 3; do i=0; for v in "product" "ln u row from global" "normal rows from global" "both from global"; do echo "$v: $(/tmp/sp_tb_$i 2>&1 | grep "split" | awk '{print $1, $2, $4, $5, $6, $7, $8, $9}' | tr '\n' ' ')"; i=$((i+1)); done; done
