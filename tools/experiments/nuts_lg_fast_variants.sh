# Round 5: the product's doubling in its best case (tools/nuts_lg_floor_probe.hip) under build variants of the fast path.
#   VARIANTS_FILE=f bash tools/experiments/nuts_lg_fast_variants.sh     (one set of -D flags per line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I mini_mcmc_amd/csrc tools/nuts_lg_floor_probe.hip"
i=0
while read -r v; do
  $B $v -o /tmp/lgf_$i 2>/tmp/lgf_$i.err || tail -3 /tmp/lgf_$i.err &
  i=$((i+1))
done < $VARIANTS_FILE
wait
i=0
while read -r v; do echo "[$v] $(/tmp/lgf_$i | grep "mm_lg_doubling" | sed 's/.*level \([0-9]*\) x.*cycles_per_leaf_iteration_at_2.4GHz": \([0-9]*\).*/j=\1: \2/' | tr '\n' ' ')"; i=$((i+1)); done < $VARIANTS_FILE
