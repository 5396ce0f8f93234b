# Round 5: MH split kernel, accept test through the table-free filter (MM_SPLIT_MH_UFILT=1, the product) against ln u from
# the table in the noise waves (=0, the form before), and role layouts around it (QP: noise pairs per batch the transition
# wave draws itself, RB: ring half, NN: noise waves per pair); same box, three alternating rounds.
#   bash tools/experiments/split_mh_ufilt.sh            (the default variant list)
#   VARIANTS_FILE=f bash tools/experiments/split_mh_ufilt.sh   (one set of -D flags per line)
cd $GRAFT_REPO_ROOT
B="hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_HMC_NN=3 tools/split_probe.hip"
if [ -z "$VARIANTS_FILE" ]; then
  VARIANTS_FILE=/tmp/sp_uf_variants.txt
  cat > $VARIANTS_FILE <<'EOV'
-DMM_PROBE_MH_NN=3 -DMM_SPLIT_MH_UFILT=0
-DMM_PROBE_MH_NN=3 -DMM_SPLIT_MH_UFILT=1 -DMM_PROBE_MH_QP=2
-DMM_PROBE_MH_NN=3 -DMM_SPLIT_MH_UFILT=1 -DMM_PROBE_MH_QP=0
-DMM_PROBE_MH_NN=3 -DMM_SPLIT_MH_UFILT=1 -DMM_PROBE_MH_QP=1
EOV
fi
i=0
while read -r v; do
  $B $v -o /tmp/sp_uf_$i 2>/tmp/sp_uf_$i.err || { echo "build $i failed"; tail -5 /tmp/sp_uf_$i.err; } &
  i=$((i+1))
  if [ $((i % 8)) -eq 0 ]; then wait; fi
done < $VARIANTS_FILE
wait
n=$i
for r in 1 2 3; do i=0; while read -r v; do echo "[$v] $(/tmp/sp_uf_$i 2>&1 | grep "split" | grep "mh cfg2" | awk '{print $5, $6, $7, $8, $10}' | tr '\n' ' ')"; i=$((i+1)); done < $VARIANTS_FILE; done
