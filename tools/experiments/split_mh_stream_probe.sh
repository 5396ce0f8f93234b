# Round 5: what would ONE redefinition of the f32 noise stream buy the MH split kernel?  TIMING ONLY, wrong
# numbers: the product's sources are copied to /tmp and patched there --
#   a  8-byte inverse-CDF rows (two coefficients read, the other two constants): half the LDS bytes per normal;
#   b  Philox words shared across steps at D <= 2: ONE block per pair of transitions (w0, w1 -> step t; w2, w3 -> step t + 1);
#   c  both.
# bash tools/experiments/split_mh_stream_probe.sh
cd $GRAFT_REPO_ROOT
P=/tmp/csrc_stream_probe
rm -rf $P
for v in a b c; do mkdir -p $P/$v/mini_mcmc_amd $P/$v/tools; cp -r mini_mcmc_amd/csrc $P/$v/mini_mcmc_amd/; cp tools/split_probe.hip $P/$v/tools/; done
python3 - <<'EOP'
import re
P = "/tmp/csrc_stream_probe"
def patch_rows(path):
    s = open(path).read()
    old = """        const unsigned int off = (b >> 15) & 0x1ff0u;
        const mm_v4f v = *(mm_lds_f4_ptr)((unsigned int)(unsigned long long)tab + off);
        c[0] = v[0];
        c[1] = v[1];
        c[2] = v[2];
        c[3] = v[3];"""
    new = """        const unsigned int off = (b >> 16) & 0x0ff8u;
        typedef float mm_v2f_probe __attribute__((vector_size(8)));
        const mm_v2f_probe v = *(__attribute__((address_space(3))) const mm_v2f_probe *)((unsigned int)(unsigned long long)tab + off);
        c[0] = v[0];
        c[1] = v[1];
        c[2] = 0.25f;
        c[3] = 0.125f;"""
    assert old in s
    open(path, "w").write(s.replace(old, new))
def patch_philox(path):
    s = open(path).read()
    old = """        mm_u32x4x2 blk = mm_block_pair(seed, chain, iter, (uint32_t)b);
        if (b == 0) {
            const mm_f2 u = mm_spare_u24x2(blk);"""
    new = """        mm_u32x4x2 blk;
        if (D <= 2) { /* timing probe: one block for two steps */
            const mm_u32x4 one = mm_block(seed, chain, iter, (uint32_t)b);
            blk.w[0] = mm_u2{one.w[0], one.w[2]};
            blk.w[1] = mm_u2{one.w[1], one.w[3]};
            blk.w[2] = mm_u2{one.w[2], one.w[0]};
            blk.w[3] = mm_u2{one.w[3], one.w[1]};
        } else
            blk = mm_block_pair(seed, chain, iter, (uint32_t)b);
        if (b == 0) {
            const mm_f2 u = mm_spare_u24x2(blk);"""
    assert old in s
    open(path, "w").write(s.replace(old, new))
patch_rows(P + "/a/mini_mcmc_amd/csrc/mm_rng.h")
patch_philox(P + "/b/mini_mcmc_amd/csrc/mm_samplers.h")
patch_rows(P + "/c/mini_mcmc_amd/csrc/mm_rng.h")
patch_philox(P + "/c/mini_mcmc_amd/csrc/mm_samplers.h")
EOP
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMM_PROBE_MH_NN=3 -DMM_PROBE_HMC_NN=3"
hipcc $F tools/split_probe.hip -o /tmp/sp_st_0 2>/dev/null &
i=1
for v in a b c; do (cd $P/$v && hipcc $F tools/split_probe.hip -o /tmp/sp_st_$i 2>/tmp/sp_st_$i.err || tail -5 /tmp/sp_st_$i.err) & i=$((i+1)); done
wait
# more transition-wave draws may suit the cheaper noise: also QP = 0 for variant c
(cd $P/c && hipcc $F -DMM_PROBE_MH_QP=0 tools/split_probe.hip -o /tmp/sp_st_4 2>/dev/null)
for r in 1 2 3; do i=0; for v in "product" "a: 8-byte rows" "b: one Philox block per two steps" "c: both" "c with QP=0"; do echo "$v: $(/tmp/sp_st_$i 2>&1 | grep "split" | grep "mh cfg2" | awk '{print $5, $6, $7, $8, $10}' | tr '\n' ' ')"; i=$((i+1)); done; done
