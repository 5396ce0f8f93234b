# rocprofv3 PMC passes over the one-chain-per-lane NUTS kernel (RosenbrockND(3), 65 536 chains, 100 + 100 transitions)
# usage (on the GPU box): NUTS_VARIANT=5 NUTS_MODE=0 bash tools/pmc_nuts_small.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${NUTS_VARIANT:-5}
M=${NUTS_MODE:-0}
O=$R/gpurun_out/pmc_nuts_small_v${V}_m${M}
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts3 $M $V > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts3 $M $V > $O/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py nuts3 $M $V > $O/p3.log 2>&1
python3 $R/tools/pmc_sum.py $O nuts
tail -1 $O/p1.log
