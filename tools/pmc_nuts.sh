cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_nuts
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $R/gpurun_out/pmc_nuts/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 16384 > $R/gpurun_out/pmc_nuts/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_nuts/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 16384 > $R/gpurun_out/pmc_nuts/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH -d $R/gpurun_out/pmc_nuts/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 16384 > $R/gpurun_out/pmc_nuts/p3.log 2>&1
tail -2 $R/gpurun_out/pmc_nuts/p*.log
find $R/gpurun_out/pmc_nuts -name "*counter_collection.csv" | head
