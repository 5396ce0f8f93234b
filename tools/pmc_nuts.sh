# rocprofv3 PMC passes over the config-5 NUTS kernel; NUTS_VARIANT selects the mapping (1, 2, 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_nuts${NUTS_VARIANT}
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 16384 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_probe.py nuts5 16384 > $O/p2.log 2>&1
find $O -name "*counter_collection.csv" | head
