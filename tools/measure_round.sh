# usage (on the GPU box): ROUND_TAG=r1g bash tools/measure_round.sh  -> gpurun_out/$ROUND_TAG/{bench.json,configs.jsonl,prof_*}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${ROUND_TAG:-r1g}
python tools/bench_configs.py --skip nuts > gpurun_out/${ROUND_TAG:-r1g}/configs.jsonl 2> gpurun_out/${ROUND_TAG:-r1g}/configs.err
for v in 0 1 2 3; do  # kernel mappings of mmcmc_nuts_set_kernel_variant
  c=65536; [ $v = 0 ] && c=4096
  timeout 600 python tools/bench_configs.py --skip mh,hmc,stats --nuts-chains $c --nuts-variant $v 2>/dev/null | grep '"config": 5' >> gpurun_out/${ROUND_TAG:-r1g}/configs.jsonl
done
timeout 600 python tools/bench_configs.py --skip mh,hmc,stats --nuts-chains 16384 --nuts-variant 3 2>/dev/null | grep '"config": 5' >> gpurun_out/${ROUND_TAG:-r1g}/configs.jsonl
timeout 600 python tools/bench_configs.py --skip mh,hmc,stats --nuts-chains 16384 --nuts-variant 1 2>/dev/null | grep '"config": 5' >> gpurun_out/${ROUND_TAG:-r1g}/configs.jsonl
python bench.py > gpurun_out/${ROUND_TAG:-r1g}/bench.json 2> gpurun_out/${ROUND_TAG:-r1g}/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/prof_bench -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/bench_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/prof_nuts -o nuts --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --skip mh,hmc,stats --nuts-chains 65536 > $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/nuts_prof.json 2>/dev/null
cat $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/bench.json
ls $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/prof_bench $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r1g}/prof_nuts
