cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r6zo_bench.json 2> gpurun_out/r6zo_bench.err < /dev/null; echo rc=$?; cut -c1-400 gpurun_out/r6zo_bench.json
MMCMC_BENCH_DRYRUN_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/r6zo_bench_dryrun2.json 2> gpurun_out/r6zo_bench_dryrun2.err < /dev/null; echo rc=$?; grep "^{" gpurun_out/r6zo_bench_dryrun2.json | cut -c1-600
