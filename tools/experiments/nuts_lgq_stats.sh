# Scheduler statistics of config 5 (pick / fetch / work / hand-over ticks, units, chains per unit) from the TUNING build:
#   make -C mini_mcmc_amd/csrc TUNING=1 OUT=../libmmcmc_tuning.so     (here), then on the GPU box:
#   bash tools/experiments/nuts_lgq_stats.sh "ENV=VALUE ENV2=VALUE" "ENV=VALUE" ...     (one run per argument; "" = defaults)
cd $GRAFT_REPO_ROOT
cp mini_mcmc_amd/libmmcmc.so /tmp/libmmcmc_product.so
cp mini_mcmc_amd/libmmcmc_tuning.so mini_mcmc_amd/libmmcmc.so
[ $# -eq 0 ] && set -- ""
for kvs in "$@"; do
  echo "== [$kvs]"
  env $kvs MMCMC_LGQ_STATS=1 python tools/nuts_cfg5_timing.py 200 100 1 2>&1 | grep -E "lgq|kernel_ms" | cut -c1-330
done
cp /tmp/libmmcmc_product.so mini_mcmc_amd/libmmcmc.so
