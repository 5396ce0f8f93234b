cd $GRAFT_REPO_ROOT
cp mini_mcmc_amd/libmmcmc.so /tmp/libmmcmc_product.so
cp mini_mcmc_amd/libmmcmc_tuning.so mini_mcmc_amd/libmmcmc.so
timeout 2400 python tools/experiments/rtc_lanes_in_step.py > gpurun_out/r6zr_rtc_lanes_in_step.jsonl 2>&1 < /dev/null
cp /tmp/libmmcmc_product.so mini_mcmc_amd/libmmcmc.so
cut -c1-420 gpurun_out/r6zr_rtc_lanes_in_step.jsonl
