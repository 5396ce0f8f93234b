cd $GRAFT_REPO_ROOT
cp mini_mcmc_amd/libmmcmc.so /tmp/libmmcmc_product.so
cp mini_mcmc_amd/libmmcmc_tuning.so mini_mcmc_amd/libmmcmc.so
timeout 600 python tools/experiments/nuts_pair_lanes.py 2>&1 < /dev/null | grep -v amdgpu > gpurun_out/r6zp_nuts_pair_lanes.jsonl
cp /tmp/libmmcmc_product.so mini_mcmc_amd/libmmcmc.so
cat gpurun_out/r6zp_nuts_pair_lanes.jsonl
