cd $GRAFT_REPO_ROOT
timeout 500 python tools/fuzz_variants.py 240 x > gpurun_out/r6zw_fuzz_long_diag.log 2>&1 < /dev/null; tail -3 gpurun_out/r6zw_fuzz_long_diag.log
timeout 1100 python tools/fuzz_variants.py 120 dgtnlhsm > gpurun_out/r6zw_fuzz_all.log 2>&1 < /dev/null; grep -v amdgpu gpurun_out/r6zw_fuzz_all.log | tail -12
