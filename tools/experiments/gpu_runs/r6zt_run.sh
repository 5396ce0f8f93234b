cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_user_target.py tests/test_abi_and_host.py -x -q > gpurun_out/r6zt_user_target_tests.log 2>&1 < /dev/null; tail -3 gpurun_out/r6zt_user_target_tests.log
python -c "
from mini_mcmc_amd.distributions import rtc_compiler_info as f; print(f())"
