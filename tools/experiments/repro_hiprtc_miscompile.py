"""Stand-alone reproducer of the run-time compiler's miscompile, through the public API only.

A user restatement of RosenbrockND(D) is registered twice from the SAME source: once built by hipRTC (pinned with
mmcmc_rtc_set_compiler), once by `hipcc --genco` in a child process.  mmcmc_nuts_create verifies a user unit by running
its asynchronous-lane pair kernel against its lanes-in-step kernel (96 chains, 5 + 5 transitions, bit for bit,
csrc/mm_nuts_api.hip: rtc_unit_verified).  At the dimensions the round-3 fuzz found (19 and 23 in f64) the hipRTC build
of the lanes-in-step kernel is wrong, so the hipRTC unit is REFUSED (MMCMC_ERR_UNSUPPORTED) while the hipcc unit of the
same source is accepted and equals the library's own kernels; at dimensions where hipRTC is right both are accepted.
    python3 tools/experiments/repro_hiprtc_miscompile.py          (one child process per case: a wrong kernel may fault)"""
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CHILD = r'''
import sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np
from mini_mcmc_amd import _lib as L
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, UserTarget, set_rtc_compiler
from mini_mcmc_amd.nuts import NUTS
dim, mode, compiler = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
SRC = open(sys.argv[1] + "/tests/test_user_target.py").read().split('ROSENBROCK3 = r"""')[1].split('"""')[0]
SRC = SRC.replace("static constexpr int dim = 3;", f"static constexpr int dim = {dim};")
set_rtc_compiler(compiler)
user = UserTarget(f"ros{dim}_{compiler}", dim, SRC)
init = init_with_seed(77, dim, 31) * 0.5
res = {"dim": dim, "mode": mode, "compiler": user.compiler}
try:
    s = NUTS(user, init, 0.8, mode=mode).set_seed(5)
    out = s._run(4, 7, False, "numpy")
    b = NUTS(RosenbrockND(dim), init, 0.8, mode=mode).set_seed(5).set_kernel_variant(6)
    res["accepted"] = True
    res["equals_library_kernel"] = bool(np.array_equal(out, b._run(4, 7, False, "numpy")))
except L.MmcmcError as e:
    res["accepted"] = False
    res["status"] = e.status
print("RESULT " + json.dumps(res), flush=True)
'''
for dim, mode in ((19, 2), (23, 2), (12, 2), (19, 0)):
    for compiler in ("hiprtc", "hipcc"):
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(dim), str(mode), compiler], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if line:
            print(line[0][7:], flush=True)
        else:
            print(json.dumps({"dim": dim, "mode": mode, "compiler": compiler, "crash_rc": r.returncode,
                              "stderr_tail": (r.stderr.strip().splitlines() or [""])[-1][:200]}), flush=True)
