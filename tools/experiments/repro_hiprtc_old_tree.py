"""Which compiler got the lanes-in-step NUTS kernel of a run-time compiled unit wrong in rounds 3-4?  Runs round 4's OWN library
and sources (a tree extracted from the round-4 commit into _r4tree/, built there) through round 4's reproducer -- a user
restatement of RosenbrockND(D) built by hipRTC, accepted only if its pair kernel and its lanes-in-step kernel agree -- twice per
case: with torch imported first (the process maps the hipRTC / comgr 7.0.2 PyTorch bundles) and with the import blocked (the
system's ROCm 7.2 libraries).  Prints which libraries each child mapped.
    git archive <round-4 commit> mini_mcmc_amd tools tests/test_user_target.py include | tar -x -C _r4tree; make -C _r4tree/mini_mcmc_amd/csrc
    python3 tools/experiments/repro_hiprtc_old_tree.py _r4tree"""
import json, os, subprocess, sys
ROOT = os.path.abspath(sys.argv[1])
CHILD = r'''
import sys, json
with_torch = sys.argv[5] == "1"
if not with_torch:
    sys.modules["torch"] = None
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
res = {"dim": dim, "mode": mode, "compiler": user.compiler, "torch_imported": with_torch}
try:
    s = NUTS(user, init, 0.8, mode=mode).set_seed(5)
    out = s._run(4, 7, False, "numpy")
    b = NUTS(RosenbrockND(dim), init, 0.8, mode=mode).set_seed(5).set_kernel_variant(6)
    res["accepted"] = True
    res["equals_library_kernel"] = bool(np.array_equal(out, b._run(4, 7, False, "numpy")))
except L.MmcmcError as e:
    res["accepted"] = False
    res["status"] = e.status
res["mapped"] = sorted({l.split()[-1].split("/")[-1] + (" (torch)" if "/torch/" in l else "") for l in open("/proc/self/maps") if "hiprtc" in l or "comgr" in l})
print("RESULT " + json.dumps(res), flush=True)
'''
for dim, mode in ((19, 2), (23, 2), (12, 2), (19, 0)):
    for compiler, with_torch in (("hiprtc", "1"), ("hiprtc", "0"), ("hipcc", "1")):
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(dim), str(mode), compiler, with_torch], capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            print(json.dumps({"dim": dim, "mode": mode, "compiler": compiler, "torch_imported": with_torch == "1", "timeout": True}), flush=True)
            continue
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if line:
            print(line[0][7:], flush=True)
        else:
            print(json.dumps({"dim": dim, "mode": mode, "compiler": compiler, "torch_imported": with_torch == "1", "crash_rc": r.returncode,
                              "stderr_tail": (r.stderr.strip().splitlines() or [""])[-1][:200]}), flush=True)
