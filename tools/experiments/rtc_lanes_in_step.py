"""Does hipRTC still get the lanes-in-step NUTS kernel of a run-time compiled unit wrong (round 3-4: RosenbrockND(19) / (23) in
f64, StandardNormal(25) in f32)?  A TUNING build launches that kernel from an ordinary handle (MMCMC_RTC_RUN_KERNEL=0; the product
never launches it from a hipRTC-built unit), here for the built-in targets at those dimensions, against the run-time-dimension
kernel (variant 6), with the unit built by hipcc, by the system's hipRTC (the process never imports torch) and by the hipRTC
PyTorch bundles (torch imported first).  One child process per case; the child says which libhiprtc / libamd_comgr it mapped.
    python3 tools/experiments/rtc_lanes_in_step.py        (with mini_mcmc_amd/libmmcmc.so a TUNING build)"""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CHILD = r'''
import sys, json, os
root, name, d, mode, compiler, with_torch, run_kernel = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6] == "1", sys.argv[7]
if not with_torch:
    sys.modules["torch"] = None  # `import torch` raises ImportError: the engine then runs on the system's ROCm libraries
sys.path.insert(0, root)
import numpy as np
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal, set_rtc_compiler
from mini_mcmc_amd.nuts import NUTS
set_rtc_compiler(compiler)
tgt = (RosenbrockND if name == "Rosenbrock" else StandardNormal)(d)
init = init_with_seed(77, d, 31) * 0.5
res = {"target": name, "dim": d, "mode": mode, "compiler": compiler, "torch_imported": with_torch, "run_kernel": run_kernel}
ref = NUTS(tgt, init, 0.8, mode=mode).set_seed(5).set_kernel_variant(6)._run(4, 7, False, "numpy")
os.environ["MMCMC_RTC_RUN_KERNEL"] = run_kernel
outs = []
for rep in range(2):
    s = NUTS(tgt, init, 0.8, mode=mode).set_seed(5)
    res["variant"] = s.kernel_variant
    outs.append(s._run(4, 7, False, "numpy"))
res["equals_run_time_dimension_kernel"] = bool(np.array_equal(outs[0], ref))
res["reproduces_itself"] = bool(np.array_equal(outs[0], outs[1]))
res["max_abs_diff"] = float(np.nanmax(np.abs(outs[0].astype(np.float64) - ref.astype(np.float64))))
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "hiprtc" in l or "comgr" in l})
res["mapped"] = libs
print("RESULT " + json.dumps(res), flush=True)
'''
cases = (("Rosenbrock", 19, 2), ("Rosenbrock", 23, 2), ("StdNormal", 25, 1), ("StdNormal", 25, 0), ("Rosenbrock", 12, 2))
for name, d, mode in cases:
    for compiler, with_torch in (("hipcc", "1"), ("hiprtc", "0"), ("hiprtc", "1")):
        for run_kernel in ("0", "2"):
            try:
                r = subprocess.run([sys.executable, "-c", CHILD, ROOT, name, str(d), str(mode), compiler, with_torch, run_kernel],
                                   capture_output=True, text=True, timeout=600)
            except subprocess.TimeoutExpired:
                print(json.dumps({"target": name, "dim": d, "mode": mode, "compiler": compiler, "torch_imported": with_torch == "1", "run_kernel": run_kernel, "timeout": True}), flush=True)
                continue
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if line:
                print(line[0][7:], flush=True)
            else:
                print(json.dumps({"target": name, "dim": d, "mode": mode, "compiler": compiler, "torch_imported": with_torch == "1", "run_kernel": run_kernel,
                                  "crash_rc": r.returncode, "stderr_tail": (r.stderr.strip().splitlines() or [""])[-1][:200]}), flush=True)
