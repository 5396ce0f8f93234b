"""Run-time compiled NUTS units (built-in targets at dimensions without a compiled instance) against the run-time-D kernel,
one child process per case (a faulting case must not hide the others)."""
import sys, os, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
name, d, mode = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
tgt = (RosenbrockND if name == "Rosenbrock" else StandardNormal)(d)
init = init_with_seed(77, d, 31) * 0.5
outs = {}
used = None
for tag, v in (("7a", None), ("0a", None), ("0b", None), ("6", 6)):
    s = NUTS(tgt, init, 0.8, mode=mode).set_seed(5)
    if v is None:
        used = s.kernel_variant
    else:
        s.set_kernel_variant(v)
    outs[tag] = s._run(4, 7, False, "numpy")
print(name, d, "mode", mode, "default variant", used, "default==generic", np.array_equal(outs["7a"], outs["6"]), "again==generic", np.array_equal(outs["0a"], outs["6"]),
      "deterministic", np.array_equal(outs["0a"], outs["0b"]), flush=True)
'''
for name in ("Rosenbrock", "StdNormal"):
    for d in (9, 12, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27, 28, 29, 30, 31):
        for mode in (0, 1, 2):
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, name, str(d), str(mode)], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith(name)]
            if r.returncode != 0 or not line:
                print(name, d, "mode", mode, "CRASH rc", r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:120], flush=True)
            elif "False" in line[0] or "variant 6" in line[0]:  # a failure, or a unit that was refused by its check
                print(line[0], flush=True)
print("done")
