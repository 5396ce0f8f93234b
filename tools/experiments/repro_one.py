import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND, StandardNormal
from mini_mcmc_amd.nuts import NUTS
name, d, mode, v, prog = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))
tgt = (RosenbrockND if name == "Rosenbrock" else StandardNormal)(d)
init = init_with_seed(77, d, 31) * 0.5
outs = []
for rep in range(4):
    s = NUTS(tgt, init, 0.8, mode=mode).set_seed(5).set_kernel_variant(v)
    outs.append(s._run(4, 7, prog, "numpy"))
    torch.cuda.synchronize()
print(name, d, mode, "variant", v, "progress", prog, "4 runs identical:", all(np.array_equal(outs[0], o) for o in outs), "checksum", float(outs[0].astype(np.float64).sum()), flush=True)
