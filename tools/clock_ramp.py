"""How long does the sampling kernel take as a function of the time the GPU has been busy?  After an idle period the
kernel of bench.py (config 3, run(400, 50)) is launched back to back, a HIP event every 10 launches: per-launch time of
every group of 10.  (bench.py's pre-roll is sized from this.)

    python tools/clock_ramp.py [idle_ms] [launches]
"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import _lib as L
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

idle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda", 0)
h = HMC(RosenbrockND(3), init_with_seed(65536, 3, 42, np.float32), 0.032, 10).set_seed(42)
out = torch.empty((65536, 400, 3), dtype=torch.float32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
lib = L.lib()
h.enable_timing(False)


def step():
    L.check(lib.mmcmc_hmc_run(h._h, 400, 50, out.data_ptr(), 1, None, stream), "run")


for rep in range(3):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch // 10 + 1)]
    evs[0].record()
    for i in range(n_launch):
        step()
        if (i + 1) % 10 == 0:
            evs[(i + 1) // 10].record()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) / 10 for i in range(len(evs) - 1)]
    t = np.cumsum([m * 10 for m in ms])
    print(f"rep {rep} idle {idle_ms} ms: per-launch ms by group of 10 (busy ms since start: value)")
    print("  " + "  ".join(f"{t[i]:.0f}:{ms[i]:.4f}" for i in range(0, len(ms), max(1, len(ms) // 30))), flush=True)
# the same with a synchronize every 100 launches (the shape of a pre-roll loop)
for gap_every in (100, 25):
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    ms = []
    for blk in range(n_launch // gap_every):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(gap_every):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / gap_every)
    print(f"synchronize every {gap_every} launches: per-launch ms per block: " + " ".join(f"{m:.4f}" for m in ms), flush=True)
