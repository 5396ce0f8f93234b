"""The PCIe-inclusive rate of the headline call (never bench.py's `value`): HMC::run(400, 50) of config 3 with the sample
returned into HOST memory -- a pageable numpy array (what `to="numpy"` does) and a pinned buffer handed to the C ABI as
`out_is_device = 0`.   python tools/pcie_inclusive.py   (GPU box)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_mcmc_amd import _lib as L
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import RosenbrockND
from mini_mcmc_amd.hmc import HMC

Cn, D, NC, ND = 65536, 3, 400, 50
s = HMC(RosenbrockND(D), init_with_seed(Cn, D, 42, np.float32), 0.032, 10, device=0).set_seed(42)
lib = L.lib()
for _ in range(3):
    s.run(NC, ND, to="numpy", accept_counts=False)
torch.cuda.synchronize()
t, t_drop = [], []
for _ in range(5):
    t0 = time.perf_counter()
    res = s.run(NC, ND, to="numpy", accept_counts=False)  # a fresh numpy array per call: mapped by np.empty, faulted in by the copy
    t.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    del res  # unmapping 315 MB is the caller's cost (round 4's figure had it inside the timed region: 15 of its 31.8 ms)
    t_drop.append(time.perf_counter() - t0)
pageable = sorted(t)[len(t) // 2]
drop = sorted(t_drop)[len(t_drop) // 2]
# the same into ONE numpy array used again and again (its pages are already mapped: no page faults inside the call)
reuse = np.empty((Cn, NC, D), dtype=np.float32)
reuse[:] = 0
t = []
for _ in range(8):
    t0 = time.perf_counter()
    L.check(lib.mmcmc_hmc_run(s._h, NC, ND, reuse.ctypes.data, 0, None, None), "mmcmc_hmc_run")
    t.append(time.perf_counter() - t0)
reused = sorted(t[3:])[len(t[3:]) // 2]
t = []
for _ in range(5):  # a fresh allocation per call, timed without the allocation itself (np.empty maps, the call faults the pages in)
    fresh = np.empty((Cn, NC, D), dtype=np.float32)
    t0 = time.perf_counter()
    L.check(lib.mmcmc_hmc_run(s._h, NC, ND, fresh.ctypes.data, 0, None, None), "mmcmc_hmc_run")
    t.append(time.perf_counter() - t0)
    del fresh
fresh_ms = sorted(t)[len(t) // 2]
pinned = torch.empty((Cn, NC, D), dtype=torch.float32).pin_memory()
t = []
for _ in range(8):
    t0 = time.perf_counter()
    L.check(lib.mmcmc_hmc_run(s._h, NC, ND, pinned.data_ptr(), 0, None, None), "mmcmc_hmc_run")
    L.check(lib.mmcmc_hmc_sync(s._h), "sync")
    t.append(time.perf_counter() - t0)
pin = sorted(t[3:])[len(t[3:]) // 2]
b = Cn * NC * D * 4
print(json.dumps({"call": "HMC::run(400, 50), 65536 chains, D = 3, f32, sample to host", "sample_bytes": b,
                  "pageable_reused_destination_ms": reused * 1e3, "pageable_reused_GBps": b / reused / 1e9,
                  "pageable_fresh_destination_ms": fresh_ms * 1e3,
                  "pageable_ms": pageable * 1e3, "dropping_the_result_ms": drop * 1e3, "pageable_samples_per_s": Cn * NC / pageable, "pageable_GBps": b / pageable / 1e9,
                  "pinned_ms": pin * 1e3, "pinned_samples_per_s": Cn * NC / pin, "pinned_GBps": b / pin / 1e9}))
