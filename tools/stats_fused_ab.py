"""Round 5: the diagnostics of a [65536, n, 3] f32 sample through the C ABI -- the reduction's tail inside the transform kernel
("fft": one launch) against tail and inverse as launches of their own ("fft_separate": three), alternating, same process, same
sample; milliseconds per call (median of 50) and the largest relative difference of R-hat / ESS between the two."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mini_mcmc_amd import _lib as L
from mini_mcmc_amd import stats as S

lib = L.lib()
for n in (400, 1000, 200):
    x = torch.randn(65536, n, 3, device="cuda")
    rh, es = (C.c_float * 3)(), (C.c_float * 3)()

    def call():
        L.check(lib.mmcmc_split_rhat_mean_ess(C.c_void_p(x.data_ptr()), 1, L.F32, 65536, n, 3, rh, es, 0, None), "stats")
        return np.array(rh[:]), np.array(es[:])

    res = {}
    for rnd in range(3):
        for k in ("fft", "fft_separate"):
            S.set_kernel(k)
            for _ in range(5):
                call()
            t = []
            for _ in range(50):
                t0 = time.perf_counter()
                r = call()
                t.append(time.perf_counter() - t0)
            res.setdefault(k, []).append(sorted(t)[25] * 1e3)
            res[k + "_val"] = r
    S.set_kernel("auto")
    print(json.dumps({"shape": [65536, n, 3], "fused_ms": res["fft"], "separate_ms": res["fft_separate"],
                      "max_rel_diff_rhat": float(np.max(np.abs(res["fft_val"][0] / res["fft_separate_val"][0] - 1))),
                      "max_rel_diff_ess": float(np.max(np.abs(res["fft_val"][1] / res["fft_separate_val"][1] - 1)))}))
