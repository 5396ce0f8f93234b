"""HMC on the dense Gaussian (GaussianND, D = 8 / 16 / 32): kernel time per variant -> one JSON line each.
Variant 3 = lane groups + MFMA (csrc/mm_hmc_lg.h: v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4), variants 2 / 0 = one chain per
lane, pipelined / plain (csrc/mm_kernels.h).  mfma_roofline_frac: 2 D^2 flops per leapfrog step against the dense matrix
peak of the type (f64 78.6, f32 157.3 TFLOP/s)."""
import sys, os, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mini_mcmc_amd.core import init_with_seed
from mini_mcmc_amd.distributions import GaussianND
from mini_mcmc_amd.hmc import HMC
C = 65536
for D, dt, variants in ((32, np.float64, (3, 2, 0)), (16, np.float64, (3, 2)), (32, np.float32, (3, 2, 0)), (16, np.float32, (3, 2)),
                        (8, np.float32, (2,))):
    g = GaussianND.ill_conditioned(D, 100.0, 7)
    for v in variants:
        s = HMC(g, init_with_seed(C, D, 42, dt) * 0.1, 0.05, 10).set_seed(1)
        s.set_kernel_variant(v)
        s.run(20, 5, to="torch", accept_counts=False); torch.cuda.synchronize()
        s.run(100, 20, to="torch", accept_counts=False); torch.cuda.synchronize()
        ms = s.timing()["kernel_ms"]
        print(json.dumps({"target": f"GaussianND D={D} {dt.__name__}", "variant": s.kernel_variant, "chains": C,
                          "run": "(100,20) L=10", "kernel_ms": ms, "leapfrog_steps_per_s": C * 120 * 10 / (ms * 1e-3),
                          "samples_per_s": C * 100 / (ms * 1e-3),
                          "mfma_roofline_frac": (C * 120 * 10 / (ms * 1e-3)) * 2 * D * D / (78.6e12 if dt is np.float64 else 157.3e12)}),
              flush=True)
        s.close()
