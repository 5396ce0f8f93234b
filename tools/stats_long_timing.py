"""split-R-hat / ESS of long chains ([65536, 4000, 3] N1 = 2, [65536, 8000, 3] N1 = 4, [16384, 16000, 3] N1 = 8, [16384, 20000, 3] N1 = 16; f32 in HBM): end-to-end ms per call
through the long-chain power-spectrum kernels (mm_chain_fft_long_kernel at N1 = 2, mm_chain_fft_res_kernel beyond: round 6; the last three shapes
had no power-spectrum kernel before it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_mcmc_amd import stats as S
for c, n, d in ((65536, 4000, 3), (65536, 8000, 3), (16384, 16000, 3), (16384, 20000, 3), (65536, 4000, 2), (65536, 4000, 1), (16384, 40000, 3), (4096, 200000, 3), (64, 20000, 2), (16384, 5000, 3), (16384, 12000, 3), (16384, 32768, 3)):
    x = torch.randn(c, n, d, device="cuda")
    for _ in range(2):
        r, e = S.split_rhat_mean_ess(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r, e = S.split_rhat_mean_ess(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"[{c}, {n}, {d}]: {ms:.3f} ms = {c * n * d * 4 / ms / 1e9:.2f} TB/s of sample  rhat {r[:1]} ess {e[:1]}", flush=True)
    del x
