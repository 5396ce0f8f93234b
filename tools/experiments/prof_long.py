import sys, torch
sys.path.insert(0, "/root/repo")
from mini_mcmc_amd import stats as S
c, n, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(c, n, d, device="cuda")
for _ in range(4):
    S.split_rhat_mean_ess(x)
torch.cuda.synchronize()
