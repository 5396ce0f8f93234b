import numpy as np, sys, time
sys.path.insert(0,'.')
from mini_mcmc_amd import stats as S
from oracle import pyoracle as O
rng=np.random.default_rng(1)
for (c,n,p) in [(5,2050,2),(4,2048,3),(3,2051,1),(2,32770,1),(2,40000,2),(1,2050,1),(1,100001,3),(3,70000,1),(40,36000,2),(2,33000,4)]:
    x=rng.standard_normal((c,n,p)).astype(np.float32).cumsum(axis=1)*0.01+rng.standard_normal((c,n,p)).astype(np.float32)
    try:
        t0=time.perf_counter(); r1,e1=S.split_rhat_mean_ess(x); dt=time.perf_counter()-t0
        r64,e64=S.split_rhat_mean_ess(x.astype(np.float64))
        r0,e0=O.split_rhat_mean_ess(x)
        print((c,n,p),'ok', np.max(np.abs(r1/r0-1)), np.max(np.abs(e1/e0-1)), 'f64 sample', np.max(np.abs(r64/r1-1)), np.max(np.abs(e64/e1-1)), '%.1f ms' % (dt*1e3))
    except Exception as ex:
        print((c,n,p),'EXC',type(ex).__name__,str(ex)[:200])
