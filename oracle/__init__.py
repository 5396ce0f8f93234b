"""oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (mini-mcmc v0.8.3: MH, HMC, NUTS, split-R-hat/ESS) in plain C,
loaded through ctypes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (mini_mcmc_amd, libmmcmc.so) never does and has no CPU fallback.
"""
from .pyoracle import *  # noqa: F401,F403
