"""mini_mcmc_amd -- MI355X-native many-chain engine for mini-mcmc's per-chain inner loop.

Python here is host plumbing over the C ABI (include/mmcmc.h -> libmmcmc.so, hand-written HIP for gfx950):
module names and call shapes follow the reference crate (core, distributions, metropolis_hastings, hmc, ...).
There is no CPU fallback; importing works without a GPU, constructing a sampler does not.
"""
from . import core, discrete, distributions, gibbs, hmc, metropolis_hastings, nuts, stats  # noqa: F401
from ._lib import LIB_PATH, MmcmcError, lib  # noqa: F401

__all__ = ["core", "discrete", "distributions", "gibbs", "hmc", "metropolis_hastings", "nuts", "stats", "lib", "LIB_PATH", "MmcmcError"]
