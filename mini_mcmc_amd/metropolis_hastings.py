"""`MetropolisHastings` -- host-side mirror of src/metropolis_hastings.rs:149-193 over the GPU engine."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .core import _Sampler
from .distributions import IsotropicGaussian, Target


class MetropolisHastings(_Sampler):
    """MetropolisHastings::new(target, proposal, initial_states) (metropolis_hastings.rs:149-159).

    One chain per GPU lane.  `seed(s)` mirrors metropolis_hastings.rs:187-193; `run` is ChainRunner::run
    (core.rs:176-186).  The state element type follows `initial_states.dtype` (float32 / float64)."""

    _prefix = "mh"

    def __init__(self, target: Target, proposal: IsotropicGaussian, initial_states, device: int = 0):
        super().__init__()
        init = np.ascontiguousarray(initial_states)
        if init.dtype not in (np.float32, np.float64):
            init = init.astype(np.float64)
        if init.ndim != 2:
            raise ValueError("initial_states must be [n_chains, dim]")
        self.n_chains, self.dim = init.shape
        self.dtype = init.dtype.type
        self.device = device
        self.target, self.proposal = target, proposal
        if target.dim != self.dim:
            if type(target).__name__ in ("RosenbrockND", "StandardNormal", "IsotropicGaussian"):
                target = type(target)(self.dim) if not isinstance(target, IsotropicGaussian) else IsotropicGaussian(
                    target.std, self.dim)
                self.target = target
            else:
                raise ValueError(f"target dim {target.dim} != state dim {self.dim}")
        d, p = target.desc(), proposal.proposal_desc()
        st = L.lib().mmcmc_mh_create(C.byref(self._h), C.byref(d), C.byref(p), init.ctypes.data, self.n_chains,
                                     L.F32 if self.dtype == np.float32 else L.F64, device)
        L.check(st, "mmcmc_mh_create")

    def seed(self, seed: int) -> "MetropolisHastings":
        L.check(L.lib().mmcmc_mh_seed(self._h, int(seed)), "mmcmc_mh_seed")
        return self
