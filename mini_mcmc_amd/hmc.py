"""`HMC` -- host-side mirror of src/hmc.rs:87-158, 304-377 over the GPU engine."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .core import _Sampler
from .distributions import IsotropicGaussian, Target


class HMC(_Sampler):
    """HMC::new(target, initial_positions, step_size, n_leapfrog) (hmc.rs:87-109)."""

    _prefix = "hmc"

    def __init__(self, target: Target, initial_positions, step_size: float, n_leapfrog: int, device: int = 0):
        super().__init__()
        init = np.ascontiguousarray(initial_positions)
        if init.dtype not in (np.float32, np.float64):
            init = init.astype(np.float32)
        if init.ndim != 2:
            raise ValueError("initial_positions must be [n_chains, dim]")
        self.n_chains, self.dim = init.shape
        self.dtype = init.dtype.type
        self.device = device
        self.step_size, self.n_leapfrog = float(step_size), int(n_leapfrog)
        if target.dim != self.dim:
            if type(target).__name__ in ("RosenbrockND", "StandardNormal"):
                target = type(target)(self.dim)
            elif isinstance(target, IsotropicGaussian):
                target = IsotropicGaussian(target.std, self.dim)
            else:
                raise ValueError(f"target dim {target.dim} != state dim {self.dim}")
        self.target = target
        d = target.desc()
        st = L.lib().mmcmc_hmc_create(C.byref(self._h), C.byref(d), init.ctypes.data, self.n_chains,
                                      self.step_size, self.n_leapfrog,
                                      L.F32 if self.dtype == np.float32 else L.F64, device)
        L.check(st, "mmcmc_hmc_create")

    def set_seed(self, seed: int) -> "HMC":
        """hmc.rs:118-121."""
        L.check(L.lib().mmcmc_hmc_seed(self._h, int(seed)), "mmcmc_hmc_seed")
        return self

    @property
    def kernel_variant(self) -> int:
        """0 / 2: one chain per lane (plain / paired noise); 3: lane groups + MFMA (f64 GaussianND of dim 16 or 32)."""
        return int(L.lib().mmcmc_hmc_kernel_variant(self._h))

    def step(self) -> None:
        """hmc.rs:304-377: one transition of every chain."""
        L.check(L.lib().mmcmc_hmc_step(self._h, None), "mmcmc_hmc_step")
        L.check(L.lib().mmcmc_hmc_sync(self._h), "mmcmc_hmc_sync")

    @property
    def positions(self) -> np.ndarray:
        """hmc.rs:49 `positions` [n_chains, D]."""
        return self.state()
