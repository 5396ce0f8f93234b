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


def run_chain_of_handles(target: Target, initial_positions, eps_range, leapfrog_range, n_per_launch: int, burn_launches: int,
                         keep_launches: int, seed: int = 42, schedule_seed: int = 7, device: int = 0):
    """A converged sample from the REFERENCE's sampler on a target where one fixed (step_size, n_leapfrog) does not mix
    (RosenbrockND(3): profiles/r5b_, r6m_converged_probe.jsonl, r6s_converged_hmc_robustness.jsonl).  `HMC` has no jitter (hmc.rs:87-121, 304-431), but its
    constructor takes any initial positions: launch k is `HMC::new(target, positions of launch k - 1, eps_k, L_k)` followed
    by `run(n_per_launch, 0)` (hmc.rs:137-158), with eps_k ~ U(eps_range), L_k ~ U{leapfrog_range} drawn on the host from
    numpy's PCG64(schedule_seed) -- every launch IS the reference's sampler, each leaves the target invariant, so does their
    sequence.  Handle k is seeded `seed + k` (a handle's stream starts at iteration 0).  The first `burn_launches` launches are
    discarded.  Returns (sample [n_chains, keep_launches * n_per_launch, dim] as a torch tensor on the device, info)."""
    import torch

    rng = np.random.default_rng(schedule_seed)
    state = np.ascontiguousarray(initial_positions)
    n_chains, dim = state.shape
    keep = torch.empty((n_chains, keep_launches * n_per_launch, dim), dtype=torch.float32 if state.dtype == np.float32 else torch.float64,
                       device=torch.device("cuda", device))
    kernel_ms, accepts, leapfrogs, schedule = 0.0, 0.0, 0.0, []
    for k in range(burn_launches + keep_launches):
        eps = float(rng.uniform(eps_range[0], eps_range[1]))
        n_leap = int(rng.integers(leapfrog_range[0], leapfrog_range[1] + 1))
        h = HMC(target, state, eps, n_leap, device=device).set_seed(seed + k)
        t = h.run(n_per_launch, 0, to="torch")
        torch.cuda.synchronize(device)
        kernel_ms += float(h.timing()["kernel_ms"])
        leapfrogs += float(n_chains) * n_per_launch * n_leap
        if k >= burn_launches:
            keep[:, (k - burn_launches) * n_per_launch:(k - burn_launches + 1) * n_per_launch] = t
            accepts += float(h.accept_counts.mean())
        schedule.append((eps, n_leap))
        state = h.state()
        h.close()
        del t
    info = {"kernel_ms": kernel_ms, "launches": burn_launches + keep_launches, "leapfrogs": leapfrogs,
            "accept_rate": accepts / (keep_launches * n_per_launch), "schedule_head": schedule[:4]}
    return keep, info
