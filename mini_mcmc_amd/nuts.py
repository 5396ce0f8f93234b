"""`NUTS` -- host-side mirror of src/nuts.rs:123-170, 194-353 over the GPU engine."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .distributions import IsotropicGaussian, Target


class NUTS:
    """NUTS::new(target, initial_positions, target_accept_p) (nuts.rs:123-129).

    mode 0: f32 tensors + f64 scalars (the reference's `NUTS<f64, Autodiff<NdArray>, _>`), 1: f32 + f32,
    2: f64 + f64.  `run(n_collect, n_discard)` keeps the reference's N-1-steps semantics; `run_progress` takes all
    N steps and returns (sample, RunStats) like nuts.rs:194-338 (without the terminal UI)."""

    def __init__(self, target: Target, initial_positions, target_accept_p: float, mode: int = 0, device: int = 0):
        init = np.ascontiguousarray(initial_positions, dtype=np.float64)
        if init.ndim == 1:
            init = init[None, :]
        self.n_chains, self.dim = init.shape
        self.mode, self.device = mode, device
        self.dtype = np.float64 if mode == 2 else np.float32
        if target.dim != self.dim:
            if type(target).__name__ in ("RosenbrockND", "StandardNormal"):
                target = type(target)(self.dim)
            elif isinstance(target, IsotropicGaussian):
                target = IsotropicGaussian(target.std, self.dim)
            else:
                raise ValueError(f"target dim {target.dim} != state dim {self.dim}")
        self.target = target
        self._h = C.c_void_p()
        d = target.desc()
        st = L.lib().mmcmc_nuts_create(C.byref(self._h), C.byref(d), init.ctypes.data_as(C.POINTER(C.c_double)),
                                       self.n_chains, float(target_accept_p), mode, device)
        L.check(st, "mmcmc_nuts_create")

    def set_seed(self, seed: int) -> "NUTS":
        """nuts.rs:347-353"""
        L.check(L.lib().mmcmc_nuts_seed(self._h, int(seed)), "mmcmc_nuts_seed")
        return self

    def set_chain_offset(self, first_global_chain: int) -> "NUTS":
        L.check(L.lib().mmcmc_nuts_set_chain_offset(self._h, int(first_global_chain)), "set_chain_offset")
        return self

    def set_max_depth(self, max_depth: int) -> "NUTS":
        L.check(L.lib().mmcmc_nuts_set_max_depth(self._h, int(max_depth)), "mmcmc_nuts_set_max_depth")
        return self

    def set_kernel_variant(self, variant: int) -> "NUTS":
        """One chain per lane: 0 = the lanes of a wave take their transitions in step, 4 = every lane advances on its
        own, one leaf per tick (default for dim <= 8; same results as 0).  1 = lane-group / MFMA mapping in one launch;
        2 / 3 = the same with tree-depth compaction by one launch per level / by a persistent scheduler (1..3: mode 2,
        GaussianND, dim 16 or 32, where 3 is the default; all three give bit-identical samples).  6 = run-time dimension
        (the N-dimensional built-in targets at any dim; the default where no compiled instance exists; same results as 0)."""
        L.check(L.lib().mmcmc_nuts_set_kernel_variant(self._h, int(variant)), "mmcmc_nuts_set_kernel_variant")
        return self

    def set_compaction(self, first_level: int = 5, n_groups: int = 0) -> "NUTS":
        """variants 2, 3: doublings below `first_level` run before the first re-packing; variant 2: `n_groups` chain groups
        with their own launch sequences (0 = one per 16 384 chains).  Neither changes a result."""
        L.check(L.lib().mmcmc_nuts_set_compaction(self._h, int(first_level), int(n_groups)), "mmcmc_nuts_set_compaction")
        return self

    @property
    def kernel_variant(self) -> int:
        return int(L.lib().mmcmc_nuts_kernel_variant(self._h))

    def _run(self, n_collect, n_discard, progress, to):
        if to == "torch":
            import torch

            dev = torch.device("cuda", self.device)
            out = torch.empty((self.n_chains, n_collect, self.dim), device=dev,
                              dtype=torch.float64 if self.mode == 2 else torch.float32)
            stream = torch.cuda.current_stream(dev).cuda_stream
            st = L.lib().mmcmc_nuts_run(self._h, n_collect, n_discard, out.data_ptr(), 1, int(progress), C.c_void_p(stream))
            L.check(st, "mmcmc_nuts_run")
            return out
        out = np.empty((self.n_chains, n_collect, self.dim), dtype=self.dtype)
        st = L.lib().mmcmc_nuts_run(self._h, n_collect, n_discard, out.ctypes.data, 0, int(progress), None)
        L.check(st, "mmcmc_nuts_run")
        L.check(L.lib().mmcmc_nuts_sync(self._h), "mmcmc_nuts_sync")
        return out

    def run(self, n_collect: int, n_discard: int, to: str = "numpy"):
        """NUTS::run (nuts.rs:163-170): sample [n_chains, n_collect, dim]."""
        return self._run(n_collect, n_discard, False, to)

    def run_progress(self, n_collect: int, n_discard: int, to: str = "numpy", every: int = 0, callback=None):
        """NUTS::run_progress (nuts.rs:172-345): (sample, RunStats); per-chain ChainTrackers fed the initial position and
        all n_discard + n_collect states (nuts.rs:486-506) end up in `self.tracker` (mmcmc_nuts_run_progress)."""
        from .core import _run_progress_c

        return _run_progress_c(self, L.lib().mmcmc_nuts_run_progress, n_collect, n_discard, every, callback, to, self.dtype)

    def positions(self) -> np.ndarray:
        out = np.empty((self.n_chains, self.dim), dtype=self.dtype)
        L.check(L.lib().mmcmc_nuts_state(self._h, out.ctypes.data), "mmcmc_nuts_state")
        return out

    def adapt_state(self) -> dict:
        out = np.empty((self.n_chains, 4), dtype=np.float64)
        L.check(L.lib().mmcmc_nuts_adapt_state(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "adapt_state")
        return dict(epsilon=out[:, 0], epsilon_bar=out[:, 1], h_bar=out[:, 2], mu=out[:, 3])

    def leapfrog_counts(self) -> np.ndarray:
        out = np.zeros(self.n_chains, dtype=np.uint64)
        L.check(L.lib().mmcmc_nuts_leapfrog_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "leapfrog_counts")
        return out

    def depth_histogram(self) -> np.ndarray:
        out = np.zeros(13, dtype=np.uint32)
        L.check(L.lib().mmcmc_nuts_depth_histogram(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "depth_histogram")
        return out

    def timing(self) -> dict:
        t = L.Timing()
        L.check(L.lib().mmcmc_nuts_timing(self._h, C.byref(t)), "mmcmc_nuts_timing")
        return dict(kernel_ms=t.kernel_ms, n_launches=t.n_launches, out_bytes=t.out_bytes, state_bytes=t.state_bytes)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().mmcmc_nuts_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
