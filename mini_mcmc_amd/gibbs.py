"""`GibbsSampler` -- host-side mirror of src/gibbs.rs:139-203 over the GPU engine, for the built-in conditional.

A `Conditional` written by the user cannot run on the GPU; the one the reference ships -- the two-component Gaussian
mixture over the state [x, z] (gibbs.rs:231-285, examples/mixture_gibbs.rs:24-76) -- is `MixtureConditional` here."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass(frozen=True)
class MixtureConditional:
    """x | z ~ N(mu_z, sigma_z^2);  z | x = 1 with probability p1 / (p0 + p1), p0 = pi0 N(x; mu0, sigma0^2),
    p1 = (1 - pi0) N(x; mu1, sigma1^2)."""
    mu0: float
    sigma0: float
    mu1: float
    sigma1: float
    pi0: float

    def params(self):
        return [float(self.mu0), float(self.sigma0), float(self.mu1), float(self.sigma1), float(self.pi0)]


class GibbsSampler:
    """GibbsSampler::new(conditional, initial_states) (gibbs.rs:139-157); one chain per GPU lane, states f64
    [n_chains, 2].  `set_seed` mirrors gibbs.rs:179-187, `run` is ChainRunner::run (core.rs:176-186)."""

    def __init__(self, conditional: MixtureConditional, initial_states, device: int = 0):
        init = np.ascontiguousarray(initial_states, dtype=np.float64)
        if init.ndim != 2 or init.shape[1] != 2:
            raise ValueError("initial_states must be [n_chains, 2] (the mixture's state is [x, z])")
        self.n_chains, self.dim, self.device = init.shape[0], 2, device
        self.conditional = conditional
        self._h = C.c_void_p()
        p = np.asarray(conditional.params(), dtype=np.float64)
        st = L.lib().mmcmc_gibbs_mixture_create(C.byref(self._h), p.ctypes.data_as(C.POINTER(C.c_double)),
                                                init.ctypes.data_as(C.POINTER(C.c_double)), self.n_chains, device)
        L.check(st, "mmcmc_gibbs_mixture_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib().mmcmc_gibbs_mixture_destroy(h)
            except Exception:
                pass
            self._h = None

    def set_seed(self, seed: int) -> "GibbsSampler":
        L.check(L.lib().mmcmc_gibbs_mixture_seed(self._h, int(seed)), "mmcmc_gibbs_mixture_seed")
        return self

    def set_kernel_variant(self, variant: int) -> "GibbsSampler":
        """1: noise waves + transition waves (default), 0: one wave per SIMD; bit-identical."""
        L.check(L.lib().mmcmc_gibbs_mixture_set_kernel_variant(self._h, int(variant)), "mmcmc_gibbs_mixture_set_kernel_variant")
        return self

    def set_chain_offset(self, first_global_chain: int) -> "GibbsSampler":
        L.check(L.lib().mmcmc_gibbs_mixture_set_chain_offset(self._h, int(first_global_chain)), "set_chain_offset")
        return self

    def run(self, n_collect: int, n_discard: int = 0, to: str = "numpy"):
        if to == "torch":
            import torch

            dev = torch.device("cuda", self.device)
            out = torch.empty((self.n_chains, n_collect, 2), device=dev, dtype=torch.float64)
            st = L.lib().mmcmc_gibbs_mixture_run(self._h, n_collect, n_discard, C.c_void_p(out.data_ptr()), 1,
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            L.check(st, "mmcmc_gibbs_mixture_run")
            return out
        out = np.empty((self.n_chains, n_collect, 2), dtype=np.float64)
        L.check(L.lib().mmcmc_gibbs_mixture_run(self._h, n_collect, n_discard, out.ctypes.data_as(C.c_void_p), 0, None),
                "mmcmc_gibbs_mixture_run")
        return out

    def run_progress(self, n_collect: int, n_discard: int = 0):
        """gibbs.rs run_progress via ChainRunner (core.rs:208-360): (sample, RunStats)."""
        from . import stats as S

        sample = self.run(n_collect, n_discard, to="torch")
        return sample.cpu().numpy(), S.run_stats(sample)

    def state(self) -> np.ndarray:
        out = np.empty((self.n_chains, 2), dtype=np.float64)
        L.check(L.lib().mmcmc_gibbs_mixture_state(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "state")
        return out
