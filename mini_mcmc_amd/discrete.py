"""Metropolis-Hastings on integer states -- host-side mirror of the reference's discrete users over the GPU engine.

`MetropolisHastings<S = i32 / usize, ...>` is generic in the reference (metropolis_hastings.rs:28-48); its discrete
targets and proposals live in test / example code, and are restated here as a registry:

    PoissonDist + PoissonRandomWalk       tests/metrohast_poisson_test.rs:18-85     -> PoissonReflect(lambda)
    BinomialDist + BinomialRandomWalk     tests/metrohast_poisson_test.rs:150-212   -> BinomialClamp(n, p)
    PoissonTarget + NonnegativeProposal   examples/poisson_mh.rs:8-76               -> PoissonNonneg(lambda)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass(frozen=True)
class PoissonReflect:
    """Poisson(lambda) on k >= 0; proposal k +- 1, a move below 0 stays at 0 (symmetric, log q = ln 0.5)."""
    lam: float
    kind = 16

    def params(self):
        return [float(self.lam)]


@dataclass(frozen=True)
class BinomialClamp:
    """Binomial(n, p) on 0..n; proposal k +- 1 clamped to [0, n]."""
    n: int
    p: float
    kind = 17

    def params(self):
        return [float(self.n), float(self.p)]


@dataclass(frozen=True)
class PoissonNonneg:
    """Poisson(lambda) on usize; proposal 0 -> 1 always, else k +- 1 with probability 1/2 each (asymmetric at 0)."""
    lam: float
    kind = 18

    def params(self):
        return [float(self.lam)]


class UserDiscreteModel:
    """An integer-state model of the user's own: `impl Target<i32, f64>` + `impl Proposal<i32, f64>` as HIP source defining
    `struct mmcmc_user_discrete` with `logp(params, k)`, `sample(params, current, step)`, `logq(params, from, to)`
    (include/mmcmc.h: mmcmc_discrete_register_source), compiled at run time.  `params`: up to eight numbers."""

    def __init__(self, name: str, source: str, params=()):
        kind = C.c_int(0)
        log = C.create_string_buffer(1 << 16)
        st = L.lib().mmcmc_discrete_register_source(name.encode(), source.encode(), C.byref(kind), log, len(log))
        self.compile_log = log.value.decode(errors="replace")
        if st != L.OK:
            raise L.MmcmcError(st, "mmcmc_discrete_register_source" + (": " + self.compile_log[-2000:] if self.compile_log else ""))
        self.kind, self.name = kind.value, name
        self._params = [float(v) for v in params] + [0.0] * (8 - len(params))

    def params(self):
        return self._params


class DiscreteMetropolisHastings:
    """MetropolisHastings::new(target, proposal, initial_states) for the models above; one chain per GPU lane.

    initial_states: [n_chains] or [n_chains, 1] integers.  run(n_collect, n_discard) -> int32 [n_chains, n_collect, 1]
    (the reference's Array3<i32>); `accept_counts()` since creation."""

    def __init__(self, model, initial_states, device: int = 0):
        init = np.ascontiguousarray(np.asarray(initial_states).reshape(-1), dtype=np.int32)
        self.n_chains = init.size
        self.model, self.device = model, device
        self._h = C.c_void_p()
        p = np.asarray(model.params(), dtype=np.float64)
        st = L.lib().mmcmc_mh_discrete_create(C.byref(self._h), model.kind, p.ctypes.data_as(C.POINTER(C.c_double)),
                                              init.ctypes.data_as(C.POINTER(C.c_int32)), self.n_chains, device)
        L.check(st, "mmcmc_mh_discrete_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib().mmcmc_mh_discrete_destroy(h)
            except Exception:
                pass
            self._h = None

    def seed(self, seed: int) -> "DiscreteMetropolisHastings":
        """metropolis_hastings.rs:187-193 (keys the counter-based stream; chains differ by their global index)"""
        L.check(L.lib().mmcmc_mh_discrete_seed(self._h, int(seed)), "mmcmc_mh_discrete_seed")
        return self

    def set_kernel_variant(self, variant: int) -> "DiscreteMetropolisHastings":
        """1: noise waves + transition waves (default), 0: one wave per SIMD; bit-identical."""
        L.check(L.lib().mmcmc_mh_discrete_set_kernel_variant(self._h, int(variant)), "mmcmc_mh_discrete_set_kernel_variant")
        return self

    def set_chain_offset(self, first_global_chain: int) -> "DiscreteMetropolisHastings":
        L.check(L.lib().mmcmc_mh_discrete_set_chain_offset(self._h, int(first_global_chain)), "set_chain_offset")
        return self

    def run(self, n_collect: int, n_discard: int = 0, to: str = "numpy"):
        """ChainRunner::run (core.rs:176-186)"""
        if to == "torch":
            import torch

            dev = torch.device("cuda", self.device)
            out = torch.empty((self.n_chains, n_collect, 1), device=dev, dtype=torch.int32)
            st = L.lib().mmcmc_mh_discrete_run(self._h, n_collect, n_discard, C.c_void_p(out.data_ptr()), 1,
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            L.check(st, "mmcmc_mh_discrete_run")
            return out
        out = np.empty((self.n_chains, n_collect, 1), dtype=np.int32)
        st = L.lib().mmcmc_mh_discrete_run(self._h, n_collect, n_discard, out.ctypes.data_as(C.c_void_p), 0, None)
        L.check(st, "mmcmc_mh_discrete_run")
        return out

    def state(self) -> np.ndarray:
        out = np.empty(self.n_chains, dtype=np.int32)
        L.check(L.lib().mmcmc_mh_discrete_state(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "state")
        return out

    def accept_counts(self) -> np.ndarray:
        out = np.empty(self.n_chains, dtype=np.uint64)
        L.check(L.lib().mmcmc_mh_discrete_accept_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "accept")
        return out
