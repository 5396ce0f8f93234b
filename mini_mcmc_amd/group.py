"""Device groups: one `run` for every chain on N GPUs of this process (include/mmcmc.h: mmcmc_*_group_*).

The reference's `run` executes ALL chains of a sampler (ChainRunner::run core.rs:176-186, HMC::run hmc.rs:137-158); a
group keeps that contract across GPUs: contiguous shards, the stream keyed by the global chain index (the sample does
not depend on the number of devices), no collective while sampling, split-R-hat / ESS reduced over RCCL inside the
library.  (One process per GPU with torch.distributed -- bench.py, stats.split_rhat_mean_ess_distributed -- is the other
way to use several GPUs; this one needs no launcher.)"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .distributions import IsotropicGaussian, Target


class _Group:
    _prefix = ""

    def _fn(self, name):
        return getattr(L.lib(), f"mmcmc_{self._prefix}_group_{name}")

    def set_chain_offset(self, first_global_chain: int):
        L.check(self._fn("set_chain_offset")(self._h, int(first_global_chain)), "group_set_chain_offset")
        return self

    def run_async(self, n_collect: int, n_discard: int = 0):
        """mmcmc_{mh,hmc}_group_run_async: the launches are only ENQUEUED on the shards' streams and the call returns at
        once; the sample stays on the devices (`shards`), `sync()` waits, the diagnostics and `state()` order themselves
        behind the queued work."""
        L.check(self._fn("run_async")(self._h, n_collect, n_discard), f"mmcmc_{self._prefix}_group_run_async")
        self.accept_counts = None

    def stats_phases(self) -> dict:
        """host wall-clock milliseconds of the last `split_rhat_mean_ess`, by phase"""
        ms = (C.c_double * 3)()
        L.check(self._fn("stats_phases")(self._h, ms), "group_stats_phases")
        return {"local_partials_ms": ms[0], "exchange_ms": ms[1], "finish_ms": ms[2]}

    def pci_bus_ids(self):
        """hipDeviceGetPCIBusId of every shard's device"""
        out = []
        for d in self.devices:
            buf = C.create_string_buffer(32)
            L.check(L.lib().mmcmc_device_pci_bus_id(int(d), buf, 32), "mmcmc_device_pci_bus_id")
            out.append(buf.value.decode())
        return out

    def run(self, n_collect: int, n_discard: int = 0, to_host: bool = True, accept_counts: bool = True):
        """sample [n_chains, n_collect, dim] on the host (to_host=False: it stays on the devices, see `shards`).
        Blocks until every shard's results are there, whatever is handed back (`run_async` is the queued spelling)."""
        out = np.empty((self.n_chains, n_collect, self.dim), dtype=self.dtype) if to_host else None
        acc = np.zeros(self.n_chains, dtype=np.uint64) if accept_counts else None
        st = self._fn("run")(self._h, n_collect, n_discard, out.ctypes.data if to_host else None,
                             acc.ctypes.data_as(C.POINTER(C.c_uint64)) if accept_counts else None)
        L.check(st, f"mmcmc_{self._prefix}_group_run")
        self.accept_counts = acc
        return out

    def sync(self):
        """wait for every run queued on the shards' streams"""
        L.check(self._fn("sync")(self._h), "group_sync")
        return self

    def timer_start(self):
        """record a start event on every shard's stream"""
        L.check(self._fn("stream_timer")(self._h, 0, None), "group_stream_timer")

    def timer_stop(self) -> np.ndarray:
        """record the end events, wait for them: elapsed device milliseconds per shard since `timer_start`"""
        ms = np.zeros(len(self.devices), dtype=np.float32)
        L.check(self._fn("stream_timer")(self._h, 1, ms.ctypes.data_as(C.POINTER(C.c_float))), "group_stream_timer")
        return ms

    def exchange(self):
        """(status, rccl_ranks): how the diagnostics' statistics will travel, decided when the group was created
        (1 RCCL; 0 the host by design: a device listed twice; -1 / -2 the host as a fallback)"""
        st, n = C.c_int(0), C.c_int(0)
        L.check(self._fn("exchange")(self._h, C.byref(st), C.byref(n)), "group_exchange")
        return st.value, n.value

    def state(self) -> np.ndarray:
        out = np.empty((self.n_chains, self.dim), dtype=self.dtype)
        L.check(self._fn("state")(self._h, out.ctypes.data), "group_state")
        return out

    def split_rhat_mean_ess(self):
        """(rhat[dim], ess[dim]) of the last run's sample over the chains of ALL devices (stats.rs:416-423);
        `self.used_rccl` tells whether the statistics were exchanged by RCCL or through the host."""
        rhat = np.empty(self.dim, dtype=np.float32)
        ess = np.empty(self.dim, dtype=np.float32)
        used = C.c_int(0)
        st = self._fn("split_rhat_mean_ess")(self._h, rhat.ctypes.data_as(C.POINTER(C.c_float)),
                                             ess.ctypes.data_as(C.POINTER(C.c_float)), C.byref(used))
        L.check(st, "group_split_rhat_mean_ess")
        self.exchange_status = used.value  # 1 RCCL; 0 host by design (a device listed twice); -1 / -2 host as a fallback
        self.used_rccl = used.value == 1
        return rhat, ess

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HMCGroup(_Group):
    """HMC::new(target, initial_positions, step_size, n_leapfrog) (hmc.rs:87-109) over `devices`."""

    _prefix = "hmc"

    def __init__(self, target: Target, initial_positions, step_size: float, n_leapfrog: int, devices=(0,)):
        init = np.ascontiguousarray(initial_positions)
        if init.dtype not in (np.float32, np.float64):
            init = init.astype(np.float32)
        self.n_chains, self.dim = init.shape
        self.dtype = init.dtype.type
        self.devices = [int(d) for d in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        d = target.desc()
        st = L.lib().mmcmc_hmc_group_create(C.byref(self._h), C.byref(d), init.ctypes.data, self.n_chains, float(step_size),
                                            int(n_leapfrog), L.F32 if self.dtype == np.float32 else L.F64, dev, len(self.devices))
        L.check(st, "mmcmc_hmc_group_create")

    def set_seed(self, seed: int) -> "HMCGroup":
        L.check(L.lib().mmcmc_hmc_group_seed(self._h, int(seed)), "mmcmc_hmc_group_seed")
        return self

    def shards(self):
        """[(device, first_chain, n_chains, device pointer of the shard's sample [n_i, n_collect, dim])]"""
        out = []
        for i in range(len(self.devices)):
            dev, first, n, ptr = C.c_int(), C.c_size_t(), C.c_size_t(), C.c_void_p()
            L.check(L.lib().mmcmc_hmc_group_shard(self._h, i, C.byref(dev), C.byref(first), C.byref(n), C.byref(ptr)), "group_shard")
            out.append((dev.value, first.value, n.value, ptr.value))
        return out


class MetropolisHastingsGroup(_Group):
    """MetropolisHastings::new(target, proposal, initial_states) (metropolis_hastings.rs:149-193) over `devices`."""

    _prefix = "mh"

    def __init__(self, target: Target, proposal: IsotropicGaussian, initial_states, devices=(0,)):
        init = np.ascontiguousarray(initial_states)
        if init.dtype not in (np.float32, np.float64):
            init = init.astype(np.float32)
        self.n_chains, self.dim = init.shape
        self.dtype = init.dtype.type
        self.devices = [int(d) for d in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        d, p = target.desc(), proposal.proposal_desc()
        st = L.lib().mmcmc_mh_group_create(C.byref(self._h), C.byref(d), C.byref(p), init.ctypes.data, self.n_chains,
                                           L.F32 if self.dtype == np.float32 else L.F64, dev, len(self.devices))
        L.check(st, "mmcmc_mh_group_create")

    def seed(self, seed: int) -> "MetropolisHastingsGroup":
        L.check(L.lib().mmcmc_mh_group_seed(self._h, int(seed)), "mmcmc_mh_group_seed")
        return self


class NUTSGroup(_Group):
    """NUTS::new(target, initial_positions, target_accept_p) (nuts.rs:123-129) over `devices`; mode as in `nuts.NUTS`."""

    _prefix = "nuts"

    def __init__(self, target: Target, initial_positions, target_accept_p: float, mode: int = 0, devices=(0,)):
        init = np.ascontiguousarray(initial_positions, dtype=np.float64)
        self.n_chains, self.dim = init.shape
        self.mode = int(mode)
        self.dtype = np.float64 if self.mode == 2 else np.float32
        self.devices = [int(d) for d in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        d = target.desc()
        st = L.lib().mmcmc_nuts_group_create(C.byref(self._h), C.byref(d), init.ctypes.data_as(C.POINTER(C.c_double)), self.n_chains,
                                             float(target_accept_p), self.mode, dev, len(self.devices))
        L.check(st, "mmcmc_nuts_group_create")

    def set_seed(self, seed: int) -> "NUTSGroup":
        L.check(L.lib().mmcmc_nuts_group_seed(self._h, int(seed)), "mmcmc_nuts_group_seed")
        return self

    def set_max_depth(self, max_depth: int) -> "NUTSGroup":
        L.check(L.lib().mmcmc_nuts_group_set_max_depth(self._h, int(max_depth)), "mmcmc_nuts_group_set_max_depth")
        return self

    def run(self, n_collect: int, n_discard: int = 0, to_host: bool = True, progress: bool = False):
        """NUTS::run (N - 1 transitions, row 0 may be the initial point, nuts.rs:457-471) or, progress=True, the stepping
        of run_progress (nuts.rs:491-522); sample [n_chains, n_collect, dim] of the mode's tensor type."""
        out = np.empty((self.n_chains, n_collect, self.dim), dtype=self.dtype) if to_host else None
        st = L.lib().mmcmc_nuts_group_run(self._h, n_collect, n_discard, out.ctypes.data if to_host else None, int(progress))
        L.check(st, "mmcmc_nuts_group_run")
        return out

    def leapfrog_counts(self) -> np.ndarray:
        out = np.zeros(self.n_chains, dtype=np.uint64)
        L.check(L.lib().mmcmc_nuts_group_leapfrog_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "group_leapfrog_counts")
        return out
