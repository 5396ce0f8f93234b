"""Host-side mirror of the reference's `core` module (src/core.rs): initialisers and the sampler base class
(`ChainRunner::run` is implemented by the engine: one GPU lane per chain instead of one rayon task per chain)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def init_with_seed(n: int, d: int, seed: int, dtype=np.float64) -> np.ndarray:
    """core.rs:413-419: n starting points in d dimensions from the reference's own stream
    (SmallRng::seed_from_u64(seed) + StandardNormal), so `init_det` gives the reference's values."""
    out = np.empty((n, d), dtype=np.float64)
    st = L.lib().mmcmc_init_with_seed(n, d, seed, out.ctypes.data_as(C.POINTER(C.c_double)))
    L.check(st, "mmcmc_init_with_seed")
    return out.astype(dtype)


def init_det(n: int, d: int, dtype=np.float64) -> np.ndarray:
    """core.rs:404-409: init_with_seed(n, d, 42)."""
    return init_with_seed(n, d, 42, dtype)


def init(n: int, d: int, dtype=np.float64) -> np.ndarray:
    """core.rs:394-400: seeded from OS entropy."""
    seed = int.from_bytes(__import__("os").urandom(8), "little")
    return init_with_seed(n, d, seed, dtype)


class _Sampler:
    """Shared plumbing of the GPU samplers: owns one C-ABI handle."""

    _prefix = ""

    def __init__(self):
        self._h = C.c_void_p()
        self.accept_counts = None

    def _fn(self, name):
        return getattr(L.lib(), f"mmcmc_{self._prefix}_{name}")

    def set_chain_offset(self, first_global_chain: int):
        """Multi-GPU sharding: this handle's chains are global chains [offset, offset + n_chains)."""
        L.check(self._fn("set_chain_offset")(self._h, int(first_global_chain)), "set_chain_offset")
        return self

    def set_iters_per_launch(self, iters: int):
        L.check(self._fn("set_iters_per_launch")(self._h, int(iters)), "set_iters_per_launch")
        return self

    def set_kernel_variant(self, variant: int):
        """5 = noise waves + transition waves, up to four waves per SIMD (default for f32 up to dim 8), 2 = paired + pipelined
        noise, 0 = plain; results are identical."""
        L.check(self._fn("set_kernel_variant")(self._h, int(variant)), "set_kernel_variant")
        return self

    def run(self, n_collect: int, n_discard: int = 0, to: str = "numpy", accept_counts: bool = True,
            collect: bool = True):
        """`run(n_collect, n_discard)` of the reference (core.rs:176-186 / hmc.rs:137-158): returns the sample
        [n_chains, n_collect, dim].  to="numpy": host array; to="torch": a torch tensor in HBM, written by the
        kernel on torch's current stream (no host copy)."""
        acc = np.zeros(self.n_chains, dtype=np.uint64) if accept_counts else None
        accp = acc.ctypes.data_as(C.POINTER(C.c_uint64)) if accept_counts else None
        if to == "torch":
            import torch

            tdt = torch.float32 if self.dtype == np.float32 else torch.float64
            dev = torch.device("cuda", self.device)
            out = torch.empty((self.n_chains, n_collect, self.dim), dtype=tdt, device=dev) if collect else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            st = self._fn("run")(self._h, n_collect, n_discard, out.data_ptr() if collect else None, 1, accp,
                                 C.c_void_p(stream))
            L.check(st, f"mmcmc_{self._prefix}_run")
        else:
            out = np.empty((self.n_chains, n_collect, self.dim), dtype=self.dtype) if collect else None
            st = self._fn("run")(self._h, n_collect, n_discard, out.ctypes.data if collect else None, 0, accp, None)
            L.check(st, f"mmcmc_{self._prefix}_run")
            L.check(self._fn("sync")(self._h), "sync")
        self.accept_counts = acc
        return out

    def run_progress(self, n_collect: int, n_discard: int = 0, every: int = 0, callback=None, to: str = "numpy"):
        """`run_progress(n_collect, n_discard)` of the reference (hmc.rs:222-294 / core.rs:208-360): burn-in, then
        sampling with the reference's running diagnostics; returns (sample [n_chains, n_collect, dim], RunStats).
        The tracker feeding order (HMC: one MultiChainTracker after the burn-in; MH: per-chain ChainTrackers through
        it) is the library's (mmcmc_{mh,hmc}_run_progress, csrc/mm_progress.hip).  Instead of a progress bar redrawn
        after each step, `callback(done, p_accept, max_rhat)` is called after every `every` transitions (default: ten
        times per run) -- the two numbers the reference's bar shows.  The tracker ends up in `self.tracker`."""
        from . import stats as S

        return _run_progress_c(self, self._fn("run_progress"), n_collect, n_discard, every, callback, to,
                               self.dtype, count_from=n_discard if self._prefix == "hmc" else 0)

    def enable_timing(self, on: bool = True):
        """HIP-event bracketing of run() for timing() (default on)."""
        L.check(self._fn("enable_timing")(self._h, 1 if on else 0), "enable_timing")
        return self

    def sync(self):
        L.check(self._fn("sync")(self._h), "sync")

    def timing(self) -> dict:
        t = L.Timing()
        L.check(self._fn("timing")(self._h, C.byref(t)), "timing")
        return dict(kernel_ms=t.kernel_ms, n_launches=t.n_launches, out_bytes=t.out_bytes, state_bytes=t.state_bytes)

    def state(self) -> np.ndarray:
        out = np.empty((self.n_chains, self.dim), dtype=self.dtype)
        L.check(self._fn("state")(self._h, out.ctypes.data), "state")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def draw_noise(seed: int, chain_offset: int, iteration: int, n_chains: int, dim: int, dtype=np.float32, device=0, sampler: str = ""):
    """The noise the sampling kernels draw for (chain, iteration): (z[n_chains, dim], u[n_chains]); computed on
    the GPU by the same device functions the kernels use.  sampler="mh": the Metropolis-Hastings sampler's noise, which in
    f32 at dim <= 2 is a stream of its own (two transitions per Philox block: mmcmc_draw_noise_mh)."""
    z = np.empty((n_chains, dim), dtype=dtype)
    u = np.empty(n_chains, dtype=dtype)
    fn = L.lib().mmcmc_draw_noise_mh if sampler == "mh" else L.lib().mmcmc_draw_noise
    st = fn(seed, chain_offset, iteration, n_chains, dim,
                                  L.F32 if dtype == np.float32 else L.F64, z.ctypes.data, u.ctypes.data, device)
    L.check(st, "mmcmc_draw_noise")
    return z, u


def _run_progress_c(obj, fn, n_collect, n_discard, every, callback, to, dtype, count_from=0):
    """Shared by the three samplers: mmcmc_*_run_progress with a ctypes callback.  `callback(done, p_accept, max_rhat)`:
    `done` counts COLLECTED transitions for HMC (its burn-in is unobserved, hmc.rs:222-294: done runs to n_collect) and ALL
    transitions, burn-in included, for MH and NUTS (their trackers see every state, core.rs:208-360 / nuts.rs:172-345: done
    runs to n_discard + n_collect).  An exception raised by the callback is re-raised after the run has finished."""
    from . import stats as S

    # An exception raised inside a ctypes callback is printed and swallowed by ctypes, and the C loop cannot be unwound from
    # Python: the first one (KeyboardInterrupt included) is kept, later callbacks are skipped, and it is re-raised when the C
    # call has returned (the run itself completes: the C ABI's callback has no return value to stop it with).
    raised = []

    def tramp(user, done, total, p, r):
        if raised:
            return
        try:
            callback(int(done) - count_from, float(p), float(r))
        except BaseException as e:  # noqa: BLE001 -- re-raised below
            raised.append(e)

    cb = L.PROGRESS_FN(tramp) if callback else None
    rs = L.RunStats()
    tr = C.c_void_p()
    cbp = C.cast(cb, C.c_void_p) if cb else None
    if to == "torch":
        import torch

        dev = torch.device("cuda", obj.device)
        out = torch.empty((obj.n_chains, n_collect, obj.dim), dtype=torch.float32 if dtype == np.float32 else torch.float64, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        st = fn(obj._h, n_collect, n_discard, int(every), cbp, None, C.c_void_p(out.data_ptr()), 1, C.byref(rs), C.byref(tr), stream)
    else:
        out = np.empty((obj.n_chains, n_collect, obj.dim), dtype=dtype)
        st = fn(obj._h, n_collect, n_discard, int(every), cbp, None, C.c_void_p(out.ctypes.data), 0, C.byref(rs), C.byref(tr), None)
    L.check(st, "run_progress")
    obj.tracker = S.MultiChainTracker._adopt(tr)
    if raised:
        raise raised[0]
    return out, S._run_stats_from_c(rs)
