"""Diagnostics -- host-side mirror of the reference's `stats` module (src/stats.rs) over the GPU reduction.

split_rhat_mean_ess / RunStats / basic_stats keep the reference's semantics, including its definition of the split
"R-hat" as sqrt(W / var+) (stats.rs:425-427; the inverse of Stan's), and additionally expose the conventional value.
With chains sharded over ranks (one process per GPU), `split_rhat_mean_ess_distributed` exchanges only sufficient
statistics: an all-gather of per-half-chain means / sums of squares and an all-reduce of the lag sums (RCCL over
xGMI through torch.distributed; gloo on CPU tensors in the tests).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

_fp = C.POINTER(C.c_float)


@dataclass
class BasicStats:
    """stats.rs:373-381"""
    name: str
    min: float
    median: float
    max: float
    mean: float
    std: float

    def __str__(self):  # stats.rs:383-392
        return (f"{self.name} in [{self.min:.2f}, {self.max:.2f}], median: {self.median:.2f}, "
                f"mean: {self.mean:.2f} ± {self.std:.2f}")


@dataclass
class RunStats:
    """stats.rs:338-342"""
    ess: BasicStats
    rhat: BasicStats

    def __str__(self):
        return f"{self.ess}\n{self.rhat}"


def basic_stats(name: str, data) -> BasicStats:
    """stats.rs:310-336"""
    x = np.ascontiguousarray(data, dtype=np.float32).ravel()
    out = L.BasicStats()
    L.check(L.lib().mmcmc_basic_stats_from(x.ctypes.data_as(_fp), x.size, C.byref(out)), "mmcmc_basic_stats_from")
    return BasicStats(name, out.min, out.median, out.max, out.mean, out.std)


def _sample_args(sample):
    """-> (pointer, is_device, dtype_code, shape, device index, stream, keepalive)"""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(sample, torch.Tensor):
        t = sample.contiguous()
        if t.dtype not in (torch.float32, torch.float64):
            t = t.float()
        code = L.F32 if t.dtype == torch.float32 else L.F64
        if t.is_cuda:
            stream = torch.cuda.current_stream(t.device).cuda_stream
            return t.data_ptr(), 1, code, tuple(t.shape), t.device.index or 0, C.c_void_p(stream), t
        a = t.numpy()
        return a.ctypes.data, 0, code, a.shape, 0, None, a
    a = np.ascontiguousarray(sample)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float32)
    return a.ctypes.data, 0, L.F32 if a.dtype == np.float32 else L.F64, a.shape, 0, None, a


STATS_KERNELS = {"auto": 0, "fft": 1, "tile1": 2, "tile": 3, "mfma": 4, "direct": 5}


def set_kernel(kind: str = "auto") -> None:
    """mmcmc_stats_set_kernel (include/mmcmc.h): which kernel computes the lag sums.  "auto" is the reference's own
    rule (stats.rs:549): direct sums up to 100 draws per half-chain, the power spectrum above."""
    L.check(L.lib().mmcmc_stats_set_kernel(STATS_KERNELS[kind]), "mmcmc_stats_set_kernel")


def set_direct_work_limit(max_lag_products: int = 1 << 46) -> None:
    """mmcmc_stats_set_direct_work_limit: chains x params x (n / 2)^2 the O(n^2) from-global-memory path (half-chains beyond
    131 072 draws, or under set_kernel("direct")) accepts before it refuses; 0 = no limit."""
    L.check(L.lib().mmcmc_stats_set_direct_work_limit(int(max_lag_products)), "mmcmc_stats_set_direct_work_limit")


def split_rhat_mean_ess(sample, device: int | None = None):
    """stats.rs:416-423: sample [chains, n, params] (numpy, or a torch tensor in HBM) -> (rhat[params], ess[params]).
    `rhat` is the reference's sqrt(W / var+) (quirk Q7)."""
    ptr, is_dev, code, shape, dev, stream, keep = _sample_args(sample)
    if len(shape) != 3:
        raise ValueError("sample must be [chains, n, params]")
    c, n, p = shape
    rhat = np.empty(p, dtype=np.float32)
    ess = np.empty(p, dtype=np.float32)
    st = L.lib().mmcmc_split_rhat_mean_ess(ptr, is_dev, code, c, n, p, rhat.ctypes.data_as(_fp),
                                           ess.ctypes.data_as(_fp), dev if device is None else device, stream)
    L.check(st, "mmcmc_split_rhat_mean_ess")
    return rhat, ess


def standard_split_rhat(sample):
    """The conventional split R-hat, sqrt(var+ / W) (as MultiChainTracker::rhat / collect_rhat report it)."""
    rhat, _ = split_rhat_mean_ess(sample)
    return 1.0 / rhat


def run_stats(sample) -> RunStats:
    """RunStats::from (stats.rs:360-371)."""
    rhat, ess = split_rhat_mean_ess(sample)
    return RunStats(basic_stats("ESS", ess), basic_stats("Split R-hat", rhat))


def stats_partials(sample):
    """Device-side sufficient statistics of the local chains (torch CUDA tensor in, torch CUDA tensors out):
    means [2, C, D], ssq [2, C, D] (splitcat order: half index first) and acov_sum [n/2, D]."""
    import torch

    assert sample.is_cuda and sample.dim() == 3
    t = sample.contiguous()
    if t.dtype not in (torch.float32, torch.float64):
        t = t.float()  # like _sample_args: the kernels read 4- or 8-byte floats only (RunStats::from casts to f32, stats.rs:365)
    code = L.F32 if t.dtype == torch.float32 else L.F64
    c, n, d = t.shape
    m = n // 2
    means = torch.empty((2, c, d), dtype=torch.float32, device=t.device)
    ssq = torch.empty_like(means)
    acov = torch.empty((m, d), dtype=torch.float32, device=t.device)
    stream = torch.cuda.current_stream(t.device).cuda_stream
    st = L.lib().mmcmc_stats_partials(t.data_ptr(), code, c, n, d, means.data_ptr(), ssq.data_ptr(), acov.data_ptr(),
                                      t.device.index or 0, C.c_void_p(stream))
    L.check(st, "mmcmc_stats_partials")
    return means, ssq, acov


def stats_finish(means, ssq, acov_sum):
    """Host finish (stats.rs:449-465, :425-427, :509-545) on global statistics: means, ssq [2C, D]; acov_sum [m, D]."""
    mu = np.ascontiguousarray(means, dtype=np.float32).reshape(-1, np.shape(means)[-1])
    sq = np.ascontiguousarray(ssq, dtype=np.float32).reshape(mu.shape)
    ac = np.ascontiguousarray(acov_sum, dtype=np.float32)
    c2, d = mu.shape
    m = ac.shape[0]
    rhat = np.empty(d, dtype=np.float32)
    ess = np.empty(d, dtype=np.float32)
    st = L.lib().mmcmc_stats_finish(mu.ctypes.data_as(_fp), sq.ctypes.data_as(_fp), ac.ctypes.data_as(_fp), c2, m, d,
                                    rhat.ctypes.data_as(_fp), ess.ctypes.data_as(_fp))
    L.check(st, "mmcmc_stats_finish")
    return rhat, ess


def gather_partials(means, ssq, acov_sum, group=None):
    """The only data-path exchange of a multi-GPU run.  Inputs are this rank's tensors ([2, C_local, D], [2, C_local, D],
    [m, D]; CUDA tensors with the nccl(=RCCL) backend, CPU tensors with gloo).  Returns global (means [2C, D],
    ssq [2C, D], acov_sum [m, D]) as numpy, chains ordered by rank -- the same order a single GPU would produce."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    packed = torch.stack([means, ssq], dim=0).contiguous()  # [2(stat), 2(half), C_local, D]
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed, group=group)
    acov = acov_sum.clone()
    dist.all_reduce(acov, op=dist.ReduceOp.SUM, group=group)
    allp = torch.stack(gathered, dim=0)  # [rank, stat, half, C_local, D]
    allp = allp.permute(1, 2, 0, 3, 4).contiguous()  # [stat, half, rank, C_local, D] = global splitcat order
    d = allp.shape[-1]
    g_means = allp[0].reshape(-1, d).cpu().numpy()
    g_ssq = allp[1].reshape(-1, d).cpu().numpy()
    return g_means, g_ssq, acov.cpu().numpy()


def reduce_partials(means, ssq, acov_sum, group=None):
    """The exchange of a multi-GPU run without a gather: two all-reduces (RCCL over xGMI with CUDA tensors, gloo with
    CPU tensors) of D and 2D + m*D numbers.  Inputs are this rank's statistics ([2, C_local, D], [2, C_local, D],
    [m, D]); returns the global (dsum[D], wsum[D] as float64 numpy, acov_sum [m, D] float32 numpy, number of
    half-chains): what `stats_finish_sums` needs.  Sums are taken in f64, so the result equals the gathered one
    (`gather_partials` + `stats_finish`) to f32 rounding whatever the number of ranks."""
    import torch
    import torch.distributed as dist

    m, d = acov_sum.shape
    mu64 = means.reshape(-1, d).double()
    head = torch.cat([mu64.sum(dim=0), torch.tensor([float(mu64.shape[0])], dtype=torch.float64, device=means.device)])
    dist.all_reduce(head, op=dist.ReduceOp.SUM, group=group)  # sum of the half-chain means, number of half-chains
    c2 = int(round(float(head[d].item())))
    overall = (head[:d] / head[d]).float().double()  # the reference forms the overall mean in f32 (stats.rs:452-455)
    dev = (mu64.float() - overall.float()).double()  # ... and the deviations too
    body = torch.cat([(dev * dev).sum(dim=0), (ssq.reshape(-1, d) / float(m)).double().sum(dim=0),
                      acov_sum.double().reshape(-1)])
    dist.all_reduce(body, op=dist.ReduceOp.SUM, group=group)
    body = body.cpu().numpy()
    return body[:d].copy(), body[d:2 * d].copy(), body[2 * d:].astype(np.float32).reshape(m, d), c2


def stats_finish_sums(dsum, wsum, acov_sum, n_half_chains: int):
    """Host finish (stats.rs:459-465, :425-427, :509-545) from the cross-chain sums of `reduce_partials`."""
    ds = np.ascontiguousarray(dsum, dtype=np.float64)
    ws = np.ascontiguousarray(wsum, dtype=np.float64)
    ac = np.ascontiguousarray(acov_sum, dtype=np.float32)
    m, d = ac.shape
    rhat = np.empty(d, dtype=np.float32)
    ess = np.empty(d, dtype=np.float32)
    _dp = C.POINTER(C.c_double)
    st = L.lib().mmcmc_stats_finish_sums(ds.ctypes.data_as(_dp), ws.ctypes.data_as(_dp), ac.ctypes.data_as(_fp),
                                         int(n_half_chains), m, d, rhat.ctypes.data_as(_fp), ess.ctypes.data_as(_fp))
    L.check(st, "mmcmc_stats_finish_sums")
    return rhat, ess


def split_rhat_mean_ess_distributed(sample_local, group=None):
    """split_rhat_mean_ess over chains sharded across ranks (rank r holds global chains [r*C_local, (r+1)*C_local)).
    Every rank returns the same (rhat, ess).  Exchange: `reduce_partials` (two small all-reduces); `gather_partials` +
    `stats_finish` is the variant that reproduces the single-GPU summation order exactly."""
    means, ssq, acov = stats_partials(sample_local)
    dsum, wsum, g_acov, c2 = reduce_partials(means, ssq, acov, group)
    return stats_finish_sums(dsum, wsum, g_acov, c2)


class MultiChainTracker:
    """stats.rs:189-306 on the GPU: per-chain running means, exponentially averaged acceptance indicator, running
    R-hat.  `step(states)` takes what the reference's `step` takes -- the current states [n_chains, n_params] -- or a
    block of consecutive states [n_chains, k, n_params] (numpy, or a torch tensor in HBM: no host copy)."""

    def __init__(self, n_chains: int, n_params: int, device: int = 0):
        self.n_chains, self.n_params, self.device = int(n_chains), int(n_params), device
        self._h = C.c_void_p()
        L.check(L.lib().mmcmc_tracker_create(C.byref(self._h), self.n_chains, self.n_params, device), "mmcmc_tracker_create")

    @classmethod
    def _adopt(cls, handle: C.c_void_p) -> "MultiChainTracker":
        """Own a tracker the library handed back (mmcmc_*_run_progress' tracker_out)."""
        self = cls.__new__(cls)
        self._h = handle
        c, d, dev = C.c_size_t(), C.c_size_t(), C.c_int()
        L.check(L.lib().mmcmc_tracker_shape(handle, C.byref(c), C.byref(d), C.byref(dev)), "mmcmc_tracker_shape")
        self.n_chains, self.n_params, self.device = int(c.value), int(d.value), int(dev.value)
        return self

    def within_var(self):
        """withinvar_from_cs (stats.rs:155-178) over the per-chain ChainStats: (within [n_params], var [n_params])"""
        w = np.zeros(self.n_params, dtype=np.float32)
        v = np.zeros(self.n_params, dtype=np.float32)
        L.check(L.lib().mmcmc_tracker_within_var(self._h, w.ctypes.data_as(_fp), v.ctypes.data_as(_fp), None),
                "mmcmc_tracker_within_var")
        return w, v

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib().mmcmc_tracker_destroy(h)
            except Exception:
                pass
            self._h = None

    def step(self, states, t0: int = 0, k: int | None = None) -> "MultiChainTracker":
        ptr, is_dev, code, shape, dev, stream, keep = _sample_args(states)
        if len(shape) == 2:
            shape = (shape[0], 1, shape[1])
        if len(shape) != 3 or shape[0] != self.n_chains or shape[2] != self.n_params:
            raise ValueError(f"states must be [{self.n_chains}, k, {self.n_params}]")
        k = shape[1] - t0 if k is None else k
        L.check(L.lib().mmcmc_tracker_steps(self._h, C.c_void_p(ptr), is_dev, code, shape[1], t0, k, stream),
                "mmcmc_tracker_steps")
        return self

    def init_last(self, states) -> "MultiChainTracker":
        """ChainTracker::new(n_params, initial_state) for every chain (stats.rs:60-82): the state the first step is
        compared with, without taking a step.  Use with `chain_stats()` for the generic run_progress' numbers."""
        ptr, is_dev, code, shape, dev, stream, keep = _sample_args(states)
        if tuple(shape) != (self.n_chains, self.n_params):
            raise ValueError(f"states must be [{self.n_chains}, {self.n_params}]")
        L.check(L.lib().mmcmc_tracker_init_last(self._h, C.c_void_p(ptr), is_dev, code, stream), "mmcmc_tracker_init_last")
        return self

    def chain_stats(self):
        """(collect_rhat over the chains' ChainStats [n_params] (stats.rs:150-178), its maximum, mean per-chain p_accept)"""
        r = np.zeros(self.n_params, dtype=np.float32)
        mx, p = C.c_float(), C.c_float()
        L.check(L.lib().mmcmc_tracker_chain_stats(self._h, r.ctypes.data_as(_fp), C.byref(mx), C.byref(p), None),
                "mmcmc_tracker_chain_stats")
        return r, np.float32(mx.value), np.float32(p.value)

    def _stats(self):
        r = np.zeros(self.n_params, dtype=np.float32)
        mx, p = C.c_float(), C.c_float()
        L.check(L.lib().mmcmc_tracker_stats(self._h, r.ctypes.data_as(_fp), C.byref(mx), C.byref(p), None),
                "mmcmc_tracker_stats")
        return r, np.float32(mx.value), np.float32(p.value)

    def rhat(self) -> np.ndarray:
        """stats.rs:280-286"""
        return self._stats()[0]

    def max_rhat(self) -> np.float32:
        """stats.rs:270-274"""
        return self._stats()[1]

    @property
    def p_accept(self) -> np.float32:
        return self._stats()[2]

    @property
    def n(self) -> int:
        n = C.c_uint64()
        L.check(L.lib().mmcmc_tracker_n(self._h, C.byref(n)), "mmcmc_tracker_n")
        return int(n.value)


def ess_from_chainstats(sample, tracker: MultiChainTracker) -> np.ndarray:
    """stats::ess_from_chainstats (stats.rs:668-671): the un-split ESS of sample [chains, n, params] with within / var
    from the per-chain trackers (`tracker` fed with init_last + step, or the one run_progress hands back)."""
    ptr, is_dev, code, shape, dev, stream, keep = _sample_args(sample)
    if len(shape) != 3:
        raise ValueError("sample must be [chains, n, params]")
    c, n, p = shape
    ess = np.empty(p, dtype=np.float32)
    st = L.lib().mmcmc_ess_from_chainstats(ptr, is_dev, code, c, n, p, tracker._h, ess.ctypes.data_as(_fp),
                                           dev if is_dev else tracker.device, stream)
    L.check(st, "mmcmc_ess_from_chainstats")
    return ess


def _run_stats_from_c(rs) -> RunStats:
    def b(name, x):
        return BasicStats(name, x.min, x.median, x.max, x.mean, x.std)

    return RunStats(b("ESS", rs.ess), b("Split R-hat", rs.rhat))
