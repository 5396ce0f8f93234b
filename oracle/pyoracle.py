"""ctypes bindings of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

# target kinds (oracle/targets.h)
GAUSSIAN2D, DIFFABLE_GAUSSIAN2D, ISOTROPIC_GAUSSIAN, ROSENBROCK2D, ROSENBROCK_ND, STANDARD_NORMAL, GAUSSIAN_ND = range(7)


class OTarget(C.Structure):
    _fields_ = [("kind", C.c_int), ("dim", C.c_int), ("p", C.c_double * 8), ("mat", C.POINTER(C.c_double))]


class ORng(C.Structure):
    _fields_ = [
        ("normal_f64", C.c_void_p),
        ("exp1_f64", C.c_void_p),
        ("uniform_f64", C.c_void_p),
        ("uniform_f32", C.c_void_p),
        ("accept_uniform", C.c_void_p),
        ("begin_iter", C.c_void_p),
        ("is_f32", C.c_int),
        ("rc", C.c_uint64 * 4),
        ("seed", C.c_uint64),
        ("chain", C.c_uint64),
        ("iteration", C.c_uint64),
        ("draw", C.c_uint32),
        ("n_normal", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def _stale() -> bool:
    try:
        t = os.path.getmtime(_LIB_PATH)
        t = min(t, os.path.getmtime(os.path.join(_HERE, "_build", "libengine_host.so")))
        deps = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".cpp", ".h", ".inc"))]
        csrc = os.path.join(_HERE, "..", "mini_mcmc_amd", "csrc")
        deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("mm_") and f.endswith(".h")]
        return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))
    except OSError:
        return True


_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    tp = C.POINTER(OTarget)
    sig = {
        "rc_seed_from_u64": (None, [_u64p, C.c_uint64]),
        "rc_next_u64": (C.c_uint64, [_u64p]),
        "rc_next_u32": (C.c_uint32, [_u64p]),
        "rc_uniform_f64": (C.c_double, [_u64p]),
        "rc_uniform_f32": (C.c_float, [_u64p]),
        "rc_open01_f64": (C.c_double, [_u64p]),
        "rc_standard_normal": (C.c_double, [_u64p]),
        "rc_exp1": (C.c_double, [_u64p]),
        "rc_zig_norm_x": (_dp, []),
        "rc_zig_norm_f": (_dp, []),
        "rc_zig_exp_x": (_dp, []),
        "rc_zig_exp_f": (_dp, []),
        "o_philox4x32_10": (None, [_u32p, _u32p, _u32p]),
        "o_engine_block": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _u32p]),
        "o_engine_normal_f32": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "o_engine_icdf24_words": (None, [_u32p, C.c_size_t, _fp]),
        "o_engine_icdf24_independent_words": (None, [_u32p, C.c_size_t, _fp]),
        "o_ndtri": (C.c_double, [C.c_double]),
        "o_engine_normal_f64": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "o_engine_normal_f64_libm": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "o_engine_accept_f32": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint32]),
        "o_engine_mhp_normal_f32": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "o_engine_mhp_accept_f32": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint32]),
        "o_engine_aux_u53": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "ot_logp_f32": (C.c_float, [tp, _fp]),
        "ot_logp_f64": (C.c_double, [tp, _dp]),
        "ot_logp_grad_f32": (C.c_float, [tp, _fp, _fp]),
        "ot_logp_grad_f64": (C.c_double, [tp, _dp, _dp]),
        "ot_make_diffable_gaussian2d": (None, [tp, _dp, _dp, C.c_int]),
        "o_init_with_seed": (None, [C.c_size_t, C.c_size_t, C.c_uint64, _dp]),
        "o_mh_create": (C.c_void_p, [tp, C.c_double, _dp, C.c_int, C.c_int]),
        "o_mh_destroy": (None, [C.c_void_p]),
        "o_mh_proposal_seed": (None, [C.c_void_p, C.c_uint64]),
        "o_mh_seed": (None, [C.c_void_p, C.c_uint64]),
        "o_mh_use_engine_stream": (None, [C.c_void_p, C.c_uint64, C.c_uint64]),
        "o_mh_set_proposal": (None, [C.c_void_p, C.c_int]),
        "o_mh_run": (None, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, _dp, _u64p]),
        "o_mh_state": (None, [C.c_void_p, _dp]),
        "o_hmc_create": (C.c_void_p, [tp, _dp, C.c_int, C.c_double, C.c_int, C.c_int]),
        "o_hmc_destroy": (None, [C.c_void_p]),
        "o_hmc_seed_global": (None, [C.c_void_p, C.c_uint64]),
        "o_hmc_use_engine_stream": (None, [C.c_void_p, C.c_uint64, C.c_uint64]),
        "o_hmc_seed_blocked": (None, [C.c_void_p, C.c_uint64]),
        "o_hmc_run": (None, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, _dp, _u64p]),
        "o_hmc_state": (None, [C.c_void_p, _dp]),
        "o_nuts_create": (C.c_void_p, [tp, _dp, C.c_int, C.c_double, C.c_int]),
        "o_nuts_destroy": (None, [C.c_void_p]),
        "o_nuts_set_chain_seed": (None, [C.c_void_p, C.c_int, C.c_uint64]),
        "o_nuts_set_seed": (None, [C.c_void_p, C.c_uint64]),
        "o_nuts_use_engine_stream": (None, [C.c_void_p, C.c_uint64, C.c_uint64]),
        "o_nuts_set_max_depth": (None, [C.c_void_p, C.c_int]),
        "o_nuts_run": (None, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _dp]),
        "o_nuts_chain_state": (None, [C.c_void_p, C.c_int, _dp]),
        "o_nuts_depth_hist": (None, [C.c_void_p, C.POINTER(C.c_ulonglong), _dp, _dp]),
        "o_nuts_find_reasonable_epsilon": (C.c_double, [tp, _dp, _dp, C.c_int]),
        "o_nuts_build_tree": (
            None,
            [tp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_int, _dp, _dp],
        ),
        "o_discrete_create": (C.c_void_p, [C.c_int, _dp, C.POINTER(C.c_int32), C.c_int]),
        "o_discrete_destroy": (None, [C.c_void_p]),
        "o_discrete_seed": (None, [C.c_void_p, C.c_uint64]),
        "o_discrete_proposal_seed": (None, [C.c_void_p, C.c_uint64]),
        "o_discrete_use_engine_stream": (None, [C.c_void_p, C.c_uint64, C.c_uint64]),
        "o_discrete_run": (None, [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), _u64p]),
        "o_discrete_state": (None, [C.c_void_p, C.POINTER(C.c_int32)]),
        "o_gibbs_create": (C.c_void_p, [_dp, _dp, C.c_int, C.c_uint64]),
        "o_gibbs_destroy": (None, [C.c_void_p]),
        "o_gibbs_use_engine_stream": (None, [C.c_void_p, C.c_uint64, C.c_uint64]),
        "o_gibbs_run": (None, [C.c_void_p, C.c_size_t, C.c_size_t, _dp]),
        "o_gibbs_state": (None, [C.c_void_p, _dp]),
        "o_split_rhat_mean_ess": (None, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp]),
        "o_autocov_bf": (None, [_fp, C.c_size_t, C.c_size_t, _fp]),
        "o_autocov_fft": (None, [_fp, C.c_size_t, C.c_size_t, _fp]),
        "o_basic_stats": (None, [_fp, C.c_size_t, _fp]),
        "o_multichain_tracker": (None, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp]),
        "o_chain_trackers_rhat": (None, [_fp, _fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp]),
        "o_ess_from_chainstats": (None, [_fp, C.c_size_t, C.c_size_t, C.c_size_t, _fp, _fp, C.c_size_t, _fp]),
        "o_cpu_bench": (C.c_double, [C.c_int, tp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_size_t,
                                     C.c_size_t, C.c_int, C.c_uint64, C.c_int, C.c_int, _dp, _dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def default_threads() -> int:
    return usable_cpus()


def usable_cpus() -> int:
    """Hardware threads this process may actually run on: the scheduler affinity mask, capped by the cgroup's CPU quota
    (cpu.max, cgroup v2; cfs_quota_us / cfs_period_us, v1) -- os.cpu_count() reports the machine's, not ours."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


CPU_BENCH_HMC, CPU_BENCH_MH, CPU_BENCH_NUTS = 0, 1, 2


def cpu_bench(kind, target: "Target", init, chains_per_thread, n_threads, scale, n_leapfrog=0, dtype=np.float32,
              n_collect=400, n_discard=50, reps=1, seed=42, nuts_mode=2, nuts_max_depth=10):
    """oracle/cpu_bench.c: ONE spawn, every thread a private sampler over its own block of `chains_per_thread` chains,
    `reps` x run(n_collect, n_discard) each behind a common gate.  Returns (wall seconds, per-thread seconds, NUTS
    leapfrog steps)."""
    init = np.ascontiguousarray(init, dtype=np.float64)
    assert init.shape == (chains_per_thread * n_threads, target.dim), init.shape
    per = np.zeros(n_threads)
    work = C.c_double(0.0)
    wall = lib().o_cpu_bench(int(kind), target.ref, _d(init), int(chains_per_thread), int(n_threads), float(scale),
                             int(n_leapfrog), int(dtype == np.float32), int(n_collect), int(n_discard), int(reps),
                             int(seed), int(nuts_mode), int(nuts_max_depth), _d(per), C.byref(work))
    if wall < 0:
        raise ValueError("o_cpu_bench failed")
    return wall, per, work.value


# ---------------------------------------------------------------- RNG


class SmallRng:
    """rand 0.9 SmallRng (xoshiro256++) + rand_distr 0.5 StandardNormal/Exp1 -- see oracle/rand_compat.c."""

    def __init__(self, seed: int):
        self._s = (C.c_uint64 * 4)()
        lib().rc_seed_from_u64(self._s, seed)

    @property
    def state(self):
        return [int(x) for x in self._s]

    def next_u64(self):
        return int(lib().rc_next_u64(self._s))

    def next_u32(self):
        return int(lib().rc_next_u32(self._s))

    def f64(self):
        return float(lib().rc_uniform_f64(self._s))

    def f32(self):
        return np.float32(lib().rc_uniform_f32(self._s))

    def normal(self):
        return float(lib().rc_standard_normal(self._s))

    def exp1(self):
        return float(lib().rc_exp1(self._s))


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().o_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def engine_normals_f32(seed, chain, iteration, n):
    L = lib()
    return np.array([L.o_engine_normal_f32(seed, chain, iteration, i) for i in range(n)], dtype=np.float32)


def engine_icdf24(words):
    """The oracle's restatement of the engine's f32 normal, one per Philox word."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(w.shape, dtype=np.float32)
    lib().o_engine_icdf24_words(w.ctypes.data_as(_u32p), w.size, out.ctypes.data_as(_fp))
    return out


def engine_icdf24_independent(words):
    """The f32 normal of Philox words WITHOUT the product's coefficient table: f64 inverse normal CDF (Halley on libm's
    erfc, oracle/orng.c) rounded to f32."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(w.shape, dtype=np.float32)
    lib().o_engine_icdf24_independent_words(w.ctypes.data_as(_u32p), w.size, _f(out))
    return out


def engine_normals_f64(seed, chain, iteration, n):
    L = lib()
    return np.array([L.o_engine_normal_f64(seed, chain, iteration, i) for i in range(n)], dtype=np.float64)


def engine_accept_f32(seed, chain, iteration):
    return np.float32(lib().o_engine_accept_f32(seed, chain, iteration))


def engine_mhp_noise_f32(seed, chain, iteration, dim):
    """The MH sampler's paired f32 stream at dim <= 2 (csrc/mm_rng.h, oracle/orng.c: o_engine_mhp_*): (z[dim], u)."""
    L = lib()
    return (np.array([L.o_engine_mhp_normal_f32(seed, chain, iteration, i) for i in range(dim)], dtype=np.float32),
            np.float32(L.o_engine_mhp_accept_f32(seed, chain, iteration)))


def engine_aux_u53(seed, chain, iteration, k):
    return float(lib().o_engine_aux_u53(seed, chain, iteration, k))


# ---------------------------------------------------------------- targets


class Target:
    """POD target descriptor (oracle/targets.h)."""

    def __init__(self, kind, dim, params=(), mat=None):
        self.t = OTarget()
        self.t.kind = kind
        self.t.dim = dim
        for i, v in enumerate(params):
            self.t.p[i] = float(v)
        self._mat = None
        if mat is not None:
            self._mat = np.ascontiguousarray(mat, dtype=np.float64)
            assert self._mat.shape == (dim, dim)
            self.t.mat = _d(self._mat)
        self.kind, self.dim = kind, dim

    @property
    def ref(self):
        return C.byref(self.t)

    def logp(self, x, dtype=np.float64):
        x = np.ascontiguousarray(x, dtype=dtype)
        if dtype == np.float32:
            return np.float32(lib().ot_logp_f32(self.ref, _f(x)))
        return float(lib().ot_logp_f64(self.ref, _d(x)))

    def logp_grad(self, x, dtype=np.float64):
        x = np.ascontiguousarray(x, dtype=dtype)
        g = np.zeros_like(x)
        if dtype == np.float32:
            lp = np.float32(lib().ot_logp_grad_f32(self.ref, _f(x), _f(g)))
        else:
            lp = float(lib().ot_logp_grad_f64(self.ref, _d(x), _d(g)))
        return lp, g


def gaussian2d(mean, cov):
    cov = np.asarray(cov, dtype=np.float64).reshape(4)
    return Target(GAUSSIAN2D, 2, [mean[0], mean[1], cov[0], cov[1], cov[2], cov[3]])


def diffable_gaussian2d(mean, cov, t_is_f32=False):
    t = Target(DIFFABLE_GAUSSIAN2D, 2)
    m = np.asarray(mean, dtype=np.float64)
    c = np.asarray(cov, dtype=np.float64).reshape(4)
    lib().ot_make_diffable_gaussian2d(t.ref, _d(m), _d(c), int(t_is_f32))
    return t


def isotropic_gaussian(std, dim):
    return Target(ISOTROPIC_GAUSSIAN, dim, [std])


def rosenbrock2d(a, b):
    return Target(ROSENBROCK2D, 2, [a, b])


def rosenbrock_nd(dim):
    return Target(ROSENBROCK_ND, dim)


def standard_normal(dim):
    return Target(STANDARD_NORMAL, dim)


def gaussian_nd(precision):
    a = np.asarray(precision, dtype=np.float64)
    return Target(GAUSSIAN_ND, a.shape[0], mat=a)


# ---------------------------------------------------------------- init


def init_with_seed(n, d, seed, dtype=np.float64):
    """core.rs:413-435. init_det(n, d) = init_with_seed(n, d, 42)."""
    out = np.zeros((n, d), dtype=np.float64)
    lib().o_init_with_seed(n, d, seed, _d(out))
    return out.astype(dtype)


def init_det(n, d, dtype=np.float64):
    return init_with_seed(n, d, 42, dtype)


# ---------------------------------------------------------------- samplers


class MetropolisHastings:
    """metropolis_hastings.rs:149-193 + core.rs:176-186 with IsotropicGaussian proposal."""

    def __init__(self, target: Target, proposal_std, init, dtype=np.float64, proposal_seed=None):
        init = np.ascontiguousarray(init, dtype=np.float64)
        self.n_chains, self.dim = init.shape
        self.dtype = dtype
        self._target = target
        self._h = lib().o_mh_create(target.ref, float(proposal_std), _d(init), self.n_chains, int(dtype == np.float32))
        if not self._h:
            raise ValueError("o_mh_create failed")
        if proposal_seed is not None:
            lib().o_mh_proposal_seed(self._h, proposal_seed)

    def seed(self, seed):
        lib().o_mh_seed(self._h, seed)
        return self

    def use_engine_stream(self, seed, chain_offset=0):
        lib().o_mh_use_engine_stream(self._h, seed, chain_offset)
        return self

    def set_proposal(self, kind):
        """0: IsotropicGaussian (the reference's); 1: the asymmetric custom proposal of oracle/mh_hmc.c (multiplicative
        log-normal random walk), for the parity test of user-defined proposals."""
        lib().o_mh_set_proposal(self._h, int(kind))
        return self

    def run(self, n_collect, n_discard, n_threads=None, want_out=True):
        out = np.zeros((self.n_chains, n_collect, self.dim), dtype=np.float64) if want_out else None
        acc = np.zeros(self.n_chains, dtype=np.uint64)
        lib().o_mh_run(
            self._h, n_collect, n_discard, n_threads or default_threads(), _d(out) if want_out else None,
            acc.ctypes.data_as(_u64p)
        )
        self.accept_counts = acc
        return out.astype(self.dtype) if want_out else None

    def state(self):
        out = np.zeros((self.n_chains, self.dim))
        lib().o_mh_state(self._h, _d(out))
        return out.astype(self.dtype)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().o_mh_destroy(self._h)
            self._h = None


class HMC:
    """hmc.rs:87-158, 304-431."""

    def __init__(self, target: Target, init, step_size, n_leapfrog, dtype=np.float32):
        init = np.ascontiguousarray(init, dtype=np.float64)
        self.n_chains, self.dim = init.shape
        self.dtype = dtype
        self._target = target
        self._h = lib().o_hmc_create(
            target.ref, _d(init), self.n_chains, float(step_size), int(n_leapfrog), int(dtype == np.float32)
        )
        if not self._h:
            raise ValueError("o_hmc_create failed")

    def seed_global(self, seed):
        lib().o_hmc_seed_global(self._h, seed)
        return self

    def seed_blocked(self, seed):
        """one rand-compatible stream per thread-block of chains: the CPU-baseline timing mode"""
        lib().o_hmc_seed_blocked(self._h, seed)
        return self

    def use_engine_stream(self, seed, chain_offset=0):
        lib().o_hmc_use_engine_stream(self._h, seed, chain_offset)
        return self

    def run(self, n_collect, n_discard, n_threads=None, want_out=True):
        out = np.zeros((self.n_chains, n_collect, self.dim), dtype=np.float64) if want_out else None
        acc = np.zeros(self.n_chains, dtype=np.uint64)
        lib().o_hmc_run(
            self._h, n_collect, n_discard, n_threads or default_threads(), _d(out) if want_out else None,
            acc.ctypes.data_as(_u64p)
        )
        self.accept_counts = acc
        return out.astype(self.dtype) if want_out else None

    def state(self):
        out = np.zeros((self.n_chains, self.dim))
        lib().o_hmc_state(self._h, _d(out))
        return out.astype(self.dtype)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().o_hmc_destroy(self._h)
            self._h = None


class NUTS:
    """nuts.rs:123-170, 347-353 (multi-chain) over NUTSChain nuts.rs:410-691.

    mode 0: f32 tensors + f64 scalars (reference T=f64); 1: f32/f32 (reference T=f32); 2: f64/f64."""

    def __init__(self, target: Target, init, target_accept_p, mode=0):
        init = np.ascontiguousarray(init, dtype=np.float64)
        if init.ndim == 1:
            init = init[None, :]
        self.n_chains, self.dim = init.shape
        self.mode = mode
        self._target = target
        self._h = lib().o_nuts_create(target.ref, _d(init), self.n_chains, float(target_accept_p), mode)
        if not self._h:
            raise ValueError("o_nuts_create failed")

    def set_seed(self, seed):
        lib().o_nuts_set_seed(self._h, seed)
        return self

    def set_chain_seed(self, chain, seed):
        lib().o_nuts_set_chain_seed(self._h, chain, seed)
        return self

    def use_engine_stream(self, seed, chain_offset=0):
        lib().o_nuts_use_engine_stream(self._h, seed, chain_offset)
        return self

    def set_max_depth(self, max_depth):
        """0 = unbounded (the reference); > 0 caps the doublings per transition (the GPU engine needs a cap)."""
        lib().o_nuts_set_max_depth(self._h, int(max_depth))
        return self

    def run(self, n_collect, n_discard, progress=False, n_threads=None):
        out = np.zeros((self.n_chains, n_collect, self.dim), dtype=np.float64)
        lib().o_nuts_run(self._h, n_collect, n_discard, int(progress), n_threads or default_threads(), _d(out))
        return out

    def chain_state(self, chain=0):
        out = np.zeros(7)
        lib().o_nuts_chain_state(self._h, chain, _d(out))
        keys = ["epsilon", "epsilon_bar", "h_bar", "mu", "m", "last_depth", "n_leapfrog_total"]
        return dict(zip(keys, out.tolist()))

    def depth_histogram(self):
        """(transitions of all chains by tree depth [16], epsilon [C], epsilon_bar [C]) -- diagnostics, not in the reference."""
        h = np.zeros(16, dtype=np.uint64)
        eps = np.zeros(self.n_chains)
        ebar = np.zeros(self.n_chains)
        lib().o_nuts_depth_hist(self._h, h.ctypes.data_as(C.POINTER(C.c_ulonglong)), _d(eps), _d(ebar))
        return h, eps, ebar

    def __del__(self):
        if getattr(self, "_h", None):
            lib().o_nuts_destroy(self._h)
            self._h = None


def find_reasonable_epsilon(target: Target, position, mom, mode=0):
    p = np.ascontiguousarray(position, dtype=np.float64)
    m = np.ascontiguousarray(mom, dtype=np.float64)
    return float(lib().o_nuts_find_reasonable_epsilon(target.ref, _d(p), _d(m), mode))


def build_tree(target: Target, position, mom, grad, logu, v, j, epsilon, joint_0, rng_seed=0, mode=0):
    d = target.dim
    p = np.ascontiguousarray(position, dtype=np.float64)
    m = np.ascontiguousarray(mom, dtype=np.float64)
    g = np.ascontiguousarray(grad, dtype=np.float64)
    vec = np.zeros((8, d))
    sc = np.zeros(5)
    lib().o_nuts_build_tree(
        target.ref, _d(p), _d(m), _d(g), float(logu), int(v), int(j), float(epsilon), float(joint_0), rng_seed, mode,
        _d(vec), _d(sc)
    )
    names = ["position_minus", "mom_minus", "grad_minus", "position_plus", "mom_plus", "grad_plus", "position_prime",
             "grad_prime"]
    out = {k: vec[i].copy() for i, k in enumerate(names)}
    out.update(logp_prime=sc[0], n_prime=int(sc[1]), s_prime=bool(sc[2]), alpha_prime=sc[3], n_alpha_prime=int(sc[4]))
    return out


# ---------------------------------------------------------------- diagnostics


def split_rhat_mean_ess(sample, n_threads: int = 0):
    """stats.rs:416-423 on [chains, n, params] -> (rhat[params], ess[params]); 'rhat' is sqrt(W/var+) (Q7).
    n_threads > 1 computes the per-chain autocovariances on that many threads (0: all cores for large samples) and adds
    them up in chain order as the serial loop does: the result is bit-identical."""
    s = np.ascontiguousarray(sample, dtype=np.float32)
    if n_threads == 0:
        n_threads = (os.cpu_count() or 1) if s.size >= (1 << 22) else 1
    lib().o_stats_set_threads(int(n_threads))
    c, n, p = s.shape
    rhat = np.zeros(p, dtype=np.float32)
    ess = np.zeros(p, dtype=np.float32)
    lib().o_split_rhat_mean_ess(_f(s), c, n, p, _f(rhat), _f(ess))
    return rhat, ess


def autocov_bf(data):
    x = np.ascontiguousarray(data, dtype=np.float32)
    out = np.zeros_like(x)
    lib().o_autocov_bf(_f(x), x.shape[0], x.shape[1], _f(out))
    return out


def autocov_fft(data):
    x = np.ascontiguousarray(data, dtype=np.float32)
    out = np.zeros_like(x)
    lib().o_autocov_fft(_f(x), x.shape[0], x.shape[1], _f(out))
    return out


def basic_stats(data):
    x = np.ascontiguousarray(data, dtype=np.float32)
    out = np.zeros(5, dtype=np.float32)
    lib().o_basic_stats(_f(x), x.size, _f(out))
    return dict(min=out[0], median=out[1], max=out[2], mean=out[3], std=out[4])


def run_stats(sample):
    """RunStats::from stats.rs:360-371."""
    rhat, ess = split_rhat_mean_ess(sample)
    return dict(ess=basic_stats(ess), rhat=basic_stats(rhat))


def multichain_tracker_rhat(states):
    """MultiChainTracker stats.rs:189-306 fed states [steps, chains, params]."""
    s = np.ascontiguousarray(states, dtype=np.float32)
    steps, chains, params = s.shape
    rhat = np.zeros(params, dtype=np.float32)
    pa = C.c_float()
    lib().o_multichain_tracker(_f(s), steps, chains, params, _f(rhat), C.byref(pa))
    return rhat, np.float32(pa.value)


def chain_trackers_rhat(init, states):
    """ChainTracker + collect_rhat (stats.rs:26-178): init [chains, params], states [chains, steps, params]."""
    i0 = np.ascontiguousarray(init, dtype=np.float32)
    s = np.ascontiguousarray(states, dtype=np.float32)
    chains, steps, params = s.shape
    rhat = np.zeros(params, dtype=np.float32)
    pa = np.zeros(chains, dtype=np.float32)
    lib().o_chain_trackers_rhat(_f(i0), _f(s), chains, steps, params, _f(rhat), _f(pa))
    return rhat, pa


# ---------------------------------------------------------------- host build of the engine's own arithmetic

_EH_PATH = os.path.join(_HERE, "_build", "libengine_host.so")
_eh = None


def ess_from_chainstats(sample, init, tracked):
    """stats.rs:668-671: un-split ESS of sample [chains, n, params] with within / var from per-chain ChainTrackers
    constructed on init [chains, params] and fed tracked [chains, steps, params]."""
    s = np.ascontiguousarray(sample, dtype=np.float32)
    i0 = np.ascontiguousarray(init, dtype=np.float32)
    tr = np.ascontiguousarray(tracked, dtype=np.float32)
    c, n, p = s.shape
    assert i0.shape == (c, p) and tr.shape[0] == c and tr.shape[2] == p
    ess = np.zeros(p, dtype=np.float32)
    lib().o_ess_from_chainstats(_f(s), c, n, p, _f(i0), _f(tr), tr.shape[1], _f(ess))
    return ess


def engine_host_lib() -> C.CDLL:
    """oracle/engine_host.cpp: the product's transition headers compiled for the host (bit-exact GPU<->CPU)."""
    global _eh
    if _eh is not None:
        return _eh
    build()
    if not os.path.exists(_EH_PATH):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    E = C.CDLL(_EH_PATH)
    E.eh_run.restype = C.c_int
    E.eh_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_void_p, C.c_size_t,
                         C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_size_t, C.c_void_p, _u64p, C.c_int]
    E.eh_nuts_run.restype = C.c_int
    E.eh_nuts_run.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_size_t, C.c_double, C.c_uint64, C.c_uint64,
                              C.c_uint32, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _dp, C.c_void_p, _u64p, C.c_int]
    E.eh_noise.restype = C.c_int
    E.eh_noise.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    E.eh_ratio_filter_disagreements.restype = C.c_size_t
    E.eh_ratio_filter_disagreements.argtypes = [_dp, _dp, C.c_size_t]
    E.eh_lnu_f32.restype = C.c_int
    E.eh_lnu_f32.argtypes = [_fp, C.c_size_t, _fp]
    E.eh_icdf24_words.restype = C.c_int
    E.eh_icdf24_words.argtypes = [_u32p, C.c_size_t, _fp]
    E.eh_discrete_run.restype = C.c_int
    E.eh_discrete_run.argtypes = [C.c_int, _dp, C.POINTER(C.c_int32), C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32,
                                  C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u64p]
    E.eh_gibbs_mixture_run.restype = C.c_int
    E.eh_gibbs_mixture_run.argtypes = [_dp, _dp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_size_t,
                                       _dp, _dp]
    E.eh_hmc_grouped_run.restype = C.c_int
    E.eh_hmc_grouped_run.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _dp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32,
                                     C.c_size_t, C.c_size_t, _dp, _u64p]
    E.eh_hmc_grouped_run_f32.restype = C.c_int
    E.eh_hmc_grouped_run_f32.argtypes = [C.c_int, _fp, C.c_float, C.c_int, _fp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32,
                                         C.c_size_t, C.c_size_t, _fp, _u64p]
    E.eh_logp_grad.restype = C.c_int
    E.eh_logp_grad.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    _eh = E
    return E


def _eh_target_args(kind, params, matrix):
    p = np.zeros(8, dtype=np.float64)
    p[: len(params)] = params
    m = None if matrix is None else np.ascontiguousarray(matrix, dtype=np.float64)
    return p, m


def engine_host_run(sampler, kind, dim, params, init, scale, n_collect, n_discard, seed=0, chain_offset=0, iter0=0,
                    n_leapfrog=0, matrix=None, dtype=np.float32, n_threads=None, want_out=True):
    """Run chains on the host with the ENGINE's arithmetic and stream. sampler: 'mh' | 'hmc'.
    `params` are the C-ABI description parameters (mean/cov, std, a/b ...: include/mmcmc.h).
    Returns (sample [C, n_collect, D] or None, final_state [C, D], accept_counts [C])."""
    E = engine_host_lib()
    state = np.ascontiguousarray(init, dtype=dtype).copy()
    n = state.shape[0]
    out = np.empty((n, n_collect, dim), dtype=dtype) if want_out else None
    acc = np.zeros(n, dtype=np.uint64)
    p, m = _eh_target_args(kind, params, matrix)
    code = {"mh": 0, "hmc": 1, "mh_generic": 2, "hmc_generic": 3}[sampler]  # *_generic: mm_generic.h whatever the dimension
    rc = E.eh_run(code, 0 if dtype == np.float32 else 1, kind, dim, _d(p),
                  _d(m) if m is not None else None, float(scale), int(n_leapfrog), state.ctypes.data, n, seed,
                  chain_offset, iter0, n_collect, n_discard, out.ctypes.data if want_out else None,
                  acc.ctypes.data_as(_u64p), n_threads or default_threads())
    if rc != 0:
        raise ValueError(f"eh_run: {rc}")
    return out, state, acc


def engine_host_noise(seed, chain_offset, iteration, n, dim, dtype=np.float32):
    E = engine_host_lib()
    z = np.empty((n, dim), dtype=dtype)
    u = np.empty(n, dtype=dtype)
    rc = E.eh_noise(0 if dtype == np.float32 else 1, seed, chain_offset, iteration, n, dim, z.ctypes.data, u.ctypes.data)
    if rc != 0:
        raise ValueError(f"eh_noise: {rc}")
    return z, u


def engine_host_ratio_filter_disagreements(ratio, u):
    """How often mm_ratio_exceeds_ln_u (filtered) differs from `ratio > mm_log(u)` (must be never)."""
    r = np.ascontiguousarray(ratio, dtype=np.float64)
    uu = np.ascontiguousarray(u, dtype=np.float64)
    return int(engine_host_lib().eh_ratio_filter_disagreements(_d(r), _d(uu), r.size))


def engine_host_accept_stat(d):
    """(new, old): the product's NUTS acceptance statistic (mm_nuts.h: mm_accept_stat) and the form it replaced, min(1, exp(d)),
    both from the host build of the engine's headers, value by value."""
    x = np.ascontiguousarray(d)
    assert x.dtype in (np.float32, np.float64)
    a, b = np.empty_like(x), np.empty_like(x)
    E = engine_host_lib()
    E.eh_accept_stat.restype = C.c_int
    E.eh_accept_stat.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    E.eh_accept_stat(0 if x.dtype == np.float32 else 1, x.ctypes.data, x.size, a.ctypes.data, b.ctypes.data)
    return a, b


def engine_host_lnu_f32(u):
    """The PRODUCT's ln u of the f32 accept uniform (mm_rng.h: mm_lnu_f32) compiled for the host."""
    x = np.ascontiguousarray(u, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.float32)
    engine_host_lib().eh_lnu_f32(x.ctypes.data_as(_fp), x.size, out.ctypes.data_as(_fp))
    return out


def engine_host_icdf24(words):
    """The PRODUCT's f32 normal (mm_rng.h: mm_icdf_f32) compiled for the host, one per Philox word."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(w.shape, dtype=np.float32)
    engine_host_lib().eh_icdf24_words(w.ctypes.data_as(_u32p), w.size, out.ctypes.data_as(_fp))
    return out


def engine_host_logp_grad(kind, dim, params, x, matrix=None, dtype=np.float32, with_grad=True):
    E = engine_host_lib()
    x = np.ascontiguousarray(x, dtype=dtype)
    lp = np.empty(x.shape[0], dtype=dtype)
    g = np.empty_like(x) if with_grad else None
    p, m = _eh_target_args(kind, params, matrix)
    rc = E.eh_logp_grad(0 if dtype == np.float32 else 1, kind, dim, _d(p), _d(m) if m is not None else None,
                        x.ctypes.data, x.shape[0], lp.ctypes.data, g.ctypes.data if with_grad else None)
    if rc != 0:
        raise ValueError(f"eh_logp_grad: {rc}")
    return lp, g


POISSON_REFLECT, BINOMIAL_CLAMP, POISSON_NONNEG = 16, 17, 18
_i32p = C.POINTER(C.c_int32)


class DiscreteMH:
    """oracle/discrete.c: MetropolisHastings over the reference's discrete test targets (i32 states)."""

    def __init__(self, kind, params, init):
        self.init = np.ascontiguousarray(init, dtype=np.int32).ravel()
        self.n_chains = self.init.size
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._h = lib().o_discrete_create(int(kind), p.ctypes.data_as(_dp), self.init.ctypes.data_as(_i32p), self.n_chains)
        if not self._h:
            raise ValueError("o_discrete_create")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().o_discrete_destroy(self._h)
            self._h = None

    def seed(self, seed):
        lib().o_discrete_seed(self._h, int(seed))
        return self

    def proposal_seed(self, seed):
        lib().o_discrete_proposal_seed(self._h, int(seed))
        return self

    def use_engine_stream(self, seed, chain_offset=0):
        lib().o_discrete_use_engine_stream(self._h, int(seed), int(chain_offset))
        return self

    def run(self, n_collect, n_discard):
        out = np.zeros((self.n_chains, n_collect), dtype=np.int32)
        acc = np.zeros(self.n_chains, dtype=np.uint64)
        lib().o_discrete_run(self._h, n_collect, n_discard, out.ctypes.data_as(_i32p), acc.ctypes.data_as(_u64p))
        self.accept_counts = acc
        return out

    def state(self):
        out = np.zeros(self.n_chains, dtype=np.int32)
        lib().o_discrete_state(self._h, out.ctypes.data_as(_i32p))
        return out


def engine_host_discrete_run(kind, params, init, n_collect, n_discard, seed=0, chain_offset=0, iter0=0):
    """The product's mm_discrete.h on the host: (samples [C, n_collect] i32, final states, accept counts)."""
    E = engine_host_lib()
    init = np.ascontiguousarray(init, dtype=np.int32).ravel()
    p = np.ascontiguousarray(params, dtype=np.float64)
    out = np.zeros((init.size, n_collect), dtype=np.int32)
    st = np.zeros(init.size, dtype=np.int32)
    acc = np.zeros(init.size, dtype=np.uint64)
    rc = E.eh_discrete_run(int(kind), p.ctypes.data_as(_dp), init.ctypes.data_as(_i32p), init.size, int(seed),
                           int(chain_offset), int(iter0), n_collect, n_discard, out.ctypes.data_as(_i32p),
                           st.ctypes.data_as(_i32p), acc.ctypes.data_as(_u64p))
    assert rc == 0
    return out, st, acc


class GibbsMixture:
    """oracle/gibbs.c: GibbsSampler over the reference's MixtureConditional (state [x, z], f64)."""

    def __init__(self, params, init, conditional_seed=0):
        self.init = np.ascontiguousarray(init, dtype=np.float64).reshape(-1, 2)
        self.n_chains = self.init.shape[0]
        p = np.ascontiguousarray(params, dtype=np.float64)
        self._h = lib().o_gibbs_create(_d(p), _d(self.init), self.n_chains, int(conditional_seed))
        if not self._h:
            raise ValueError("o_gibbs_create")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().o_gibbs_destroy(self._h)
            self._h = None

    def use_engine_stream(self, seed, chain_offset=0):
        lib().o_gibbs_use_engine_stream(self._h, int(seed), int(chain_offset))
        return self

    def run(self, n_collect, n_discard):
        out = np.zeros((self.n_chains, n_collect, 2), dtype=np.float64)
        lib().o_gibbs_run(self._h, n_collect, n_discard, _d(out))
        return out

    def state(self):
        out = np.zeros((self.n_chains, 2), dtype=np.float64)
        lib().o_gibbs_state(self._h, _d(out))
        return out


def engine_host_gibbs_mixture_run(params, init, n_collect, n_discard, seed=0, chain_offset=0, iter0=0):
    """The product's mm_gibbs.h on the host: (samples [C, n_collect, 2], final states)."""
    E = engine_host_lib()
    init = np.ascontiguousarray(init, dtype=np.float64).reshape(-1, 2)
    p = np.ascontiguousarray(params, dtype=np.float64)
    out = np.zeros((init.shape[0], n_collect, 2), dtype=np.float64)
    st = np.zeros_like(init)
    rc = E.eh_gibbs_mixture_run(_d(p), _d(init), init.shape[0], int(seed), int(chain_offset), int(iter0), n_collect,
                                n_discard, _d(out), _d(st))
    assert rc == 0
    return out, st


def engine_host_hmc_grouped_run(matrix, init, eps, n_leapfrog, n_collect, n_discard, seed=0, chain_offset=0, iter0=0):
    """HMC on GaussianND (f64, dim 16 / 32) with the lane-group kernel's reduction order: (samples, states, accepts)."""
    E = engine_host_lib()
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    st = np.ascontiguousarray(init, dtype=np.float64).copy()
    n, d = st.shape
    out = np.zeros((n, n_collect, d), dtype=np.float64)
    acc = np.zeros(n, dtype=np.uint64)
    rc = E.eh_hmc_grouped_run(d, _d(m), float(eps), int(n_leapfrog), _d(st), n, int(seed), int(chain_offset), int(iter0),
                              n_collect, n_discard, _d(out), acc.ctypes.data_as(_u64p))
    assert rc == 0
    return out, st, acc


def engine_host_hmc_grouped_run_f32(matrix, init, eps, n_leapfrog, n_collect, n_discard, seed=0, chain_offset=0, iter0=0):
    """The f32 lane-group kernel's twin (blocked coordinate ownership, mm_targets.h): (samples, states, accepts)."""
    E = engine_host_lib()
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    st = np.ascontiguousarray(init, dtype=np.float32).copy()
    n, d = st.shape
    out = np.zeros((n, n_collect, d), dtype=np.float32)
    acc = np.zeros(n, dtype=np.uint64)
    rc = E.eh_hmc_grouped_run_f32(d, m.ctypes.data_as(_fp), C.c_float(eps), int(n_leapfrog), st.ctypes.data_as(_fp), n,
                                  int(seed), int(chain_offset), int(iter0), n_collect, n_discard,
                                  out.ctypes.data_as(_fp), acc.ctypes.data_as(_u64p))
    assert rc == 0
    return out, st, acc


def engine_host_nuts_run(mode, kind, dim, params, init, target_accept_p, n_collect, n_discard, seed=0, chain_offset=0,
                         m0=0, progress=False, max_depth=10, matrix=None, adapt=None, n_threads=None):
    """NUTS on the host with the ENGINE's arithmetic/stream (mm_nuts.h compiled by g++).  Returns
    (sample [C, n_collect, D] of the tensor type, final positions [C, D], adapt [C, 4] f64, leapfrog counts [C])."""
    E = engine_host_lib()
    x = np.ascontiguousarray(init, dtype=np.float64).copy()
    if x.ndim == 1:
        x = x[None, :]
    n = x.shape[0]
    # mode 3: host twin of the lane-group / MFMA kernel (f64); 4, 5, 6: type modes 0, 1, 2 through mm_nuts_generic.h
    tdt = np.float64 if mode in (2, 3, 6) else np.float32
    out = np.empty((n, n_collect, dim), dtype=tdt)
    ad = np.empty((n, 4), dtype=np.float64)
    if adapt is None:
        ad[:] = [-1.0, 1.0, 0.0, np.log(10.0)]
    else:
        ad[:] = adapt
    nlf = np.zeros(n, dtype=np.uint64)
    p, m = _eh_target_args(kind, params, matrix)
    rc = E.eh_nuts_run(mode, kind, dim, _d(p), _d(m) if m is not None else None, _d(x), n, float(target_accept_p), seed,
                       chain_offset, m0, n_collect, n_discard, int(progress), int(max_depth), _d(ad), out.ctypes.data,
                       nlf.ctypes.data_as(_u64p), n_threads or default_threads())
    if rc != 0:
        raise ValueError(f"eh_nuts_run: {rc}")
    return out, x.astype(tdt), ad, nlf
