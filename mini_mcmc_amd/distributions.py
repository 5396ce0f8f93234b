"""Built-in targets and the proposal -- host-side mirror of the reference's `distributions` module.

Each class is a plain descriptor handed to the C ABI (mmcmc_target_desc / mmcmc_proposal_desc); the densities
themselves are evaluated on the GPU (csrc/mm_targets.h).  Reference: src/distributions.rs
  Gaussian2D :158-206   DiffableGaussian2D :212-316   IsotropicGaussian :344-402
  Rosenbrock2D :490-524   RosenbrockND :528-547   (StandardNormal: nuts.rs:1024-1037, test target)
GaussianND (dense precision matrix) is not in the reference; BASELINE.json config 5 needs it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class Target:
    kind: int = -1

    def __init__(self, dim: int, params=(), matrix=None):
        self.dim = int(dim)
        self.params = [float(p) for p in params]
        self._matrix = None if matrix is None else np.ascontiguousarray(matrix, dtype=np.float64)

    def desc(self) -> L.TargetDesc:
        d = L.TargetDesc()
        d.kind = self.kind
        d.dim = self.dim
        for i, p in enumerate(self.params):
            d.params[i] = p
        if self._matrix is not None:
            d.matrix = self._matrix.ctypes.data_as(C.POINTER(C.c_double))
        return d

    def unnorm_logp_batch(self, positions, dtype=np.float32, device: int = 0, with_grad: bool = False):
        """BatchedGradientTarget::unnorm_logp_batch (distributions.rs:65-76) evaluated on the GPU;
        with_grad=True also returns the analytic gradient (GradientTarget::unnorm_logp_and_grad :81-87)."""
        x = np.ascontiguousarray(positions, dtype=dtype)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"positions must be [n, {self.dim}]")
        lp = np.empty(x.shape[0], dtype=dtype)
        g = np.empty_like(x) if with_grad else None
        d = self.desc()
        st = L.lib().mmcmc_logp_grad_batch(
            C.byref(d), L.F32 if dtype == np.float32 else L.F64, x.ctypes.data, x.shape[0], lp.ctypes.data,
            g.ctypes.data if with_grad else None, device)
        L.check(st, "mmcmc_logp_grad_batch")
        return (lp, g) if with_grad else lp


class Gaussian2D(Target):
    """distributions.rs:158-206 `Gaussian2D { mean, cov }` (Target for Metropolis-Hastings)."""
    kind = L.GAUSSIAN2D

    def __init__(self, mean, cov):
        cov = np.asarray(cov, dtype=np.float64).reshape(2, 2)
        self.mean = np.asarray(mean, dtype=np.float64).reshape(2)
        self.cov = cov
        super().__init__(2, [self.mean[0], self.mean[1], cov[0, 0], cov[0, 1], cov[1, 0], cov[1, 1]])


    def logp(self, position) -> float:
        """`Normalized::logp` (distributions.rs:164-187): the fully normalised log-density, on the host in the
        reference's order: -ln(2 pi) - 1/2 ln|det| - 1/2 diff^T Sigma^-1 diff (Sigma^-1 = adj / det)."""
        (a, b), (c, d) = self.cov
        det = a * d - b * c
        term_1 = -np.log(2.0 * np.pi)
        term_2 = -0.5 * np.log(abs(det))
        diff = np.asarray(position, dtype=np.float64).reshape(2) - self.mean
        inv_cov = np.array([[d, -b], [-c, a]]) / det
        term_3 = -0.5 * diff.dot(inv_cov).dot(diff)
        return float(term_1 + term_2 + term_3)


class DiffableGaussian2D(Gaussian2D):
    """distributions.rs:212-316 `DiffableGaussian2D::new(mean, cov)` (gradient target for HMC / NUTS)."""
    kind = L.DIFFABLE_GAUSSIAN2D


class IsotropicGaussian(Target):
    """distributions.rs:344-402 `IsotropicGaussian::new(std)`: the MH proposal, and a target when given `dim`."""
    kind = L.ISOTROPIC_GAUSSIAN

    def __init__(self, std: float, dim: int = 1):
        self.std = float(std)
        super().__init__(dim, [self.std])

    def set_seed(self, seed: int) -> "IsotropicGaussian":
        """Proposal::set_seed (distributions.rs:388-391).  The GPU engine draws proposal noise from the sampler's
        counter-based stream (one key per chain), so the proposal carries no generator of its own: kept for
        call-site compatibility, no effect (the reference's cloned-generator quirk Q1 is not reproduced)."""
        self._seed = int(seed)
        return self

    def proposal_desc(self) -> L.ProposalDesc:
        p = L.ProposalDesc()
        p.kind = 0
        p.std = self.std
        return p


class Rosenbrock2D(Target):
    """distributions.rs:490-524 `Rosenbrock2D { a, b }`."""
    kind = L.ROSENBROCK2D

    def __init__(self, a: float = 1.0, b: float = 100.0):
        self.a, self.b = float(a), float(b)
        super().__init__(2, [self.a, self.b])


class RosenbrockND(Target):
    """distributions.rs:528-547 `RosenbrockND {}`; the dimension comes from the initial positions."""
    kind = L.ROSENBROCK_ND

    def __init__(self, dim: int = 3):
        super().__init__(dim)


class StandardNormal(Target):
    """nuts.rs:1024-1037 (the reference's test target): -1/2 sum x^2."""
    kind = L.STANDARD_NORMAL

    def __init__(self, dim: int):
        super().__init__(dim)


class GaussianND(Target):
    """Zero-mean Gaussian with dense precision matrix A: logp = -1/2 x^T A x (BASELINE.json config 5)."""
    kind = L.GAUSSIAN_ND

    def __init__(self, precision):
        a = np.asarray(precision, dtype=np.float64)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("precision must be square")
        self.precision = a
        super().__init__(a.shape[0], matrix=a)

    @staticmethod
    def ill_conditioned(dim: int = 32, cond: float = 1e4, seed: int = 7) -> "GaussianND":
        """The synthetic config-5 target: A = Q diag(lambda) Q^T, lambda log-spaced 1..cond, Q from the QR of a
        seeded normal matrix."""
        rng = np.random.default_rng(seed)
        q, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
        lam = np.logspace(0.0, np.log10(cond), dim)
        a = (q * lam) @ q.T
        return GaussianND((a + a.T) / 2.0)


class UserTarget(Target):
    """A target of the user's own: the GPU analogue of `impl GradientTarget for MyDensity` (distributions.rs:65-108).

    `source` is HIP C++ defining `template <class T> struct mmcmc_user_target` with `static constexpr int dim`,
    `logp(P, x)` and `logp_grad(P, x, g)` (include/mmcmc.h: mmcmc_target_register_source); it is compiled at run time
    (`hipcc --genco` in a child process where hipcc is installed, else hipRTC) into the engine's MH / HMC kernels for f32 and f64 and its NUTS kernel for the three type modes.  `params` (up to 8 numbers) arrive as `P.p[i]`,
    `matrix` ([dim, dim]) as `P.mat`.  `UserTarget.compile_log` holds the compiler's diagnostics."""

    def __init__(self, name: str, dim: int, source: str, params=(), matrix=None):
        super().__init__(dim, params, matrix)
        kind = C.c_int(0)
        log = C.create_string_buffer(1 << 16)
        st = L.lib().mmcmc_target_register_source(name.encode(), int(dim), source.encode(), C.byref(kind), log, len(log))
        self.compile_log = log.value.decode(errors="replace")
        if st != L.OK:
            raise L.MmcmcError(st, "mmcmc_target_register_source" + (": " + self.compile_log[-2000:] if self.compile_log else ""))
        self.kind = kind.value
        self.name = name
        #: "hipcc" (`hipcc --genco` in a child process: the default wherever hipcc is installed) or "hiprtc"
        self.compiler = {1: "hipcc", 2: "hiprtc"}.get(L.lib().mmcmc_rtc_unit_compiler(self.kind), "?")


RTC_COMPILERS = {"auto": 0, "hipcc": 1, "hiprtc": 2}


def set_rtc_compiler(which: str = "auto") -> None:
    """mmcmc_rtc_set_compiler: which compiler builds run-time compiled units from now on (process-wide).  "auto": hipcc in a
    child process where it is installed, else hipRTC; "hipcc" / "hiprtc" pin one (for tests and A/B runs)."""
    L.check(L.lib().mmcmc_rtc_set_compiler(RTC_COMPILERS[which]), "mmcmc_rtc_set_compiler")


def rtc_compiler_info() -> dict:
    """mmcmc_rtc_compiler_info: {"hiprtc_path": str or None, "process_hip": (major, minor, patch), "built_with": (major, minor,
    patch)} -- which copy of hipRTC this process would compile units with (the one PyTorch bundles once torch is imported: its
    7.0.2 compiler miscompiled a NUTS kernel in rounds 3-4), the HIP runtime the process bound, and the compiler the library
    was built with.  Needs no GPU."""
    buf = C.create_string_buffer(512)
    pv, bv = C.c_int(0), C.c_int(0)
    st = L.lib().mmcmc_rtc_compiler_info(buf, 512, C.byref(pv), C.byref(bv))
    split = lambda v: (v // 10000000, v // 100000 % 100, v % 100000)
    return {"hiprtc_path": buf.value.decode() if st == L.OK else None, "process_hip": split(pv.value), "built_with": split(bv.value)}


class UserProposal:
    """A proposal of the user's own: the GPU analogue of `impl Proposal for MyProposal` (distributions.rs:92-101).

    `source` is HIP C++ defining `template <class T> struct mmcmc_user_proposal` with `sample(sigma, x, z, out)` (the
    proposed state from the current one and `dim` standard normals) and `logp(sigma, from, to)` = log q(to | from)
    (include/mmcmc.h: mmcmc_proposal_register_source).  It is compiled at run time for ONE target (a built-in one at its
    dimension, or a `UserTarget`); MetropolisHastings then keeps both q-terms of the acceptance ratio as
    metropolis_hastings.rs:303-315 does.  `std` is the proposal's one run-time parameter (`sigma`)."""

    def __init__(self, name: str, target: Target, source: str, std: float = 1.0):
        kind = C.c_int(0)
        log = C.create_string_buffer(1 << 16)
        st = L.lib().mmcmc_proposal_register_source(name.encode(), int(target.kind), int(target.dim), source.encode(),
                                                    C.byref(kind), log, len(log))
        self.compile_log = log.value.decode(errors="replace")
        if st != L.OK:
            raise L.MmcmcError(st, "mmcmc_proposal_register_source" + (": " + self.compile_log[-2000:] if self.compile_log else ""))
        self.kind, self.name, self.std, self.target = kind.value, name, float(std), target

    def set_seed(self, seed: int) -> "UserProposal":
        """Proposal::set_seed: no effect, as for IsotropicGaussian (the noise is the sampler's counter-based stream)."""
        self._seed = int(seed)
        return self

    def proposal_desc(self) -> L.ProposalDesc:
        p = L.ProposalDesc()
        p.kind = self.kind
        p.std = self.std
        return p


class Categorical:
    """distributions.rs:421-477 `Categorical::new(probs)` (`Discrete` + `Target<usize>`): host-side utility, not on the
    GPU path.  Probabilities are normalised on construction; `sample` walks the cumulative sums with `r <= cum` and
    falls back to the last index; `logp` is ln p_i, -inf out of range.  The generator is numpy's (the reference seeds
    a SmallRng from the OS)."""

    def __init__(self, probs, seed=None):
        p = np.asarray(probs, dtype=np.float64)
        self.probs = p / p.sum()
        self._rng = np.random.default_rng(seed)

    def sample(self) -> int:
        r = self._rng.random()
        cum = 0.0
        for i, p in enumerate(self.probs):
            cum += p
            if r <= cum:
                return i
        return len(self.probs) - 1

    def logp(self, index: int) -> float:
        return float(np.log(self.probs[index])) if 0 <= index < len(self.probs) else float("-inf")

    def unnorm_logp(self, position) -> float:
        return self.logp(int(position[0]))
