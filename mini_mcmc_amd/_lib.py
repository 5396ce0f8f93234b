"""ctypes loader for libmmcmc.so (the C ABI declared in include/mmcmc.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C mini_mcmc_amd/csrc`.  There is NO fallback:
if the shared object is missing or a symbol cannot be bound, importing a sampler raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmcmc.so")

F32, F64 = 0, 1

# target kinds (include/mmcmc.h)
GAUSSIAN2D, DIFFABLE_GAUSSIAN2D, ISOTROPIC_GAUSSIAN, ROSENBROCK2D, ROSENBROCK_ND, STANDARD_NORMAL, GAUSSIAN_ND = range(7)

OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_SHAPE, ERR_NO_DEVICE, ERR_STATE, ERR_GROUP_BROKEN = -1, -2, -3, -4, -5, -6


class TargetDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("params", C.c_double * 8), ("matrix", C.POINTER(C.c_double))]


class ProposalDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("std", C.c_double)]


class BasicStats(C.Structure):
    _fields_ = [("min", C.c_float), ("median", C.c_float), ("max", C.c_float), ("mean", C.c_float), ("std", C.c_float)]


class RunStats(C.Structure):
    _fields_ = [("ess", BasicStats), ("rhat", BasicStats)]


# void (*mmcmc_progress_fn)(void *user, uint64_t done, uint64_t total, float p_accept, float max_rhat)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_float, C.c_float)


class Timing(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_float),
        ("n_launches", C.c_uint32),
        ("reserved", C.c_uint32),
        ("out_bytes", C.c_uint64),
        ("state_bytes", C.c_uint64),
    ]


class MmcmcError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = lib().mmcmc_status_string(status).decode() if _lib is not None else "?"
        super().__init__(f"{where}: status {status} ({msg})")


_lib = None
_vp = C.c_void_p
_TP = C.POINTER(TargetDesc)
_PP = C.POINTER(ProposalDesc)

# name -> (restype, argtypes); every symbol include/mmcmc.h declares
SIGNATURES = {
    "mmcmc_version": (C.c_int, []),
    "mmcmc_status_string": (C.c_char_p, [C.c_int]),
    "mmcmc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mmcmc_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "mmcmc_init_with_seed": (C.c_int, [C.c_size_t, C.c_size_t, C.c_uint64, C.POINTER(C.c_double)]),
    "mmcmc_mh_create": (C.c_int, [C.POINTER(_vp), _TP, _PP, _vp, C.c_size_t, C.c_int, C.c_int]),
    "mmcmc_mh_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int, C.POINTER(C.c_uint64), _vp]),
    "mmcmc_mh_state": (C.c_int, [_vp, _vp]),
    "mmcmc_mh_sync": (C.c_int, [_vp]),
    "mmcmc_mh_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "mmcmc_mh_destroy": (C.c_int, [_vp]),
    "mmcmc_mh_set_iters_per_launch": (C.c_int, [_vp, C.c_uint32]),
    "mmcmc_hmc_create": (C.c_int, [C.POINTER(_vp), _TP, _vp, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int]),
    "mmcmc_hmc_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_hmc_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_hmc_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int, C.POINTER(C.c_uint64), _vp]),
    "mmcmc_hmc_step": (C.c_int, [_vp, _vp]),
    "mmcmc_hmc_state": (C.c_int, [_vp, _vp]),
    "mmcmc_hmc_sync": (C.c_int, [_vp]),
    "mmcmc_hmc_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "mmcmc_hmc_destroy": (C.c_int, [_vp]),
    "mmcmc_hmc_set_iters_per_launch": (C.c_int, [_vp, C.c_uint32]),
    "mmcmc_mh_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "mmcmc_hmc_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "mmcmc_nuts_create": (C.c_int, [C.POINTER(_vp), _TP, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int, C.c_int]),
    "mmcmc_nuts_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_nuts_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_nuts_set_max_depth": (C.c_int, [_vp, C.c_int]),
    "mmcmc_nuts_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "mmcmc_nuts_kernel_variant": (C.c_int, [_vp]),
    "mmcmc_nuts_set_compaction": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mmcmc_nuts_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int, C.c_int, _vp]),
    "mmcmc_nuts_state": (C.c_int, [_vp, _vp]),
    "mmcmc_nuts_adapt_state": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mmcmc_nuts_leapfrog_counts": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "mmcmc_nuts_depth_histogram": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "mmcmc_nuts_sync": (C.c_int, [_vp]),
    "mmcmc_nuts_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "mmcmc_nuts_destroy": (C.c_int, [_vp]),
    "mmcmc_split_rhat_mean_ess": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, _vp]),
    "mmcmc_stats_set_kernel": (C.c_int, [C.c_int]),
    "mmcmc_stats_set_direct_work_limit": (C.c_int, [C.c_uint64]),
    "mmcmc_stats_partials": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp, C.c_int, _vp]),
    "mmcmc_stats_finish": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t,
                                     C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mmcmc_stats_finish_sums": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_size_t,
                                          C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mmcmc_basic_stats_from": (C.c_int, [C.POINTER(C.c_float), C.c_size_t, C.POINTER(BasicStats)]),
    "mmcmc_run_stats_from": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(RunStats),
                                  C.c_int, _vp]),
    "mmcmc_hmc_kernel_variant": (C.c_int, [_vp]),
    "mmcmc_mh_enable_timing": (C.c_int, [_vp, C.c_int]),
    "mmcmc_hmc_enable_timing": (C.c_int, [_vp, C.c_int]),
    "mmcmc_tracker_create": (C.c_int, [C.POINTER(_vp), C.c_size_t, C.c_size_t, C.c_int]),
    "mmcmc_tracker_steps": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, _vp]),
    "mmcmc_tracker_stats": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "mmcmc_tracker_init_last": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "mmcmc_tracker_chain_stats": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "mmcmc_tracker_n": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "mmcmc_rtc_set_compiler": (C.c_int, [C.c_int]),
    "mmcmc_rtc_unit_compiler": (C.c_int, [C.c_int]),
    "mmcmc_rtc_compiler_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_tracker_shape": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "mmcmc_tracker_within_var": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "mmcmc_ess_from_chainstats": (C.c_int, [_vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, _vp, C.POINTER(C.c_float),
                                            C.c_int, _vp]),
    "mmcmc_mh_run_rows": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp]),
    "mmcmc_hmc_run_rows": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp]),
    "mmcmc_mh_shape": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_hmc_shape": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_nuts_shape": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_mh_run_progress": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp, C.c_int, C.POINTER(RunStats),
                                        C.POINTER(_vp), _vp]),
    "mmcmc_hmc_run_progress": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp, C.c_int, C.POINTER(RunStats),
                                         C.POINTER(_vp), _vp]),
    "mmcmc_nuts_run_progress": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp, C.c_int, C.POINTER(RunStats),
                                          C.POINTER(_vp), _vp]),
    "mmcmc_tracker_destroy": (C.c_int, [_vp]),
    "mmcmc_mh_discrete_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_size_t, C.c_int]),
    "mmcmc_mh_discrete_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "mmcmc_gibbs_mixture_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
    "mmcmc_mh_discrete_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_discrete_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_discrete_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int, _vp]),
    "mmcmc_mh_discrete_state": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "mmcmc_mh_discrete_accept_counts": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "mmcmc_mh_discrete_sync": (C.c_int, [_vp]),
    "mmcmc_mh_discrete_destroy": (C.c_int, [_vp]),
    "mmcmc_gibbs_mixture_create": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_int]),
    "mmcmc_gibbs_mixture_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_gibbs_mixture_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_gibbs_mixture_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int, _vp]),
    "mmcmc_gibbs_mixture_state": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mmcmc_gibbs_mixture_sync": (C.c_int, [_vp]),
    "mmcmc_gibbs_mixture_destroy": (C.c_int, [_vp]),
    "mmcmc_save_csv": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_char_p]),
    "mmcmc_logp_grad_batch": (C.c_int, [_TP, C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_int]),
    "mmcmc_draw_noise": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "mmcmc_draw_noise_mh": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "mmcmc_target_register_source": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "mmcmc_discrete_register_source": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "mmcmc_proposal_register_source": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "mmcmc_hmc_group_create": (C.c_int, [C.POINTER(_vp), _TP, _vp, C.c_size_t, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "mmcmc_hmc_group_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_hmc_group_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_hmc_group_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.POINTER(C.c_uint64)]),
    "mmcmc_hmc_group_run_async": (C.c_int, [_vp, C.c_size_t, C.c_size_t]),
    "mmcmc_hmc_group_stats_phases": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mmcmc_hmc_group_state": (C.c_int, [_vp, _vp]),
    "mmcmc_hmc_group_split_rhat_mean_ess": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mmcmc_hmc_group_shard": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(_vp)]),
    "mmcmc_hmc_group_sync": (C.c_int, [_vp]),
    "mmcmc_hmc_group_stream_timer": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "mmcmc_hmc_group_exchange": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_hmc_group_destroy": (C.c_int, [_vp]),
    "mmcmc_mh_group_create": (C.c_int, [C.POINTER(_vp), _TP, _PP, _vp, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "mmcmc_mh_group_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_group_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_mh_group_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.POINTER(C.c_uint64)]),
    "mmcmc_mh_group_run_async": (C.c_int, [_vp, C.c_size_t, C.c_size_t]),
    "mmcmc_mh_group_stats_phases": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mmcmc_mh_group_state": (C.c_int, [_vp, _vp]),
    "mmcmc_mh_group_split_rhat_mean_ess": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mmcmc_mh_group_sync": (C.c_int, [_vp]),
    "mmcmc_mh_group_stream_timer": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "mmcmc_mh_group_exchange": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_mh_group_destroy": (C.c_int, [_vp]),
    "mmcmc_nuts_group_create": (C.c_int, [C.POINTER(_vp), _TP, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "mmcmc_nuts_group_seed": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_nuts_group_set_chain_offset": (C.c_int, [_vp, C.c_uint64]),
    "mmcmc_nuts_group_set_max_depth": (C.c_int, [_vp, C.c_int]),
    "mmcmc_nuts_group_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t, _vp, C.c_int]),
    "mmcmc_nuts_group_stats_phases": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "mmcmc_nuts_group_state": (C.c_int, [_vp, _vp]),
    "mmcmc_nuts_group_leapfrog_counts": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "mmcmc_nuts_group_split_rhat_mean_ess": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mmcmc_nuts_group_sync": (C.c_int, [_vp]),
    "mmcmc_nuts_group_stream_timer": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "mmcmc_nuts_group_exchange": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmcmc_nuts_group_destroy": (C.c_int, [_vp]),
}


def lib() -> C.CDLL:
    """Load libmmcmc.so and bind every entry point. Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mini_mcmc_amd/csrc`). mini_mcmc_amd has no CPU fallback."
        )
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so (soname libamdhip64.so.7).
    # Importing torch first makes the loader resolve libmmcmc.so's NEEDED entry to that already-loaded copy, so
    # torch tensors / streams and the engine share one runtime; without torch the system ROCm copy is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int, where: str) -> None:
    if status != OK:
        raise MmcmcError(status, where)
