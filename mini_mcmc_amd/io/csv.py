"""io/csv.rs: `save_csv` (:47-69) and `save_csv_tensor` (:110-147) in the reference's exact text layout."""
from __future__ import annotations

import numpy as np

from .. import _lib as L


def save_csv(data, filename: str) -> None:
    """data: [chain, observation, dimension] (numpy, or a torch tensor which is copied to the host)."""
    if hasattr(data, "detach"):
        data = data.detach().cpu().numpy()
    a = np.ascontiguousarray(data)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    if a.ndim != 3:
        raise ValueError("data must be [chain, observation, dimension]")
    c, n, d = a.shape
    st = L.lib().mmcmc_save_csv(a.ctypes.data, L.F32 if a.dtype == np.float32 else L.F64, c, n, d,
                                str(filename).encode())
    L.check(st, "mmcmc_save_csv")


def save_csv_tensor(tensor, filename: str) -> None:
    """io/csv.rs:110-147 -- same layout for a [chain, observation, dimension] tensor."""
    save_csv(tensor, filename)
