"""Arrow IPC / Parquet sinks -- host-side mirror of src/io/arrow.rs:53-119 and src/io/parquet.rs:49-130 (and their
`_tensor` variants), on pyarrow.  Schema as in the reference: `chain` UInt32, `observation` UInt32, `dim_0 .. dim_{D-1}`
Float64, all non-nullable, one record batch, chain-major rows (the tensor variants take [observation, chain, dim] and
write `observation` first, parquet.rs:154-230 / arrow.rs:166-240)."""
from __future__ import annotations

import numpy as np


def _table(sample, tensor_layout: bool):
    import pyarrow as pa

    a = np.asarray(sample)
    if a.ndim != 3:
        raise ValueError("sample must be 3-D")
    n0, n1, d = a.shape
    first = np.repeat(np.arange(n0, dtype=np.uint32), n1)
    second = np.tile(np.arange(n1, dtype=np.uint32), n0)
    names = ["observation", "chain"] if tensor_layout else ["chain", "observation"]
    flat = a.reshape(n0 * n1, d).astype(np.float64)
    fields = [pa.field(names[0], pa.uint32(), nullable=False), pa.field(names[1], pa.uint32(), nullable=False)]
    fields += [pa.field(f"dim_{k}", pa.float64(), nullable=False) for k in range(d)]
    arrays = [pa.array(first, type=pa.uint32()), pa.array(second, type=pa.uint32())]
    arrays += [pa.array(np.ascontiguousarray(flat[:, k]), type=pa.float64()) for k in range(d)]
    return pa.Table.from_arrays(arrays, schema=pa.schema(fields))


def save_arrow(sample, filename: str) -> None:
    """io/arrow.rs:53-119: sample [chains, observations, dims] -> Arrow IPC file (one record batch, also when empty)."""
    import pyarrow as pa

    t = _table(sample, False)
    with pa.OSFile(filename, "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        batches = t.to_batches()
        w.write_batch(batches[0] if batches else pa.RecordBatch.from_arrays([pa.array([], type=fl.type) for fl in t.schema], schema=t.schema))


def save_arrow_tensor(tensor, filename: str) -> None:
    """io/arrow.rs save_arrow_tensor: tensor [observations, chains, dims] (torch or numpy)."""
    import pyarrow as pa

    a = tensor.detach().cpu().numpy() if hasattr(tensor, "detach") else np.asarray(tensor)
    t = _table(a, True)
    with pa.OSFile(filename, "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        for b in t.to_batches():
            w.write_batch(b)


def save_parquet(sample, filename: str) -> None:
    """io/parquet.rs:49-130: sample [chains, observations, dims] -> Parquet."""
    import pyarrow.parquet as pq

    pq.write_table(_table(sample, False), filename)


def save_parquet_tensor(tensor, filename: str) -> None:
    """io/parquet.rs:154-230: tensor [observations, chains, dims] (torch or numpy)."""
    import pyarrow.parquet as pq

    a = tensor.detach().cpu().numpy() if hasattr(tensor, "detach") else np.asarray(tensor)
    pq.write_table(_table(a, True), filename)
