"""Arrow IPC / Parquet sinks (io/arrow.rs, io/parquet.rs): the reference's own test cases (arrow.rs:131-300,
parquet.rs tests) restated -- schema chain / observation UInt32 + dim_k Float64, one batch, chain-major rows."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")


def _read_arrow(path):
    with pa.OSFile(str(path), "rb") as f:
        r = pa.ipc.open_file(f)
        return [r.get_batch(i) for i in range(r.num_record_batches)]


def test_save_arrow_reference_cases(tmp_path):
    from mini_mcmc_amd.io import save_arrow, save_arrow_tensor

    # test_save_arrow_empty_data (arrow.rs:131-163): no chains -> a file with one empty batch of 2 columns
    p = tmp_path / "empty.arrow"
    save_arrow(np.zeros((0, 0, 0), dtype=np.float32), str(p))
    b = _read_arrow(p)
    assert p.stat().st_size > 0 and len(b) == 1 and b[0].num_rows == 0 and b[0].num_columns == 2
    # test_save_arrow_single_value (arrow.rs:165-208)
    p = tmp_path / "one.arrow"
    save_arrow(np.array([[[42.0]]]), str(p))
    (b,) = _read_arrow(p)
    assert b.num_rows == 1 and b.num_columns == 3
    assert b.schema.names == ["chain", "observation", "dim_0"]
    assert b.schema.field("chain").type == pa.uint32() and b.schema.field("dim_0").type == pa.float64()
    assert not b.schema.field("chain").nullable
    assert b.column(0)[0].as_py() == 0 and b.column(1)[0].as_py() == 0 and b.column(2)[0].as_py() == 42.0
    # test_save_arrow_multi_chain (arrow.rs:217-286): 2 chains x 2 observations x 2 dims, f32 widened to f64
    data = np.array([[[1.0, 2.5], [3.0, 4.5]], [[10.0, 20.5], [30.0, 40.5]]], dtype=np.float32)
    p = tmp_path / "multi.arrow"
    save_arrow(data, str(p))
    (b,) = _read_arrow(p)
    assert b.num_rows == 4 and b.num_columns == 4
    assert b.column(0).to_pylist() == [0, 0, 1, 1] and b.column(1).to_pylist() == [0, 1, 0, 1]
    assert b.column(2).to_pylist() == [1.0, 3.0, 10.0, 30.0] and b.column(3).to_pylist() == [2.5, 4.5, 20.5, 40.5]
    # tensor variant: [observations, chains, dims], `observation` column first
    p = tmp_path / "tensor.arrow"
    save_arrow_tensor(data, str(p))
    (b,) = _read_arrow(p)
    assert b.schema.names[:2] == ["observation", "chain"] and b.column(0).to_pylist() == [0, 0, 1, 1]


def test_save_parquet_reference_cases(tmp_path):
    import pyarrow.parquet as pq

    from mini_mcmc_amd.io import save_parquet, save_parquet_tensor

    data = np.array([[[1.0, 2.5], [3.0, 4.5]], [[10.0, 20.5], [30.0, 40.5]]])
    p = tmp_path / "multi.parquet"
    save_parquet(data, str(p))
    t = pq.read_table(str(p))
    assert t.column_names == ["chain", "observation", "dim_0", "dim_1"] and t.num_rows == 4
    assert t.schema.field("chain").type == pa.uint32() and t.schema.field("dim_1").type == pa.float64()
    assert t.column("chain").to_pylist() == [0, 0, 1, 1] and t.column("dim_1").to_pylist() == [2.5, 4.5, 20.5, 40.5]
    p = tmp_path / "tensor.parquet"
    save_parquet_tensor(data, str(p))
    t = pq.read_table(str(p))
    assert t.column_names[:2] == ["observation", "chain"] and t.column("observation").to_pylist() == [0, 0, 1, 1]
    p = tmp_path / "empty.parquet"
    save_parquet(np.zeros((0, 0, 0)), str(p))
    assert pq.read_table(str(p)).num_rows == 0
