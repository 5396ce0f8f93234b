"""CSV sink against the reference's own expectations (io/csv.rs:160-268) -- CPU only."""
import numpy as np


def test_reference_csv_cases(tmp_path):
    from mini_mcmc_amd.io import save_csv

    f = tmp_path / "a.csv"
    save_csv(np.zeros((0, 0, 0), dtype=np.float32), f)          # io/csv.rs:160-178
    assert f.read_text().strip() == "chain,observation"
    save_csv(np.array([[[42.0]]]), f)                            # io/csv.rs:181-197
    assert f.read_text().strip() == "chain,observation,dim_0\n0,0,42"
    save_csv(np.array([[[1, 2], [3, 4]], [[10, 20], [30, 40]]]), f)   # io/csv.rs:200-219
    assert f.read_text().strip() == "chain,observation,dim_0,dim_1\n0,0,1,2\n0,1,3,4\n1,0,10,20\n1,1,30,40"
    save_csv(np.array([[[1.0, 2.0], [3.0, 4.0]], [[1.1, 2.1], [3.1, 4.1]]], dtype=np.float32), f)  # :222-268
    rows = [r.split(",") for r in f.read_text().strip().split("\n")]
    assert rows[0] == ["chain", "observation", "dim_0", "dim_1"]
    assert rows[1:] == [["0", "0", "1", "2"], ["0", "1", "3", "4"], ["1", "0", "1.1", "2.1"], ["1", "1", "3.1", "4.1"]]


def test_rust_display_float_formatting(tmp_path):
    from mini_mcmc_amd.io import save_csv

    f = tmp_path / "b.csv"
    vals = np.array([[[1e-7, 1e21, -0.0, 0.1, np.nan, np.inf, -np.inf, 123456789.125]]])
    save_csv(vals, f)
    got = f.read_text().strip().split("\n")[1].split(",")[2:]
    # Rust: format!("{}", v) -- shortest round-trip, never exponent notation
    assert got == ["0.0000001", "1000000000000000000000", "-0", "0.1", "NaN", "inf", "-inf", "123456789.125"]
    x = np.random.default_rng(0).standard_normal((3, 5, 2)).astype(np.float32)
    save_csv(x, f)
    back = np.array([[float(v) for v in r.split(",")[2:]] for r in f.read_text().strip().split("\n")[1:]], dtype=np.float32)
    assert np.array_equal(back.reshape(3, 5, 2), x)  # round-trips exactly
