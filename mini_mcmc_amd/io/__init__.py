"""Sample sinks -- host-side mirror of the reference's `io` module (src/io/): CSV (io/csv.rs:47-147, through the C ABI's
mmcmc_save_csv) and Arrow IPC / Parquet (io/arrow.rs, io/parquet.rs, on pyarrow)."""
from .arrow import save_arrow, save_arrow_tensor, save_parquet, save_parquet_tensor  # noqa: F401
from .csv import save_csv, save_csv_tensor  # noqa: F401
