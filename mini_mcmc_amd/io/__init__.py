"""Sample sinks -- host-side mirror of the reference's `io` module (src/io/): CSV only (io/csv.rs:47-147)."""
from .csv import save_csv, save_csv_tensor  # noqa: F401
