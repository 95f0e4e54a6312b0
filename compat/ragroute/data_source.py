"""ragroute.data_source, served by ragroute_amd (same names as reference ragroute/data_source.py)."""
from ragroute_amd.data_source import DataSource, run_data_source  # noqa: F401
