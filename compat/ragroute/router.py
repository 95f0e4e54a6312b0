"""ragroute.router, served by ragroute_amd (same names as reference ragroute/router.py)."""
from ragroute_amd.router import CorpusRoutingNN, Router, run_router  # noqa: F401
