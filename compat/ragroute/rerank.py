"""ragroute.rerank, served by ragroute_amd (same names as reference ragroute/rerank.py)."""
from ragroute_amd.rerank import rerank_feb4rag, rerank_medrag, rerank_wikipedia  # noqa: F401
