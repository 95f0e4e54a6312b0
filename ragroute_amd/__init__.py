"""ragroute_amd — MI355X-native retrieval hot path for RAGRoute (route -> per-source top-k -> merge).

Public surface (mirrors reference ragroute/{data_source,router,rerank}.py):
    flat_index.FlatIndex, flat_index.normalize_L2      faiss-shaped index / faiss.normalize_L2
    data_source.DataSource, data_source.run_data_source
    router.Router, router.CorpusRoutingNN, router.run_router
    rerank.rerank_medrag / rerank_feb4rag / rerank_wikipedia / merge_topk
    pipeline.RetrievalPipeline, sharded.*              device-side federation (one process per GPU)
    queue_manager.QueryQueue / QueryBatcher
"""
from ._lib import RagrouteHipError  # noqa: F401

__version__ = "0.1.0"
