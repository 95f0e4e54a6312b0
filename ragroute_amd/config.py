"""Constants of the hot path, mirroring reference ragroute/config.py (values, not code):
data-source order (config.py:32-36), query-encoder map (:37-71), one-hot ids (:72-90),
padded embedding length (:92-96), k per dataset (:97-101), router input widths (router.py:32-34),
routing thresholds (router.py:277-280).  Paths and ports are overridable by environment."""
import os

SERVER_ROUTER_PORT = int(os.environ.get("RAGROUTE_SERVER_ROUTER_PORT", 5555))
ROUTER_SERVER_PORT = int(os.environ.get("RAGROUTE_ROUTER_SERVER_PORT", 5556))
SERVER_CLIENT_BASE_PORT = int(os.environ.get("RAGROUTE_SERVER_CLIENT_BASE_PORT", 6000))
CLIENT_SERVER_BASE_PORT = int(os.environ.get("RAGROUTE_CLIENT_SERVER_BASE_PORT", 7500))
MAX_QUEUE_SIZE = 100

USR_DIR = os.environ.get("RAGROUTE_DATA_DIR", "/mnt/nfs/home/dpetresc")
MODELS_USR_DIR = os.environ.get("RAGROUTE_MODELS_DIR", USR_DIR)
MEDRAG_DIR = os.path.join(USR_DIR, "MedRAG", "corpus")
FEB4RAG_DIR = os.path.join(USR_DIR, "FeB4RAG")
WIKIPEDIA_DIR = os.path.join(USR_DIR, "wiki_dataset", "dpr_wiki_index")

ROUTER_DELAY = 1
DATA_SOURCE_DELAY = 2

DATA_SOURCES = {
    "medrag": ["pubmed", "statpearls", "textbooks", "wikipedia"],
    "feb4rag": ["msmarco", "trec-covid", "nfcorpus", "scidocs", "nq", "hotpotqa", "fiqa", "arguana",
                "webis-touche2020", "dbpedia-entity", "fever", "climate-fever", "scifact"],
    "wikipedia": [str(i) for i in range(10)],
}
_MEDCPT = "ncbi/MedCPT-Query-Encoder"
_DPR = "facebook/dpr-question_encoder-single-nq-base"
EMBEDDING_MODELS_PER_DATA_SOURCE = {
    "medrag": {s: (_MEDCPT, None) for s in DATA_SOURCES["medrag"]},
    "feb4rag": {
        "msmarco": ("e5-large", "custom"),
        "trec-covid": ("SGPT-5.8B-weightedmean-msmarco-specb-bitfit", "custom"),
        "nfcorpus": ("UAE-Large-V1", "custom"),
        "scidocs": ("all-mpnet-base-v2", "beir"),
        "nq": ("multilingual-e5-large", "custom"),
        "hotpotqa": ("ember-v1", "beir"),
        "fiqa": ("all-mpnet-base-v2", "beir"),
        "arguana": ("UAE-Large-V1", "custom"),
        "webis-touche2020": ("e5-base", "custom"),
        "dbpedia-entity": ("UAE-Large-V1", "custom"),
        "fever": ("UAE-Large-V1", "custom"),
        "climate-fever": ("UAE-Large-V1", "custom"),
        "scifact": ("gte-base", "beir"),
    },
    "wikipedia": {s: (_DPR, None) for s in DATA_SOURCES["wikipedia"]},
}
FEB4RAG_SOURCE_TO_ID = {s: i for i, s in enumerate(sorted(DATA_SOURCES["feb4rag"]))}
MEDRAG_SOURCE_TO_ID = {"pubmed": 0, "statpearls": 1, "textbooks": 2, "wikipedia": 3}
EMBEDDING_MAX_LENGTH = {"medrag": 768, "feb4rag": 4096, "wikipedia": 768}
K = {"medrag": 32, "feb4rag": 10, "wikipedia": 10}
ROUTER_INPUT_DIMENSION = {"medrag": 1540, "feb4rag": 8205, "wikipedia": 1546}
ROUTER_THRESHOLD = {"medrag": 0.4924, "feb4rag": 0.5, "wikipedia": 0.5}
RANDOM_ROUTING_SAMPLE = {"medrag": 2, "feb4rag": 9, "wikipedia": 2}
