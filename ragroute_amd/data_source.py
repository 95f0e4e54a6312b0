"""Data source of the retrieval hot path — host-side mirror of reference ragroute/data_source.py.

Same class and method names (`DataSource.retrieve_docs_medrag/fed4rag/wikipedia(query_embed, k)`,
`load_faiss_index`, `run_data_source`) and the same return tuples, but `index.search` is the HIP flat
scan over an HBM-resident corpus (flat_index.FlatIndex).  Row -> id -> text lookups stay on the host
exactly as data_source.py:143-215 does them.  Beyond the reference's single-query call shape, every
`retrieve_docs_*` accepts a [B,d] batch and then returns one tuple per query."""
import json
import logging
import os
import time

import numpy as np

from . import config
from .flat_index import FlatIndex, normalize_L2

logger = logging.getLogger("client")


def read_faiss_flat_index(path, mmap=True):
    """Minimal reader for a FAISS IndexFlat file ("IxFI"/"IxF2"): returns (float32 [ntotal,d], metric).
    With mmap=True (default) the payload is memory-mapped, so a 70 GB index (MedRAG pubmed: 23.9 M x 768 f32) is streamed
    into HBM chunk by chunk by FlatIndex.add instead of being loaded into host RAM first.
    Layout per FAISS's index_write.cpp (1.7.x): fourcc, d:i32, ntotal:i64, 2 x i64 dummy, is_trained:u8,
    metric_type:i32 [, metric_arg:f32 if metric_type > 1], then size:u64 (in floats) + payload.
    Format knowledge is external to the reference tree and unverified against a real faiss file here."""
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
            raise ValueError(f"{path}: not a flat FAISS index (fourcc {fourcc!r}); only IndexFlat files are supported")
        d = int(np.frombuffer(f.read(4), np.int32)[0])
        ntotal = int(np.frombuffer(f.read(8), np.int64)[0])
        f.read(16)
        f.read(1)
        metric_type = int(np.frombuffer(f.read(4), np.int32)[0])
        if metric_type > 1:
            f.read(4)
        nfloat = int(np.frombuffer(f.read(8), np.uint64)[0])
        if nfloat != ntotal * d:
            raise ValueError(f"{path}: payload of {nfloat} floats does not match ntotal*d = {ntotal * d}")
        offset = f.tell()
        if mmap and ntotal > 0:
            xb = np.memmap(path, dtype=np.float32, mode="r", offset=offset, shape=(ntotal, d))
        else:
            xb = np.fromfile(f, dtype=np.float32, count=nfloat).reshape(ntotal, d)
    return xb, ("ip" if metric_type == 0 else "l2")


def write_faiss_flat_index(path, xb, metric="ip"):
    """Writer of the same layout (test fixtures / synthetic corpora)."""
    xb = np.ascontiguousarray(xb, np.float32)
    with open(path, "wb") as f:
        f.write(b"IxFI" if metric == "ip" else b"IxF2")
        f.write(np.int32(xb.shape[1]).tobytes())
        f.write(np.int64(xb.shape[0]).tobytes())
        f.write(np.int64(1 << 20).tobytes() * 2)
        f.write(np.uint8(1).tobytes())
        f.write(np.int32(0 if metric == "ip" else 1).tobytes())
        f.write(np.uint64(xb.size).tobytes())
        xb.tofile(f)


class DataSource:
    def __init__(self, client_id: int, dataset: str, name: str, simulate: bool = False, dtype: str = "fp16"):
        self.client_id = client_id
        self.dataset = dataset
        self.simulate = simulate
        self.dtype = dtype
        if dataset == "medrag":
            self.dataset_dir = config.MEDRAG_DIR
        elif dataset == "feb4rag":
            self.dataset_dir = config.FEB4RAG_DIR
        elif dataset == "wikipedia":
            self.dataset_dir = config.WIKIPEDIA_DIR
        else:
            raise ValueError(f"Unknown dataset when starting data source {name}: {dataset}")
        self.name = name
        self.recv_port = config.SERVER_CLIENT_BASE_PORT + client_id
        self.send_port = config.CLIENT_SERVER_BASE_PORT + client_id
        self.running = False
        if dataset == "medrag":
            self.index_dir = os.path.join(self.dataset_dir, name, "index", "ncbi/MedCPT-Article-Encoder")
            self.index_path = os.path.join(self.index_dir, "faiss.index")
            self.doc_ids_path = os.path.join(self.index_dir, "metadatas.jsonl")
        elif dataset == "feb4rag":
            self.index_dir = os.path.join(self.dataset_dir, "dataset_creation", "2_search", "embeddings", name)
            model_name = config.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][name][0]
            self.index_path = os.path.join(self.index_dir, f"{name}_{model_name}.faiss")
            self.doc_ids_path = os.path.join(self.index_dir, f"{name}_{model_name}.docids.json")
        else:
            self.index_dir = os.path.join(self.dataset_dir, "faiss_clusters", "normalized_indexes")
            self.index_path = os.path.join(self.index_dir, f"faiss_index_{name}_normalized.index")
            self.doc_ids_path = None
        self.mmlu_titles, self.mmlu_texts = [], []
        self.faiss_indexes = None
        self.cache_jsonl = {}
        self._batcher = None
        self.batch_window_ms = float(os.environ.get("RAGROUTE_BATCH_WINDOW_MS", 2.0))

    # -- loading (data_source.py:69-80) -------------------------------------------------------------
    def load_faiss_index(self):
        logger.info(f"Loading FAISS index for {self.name}")
        xb, file_metric = read_faiss_flat_index(self.index_path)
        self.index_metric = file_metric  # the index FILE decides the metric in the reference (faiss.read_index, data_source.py:71)
        if self.dataset == "medrag":
            metadatas = [json.loads(line) for line in open(self.doc_ids_path).read().strip().split("\n")]
        elif self.dataset == "feb4rag":
            with open(self.doc_ids_path, "r") as f:
                metadatas = json.load(f)
        else:
            metadatas = []
            split = os.path.join(self.dataset_dir, "faiss_clusters", "split_texts_titles")
            with open(os.path.join(split, f"titles_{self.name}.txt"), "r", encoding="utf-8") as f:
                self.mmlu_titles = f.read().splitlines()
            with open(os.path.join(split, f"texts_{self.name}.txt"), "r", encoding="utf-8") as f:
                self.mmlu_texts = f.read().splitlines()
        self.set_index(xb, metadatas)

    def set_index(self, xb, metadatas, titles=None, texts=None):
        """Install a corpus directly: xb float32 [n,d] (numpy / memmap) or an existing FlatIndex."""
        if isinstance(xb, FlatIndex):
            index = xb
        else:
            metric = getattr(self, "index_metric", "ip")
            if metric == "l2" and xb.shape[1] > 768:
                raise NotImplementedError("L2 flat indexes wider than 768 are not supported yet")
            index = FlatIndex(xb.shape[1], metric=metric, dtype=self.dtype)
            index.reserve(xb.shape[0])
            index.add(xb)
        self.faiss_indexes = index, metadatas
        if titles is not None:
            self.mmlu_titles, self.mmlu_texts = list(titles), list(texts)

    # -- retrieval (data_source.py:143-215) -----------------------------------------------------------
    def _search(self, query_embed, k):
        q = np.asarray(query_embed, dtype=np.float32)
        single = q.ndim == 1 or q.shape[0] == 1
        q = q.reshape(1, -1) if q.ndim == 1 else q
        index, metadatas = self.faiss_indexes
        D, I = index.search(q, k)
        return D, I, metadatas, single

    def _medrag_idx2txt(self, indices):
        results = []
        for i in indices:
            source, index = i["source"], i["index"]
            if source not in self.cache_jsonl:
                with open(os.path.join(self.dataset_dir, self.name, "chunk", f"{source}.jsonl"), "r") as file:
                    self.cache_jsonl[source] = file.read().strip().split("\n")
            results.append(json.loads(self.cache_jsonl[source][index]))
        return results

    def retrieve_docs_medrag(self, query_embed, k):
        D, I, metadatas, single = self._search(query_embed, k)
        out = []
        for q in range(D.shape[0]):
            rows = [int(i) for i in I[q] if i >= 0]
            indices = [metadatas[i] for i in rows]
            out.append((indices, self._medrag_idx2txt(indices), D[q][: len(rows)].tolist()))
        return out[0] if single else out

    def retrieve_docs_fed4rag(self, query_embed, k):
        D, I, docids, single = self._search(query_embed, k)
        if self.name not in self.cache_jsonl:
            corpus = {}
            path = os.path.join(self.dataset_dir, "dataset_creation/original_dataset", self.name, self.name, "corpus.jsonl")
            with open(path, "r") as file:
                for line in file:
                    entry = json.loads(line)
                    corpus[entry["_id"]] = entry
            self.cache_jsonl[self.name] = corpus
        corpus_data = self.cache_jsonl[self.name]
        out = []
        for q in range(D.shape[0]):
            ids = [docids[int(i)] for i in I[q] if i >= 0]
            out.append((ids, [corpus_data.get(doc_id, None) for doc_id in ids], []))  # no scores for FeB4RAG (data_source.py:163)
        return out[0] if single else out

    def retrieve_docs_wikipedia(self, query_embed, k):
        query_vec = np.ascontiguousarray(np.asarray(query_embed, dtype=np.float32).reshape(-1, np.asarray(query_embed).shape[-1]))
        normalize_L2(query_vec)  # data_source.py:198-199
        D, I, _, single = self._search(query_vec, k)
        out = []
        for q in range(D.shape[0]):
            rows = [int(i) for i in I[q] if i >= 0]
            docs = [(self.mmlu_titles[i], self.mmlu_texts[i]) for i in rows]
            out.append((rows, docs, D[q][: len(rows)].tolist()))
        return out[0] if single else out

    # -- batched serving ---------------------------------------------------------------
    def retrieve_batch(self, embeddings, k=None):
        """One GPU search for a window of queries; returns one (ids, docs, scores) tuple per query."""
        k = config.K[self.dataset] if k is None else k
        fn = {"medrag": self.retrieve_docs_medrag, "feb4rag": self.retrieve_docs_fed4rag,
              "wikipedia": self.retrieve_docs_wikipedia}[self.dataset]
        batch = np.stack([np.asarray(e, dtype=np.float32).reshape(-1) for e in embeddings])
        out = fn(batch, k)
        return [out] if batch.shape[0] == 1 else out

    async def handle_query(self, query_data):
        """Reply message for one request, in the reference's wire format (data_source.py:123-131).  Concurrent
        requests are coalesced by the batcher into one scan of up to 256 queries."""
        import asyncio
        start_time = time.time()
        if self.simulate:
            ids, docs, scores = ["doc1", "doc2", "doc3"], ["Document 1 content", "Document 2 content", "Document 3 content"], [0.9, 0.85, 0.8]
            await asyncio.sleep(config.DATA_SOURCE_DELAY)
        else:
            if self._batcher is None:
                from .queue_manager import QueryBatcher
                self._batcher = QueryBatcher(self.retrieve_batch, max_batch=256, max_wait_ms=self.batch_window_ms)
            ids, docs, scores = await self._batcher.submit(query_data["embedding"])
        return {"query_id": query_data["id"], "client_id": self.client_id, "name": self.name, "indices": ids, "docs": docs,
                "scores": scores, "duration": time.time() - start_time}

    # -- service loop (transport glue; needs pyzmq like the reference) ----------------------------------
    async def start(self):  # pragma: no cover - needs pyzmq
        import asyncio
        import zmq
        import zmq.asyncio
        ctx = zmq.asyncio.Context()
        self.running = True
        receiver = ctx.socket(zmq.PULL)
        receiver.bind(f"tcp://*:{self.recv_port}")
        sender = ctx.socket(zmq.PUSH)
        sender.connect(f"tcp://localhost:{self.send_port}")
        if not self.simulate and self.faiss_indexes is None:
            self.load_faiss_index()

        async def reply(query_data):
            try:
                await sender.send_json(await self.handle_query(query_data))
            except Exception as e:  # logged and dropped, as data_source.py:137-138
                logger.error(f"Error when fetching documents from data source {self.name} (query data: {query_data}): {e}")
        try:
            while self.running:
                asyncio.ensure_future(reply(await receiver.recv_json()))
        finally:
            receiver.close()
            sender.close()
            ctx.term()

    def stop(self):
        self.running = False


async def run_data_source(client_id: int, dataset: str, name: str, simulate: bool = False):
    """Process entry with the reference's signature (data_source.py:224-226)."""
    data_source = DataSource(client_id, dataset, name, simulate=simulate)
    await data_source.start()
