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


_FLAT_FOURCC = {b"IxFI": 0, b"IxF2": 1}   # IndexFlatIP -> METRIC_INNER_PRODUCT, IndexFlatL2 -> METRIC_L2


def read_faiss_flat_index(path, mmap=True):
    """Minimal reader for a FAISS IndexFlat file: returns (float32 [ntotal,d], "ip" | "l2") — what `faiss.read_index`
    (data_source.py:71) yields for the flat indexes the scan kernel replaces.
    With mmap=True (default) the payload is memory-mapped, so a 70 GB index (MedRAG pubmed: 23.9 M x 768 f32) is streamed
    into HBM chunk by chunk by FlatIndex.add instead of being loaded into host RAM first.

    Layout (FAISS 1.7.x impl/index_write.cpp, `write_index_header` + the IndexFlat branch; little endian):
        fourcc        4 B   "IxFI" (inner product) | "IxF2" (L2) | "IxFl" (IndexFlat with another metric)
        d             i32
        ntotal        i64
        dummy, dummy  2 x i64 (1 << 20)
        is_trained    u8
        metric_type   i32   0 = METRIC_INNER_PRODUCT, 1 = METRIC_L2, > 1 = other metrics, followed by
        [metric_arg   f32   only when metric_type > 1]
        size          u64   payload length in 4-byte units (= ntotal * d)
        payload       size x f32, row-major
    Anything else — a non-flat index (HNSW, IVF, PQ ...), a metric other than IP / L2, a header that contradicts its
    fourcc or its payload — raises ValueError, which the service loop logs (data_source.py:137-138).
    The layout is external knowledge (faiss is not in the reference tree nor in this image): byte-level fixtures in
    tests/test_faiss_file_format.py restate it independently of the writer below."""
    with open(path, "rb") as f:
        head = f.read(4 + 4 + 8 + 16 + 1 + 4)
        if len(head) < 4:
            raise ValueError(f"{path}: not a FAISS index file (shorter than a fourcc)")
        fourcc = head[:4]
        if fourcc not in _FLAT_FOURCC and fourcc != b"IxFl":
            raise ValueError(f"{path}: not a flat FAISS index (fourcc {fourcc!r}); only IndexFlatIP / IndexFlatL2 files are supported")
        if len(head) < 37:
            raise ValueError(f"{path}: truncated IndexFlat header")
        d = int(np.frombuffer(head, np.int32, 1, 4)[0])
        ntotal = int(np.frombuffer(head, np.int64, 1, 8)[0])
        metric_type = int(np.frombuffer(head, np.int32, 1, 33)[0])
        if metric_type not in (0, 1):
            raise ValueError(f"{path}: metric_type {metric_type} is not supported (only 0 = inner product and 1 = L2)")
        if fourcc in _FLAT_FOURCC and _FLAT_FOURCC[fourcc] != metric_type:
            raise ValueError(f"{path}: fourcc {fourcc!r} contradicts metric_type {metric_type}")
        if d <= 0 or ntotal < 0:
            raise ValueError(f"{path}: bad header (d={d}, ntotal={ntotal})")
        size = f.read(8)
        if len(size) < 8:
            raise ValueError(f"{path}: truncated IndexFlat header")
        nfloat = int(np.frombuffer(size, np.uint64)[0])
        if nfloat != ntotal * d:
            raise ValueError(f"{path}: payload of {nfloat} floats does not match ntotal*d = {ntotal * d}")
        offset = f.tell()
        f.seek(0, 2)
        if f.tell() - offset < nfloat * 4:
            raise ValueError(f"{path}: file holds {f.tell() - offset} payload bytes, header promises {nfloat * 4}")
        f.seek(offset)
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
        self._parsed_docs = {}
        self._batcher = None
        self.batch_window_ms = float(os.environ.get("RAGROUTE_BATCH_WINDOW_MS", 0.2))

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
        """data_source.py:166-183: the JSONL line `index` of chunk/{source}.jsonl, parsed.  A batched search looks up nq * k lines
        per window, so each line is parsed once and kept (the reference re-parses it on every hit); callers get their own copy."""
        results = []
        for i in indices:
            source, index = i["source"], i["index"]
            if source not in self.cache_jsonl:
                with open(os.path.join(self.dataset_dir, self.name, "chunk", f"{source}.jsonl"), "r") as file:
                    self.cache_jsonl[source] = file.read().strip().split("\n")
            parsed = self._parsed_docs.setdefault(source, {})
            doc = parsed.get(index)
            if doc is None:
                doc = parsed[index] = json.loads(self.cache_jsonl[source][index])
            results.append(dict(doc) if isinstance(doc, dict) else doc)
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
    async def start(self):
        """data_source.py:82-141: bind PULL :6000+i, connect PUSH :7500+i, load the index, then serve.  Every request
        becomes its own task so that requests arriving together meet in the batcher (one scan for the window)."""
        import asyncio
        import zmq
        import zmq.asyncio
        self.context = zmq.asyncio.Context()
        self.running = True
        self.receiver = self.context.socket(zmq.PULL)
        self.receiver.bind(f"tcp://*:{self.recv_port}")
        self.sender = self.context.socket(zmq.PUSH)
        self.sender.connect(f"tcp://localhost:{self.send_port}")
        self._serve_task = asyncio.current_task()
        pending = set()
        try:
            if not self.simulate and self.faiss_indexes is None:
                self.load_faiss_index()

            async def reply(query_data):
                try:
                    await self.sender.send_json(await self.handle_query(query_data))
                except asyncio.CancelledError:
                    raise
                except Exception as e:  # logged, reply dropped — as data_source.py:137-138
                    logger.error(f"Error when fetching documents from data source {self.name} (query data: {query_data}): {e}")

            while self.running:
                query_data = await self.receiver.recv_json()
                task = asyncio.ensure_future(reply(query_data))
                pending.add(task)
                task.add_done_callback(pending.discard)
        except asyncio.CancelledError:
            logger.info(f"Client {self.client_id} shutdown requested")
        finally:
            for task in list(pending):
                task.cancel()
            self.stop()

    def stop(self):
        """Stop serving and close the sockets (data_source.py:217-222)."""
        self.running = False
        task, self._serve_task = getattr(self, "_serve_task", None), None
        for name in ("receiver", "sender"):
            sock = getattr(self, name, None)
            if sock is not None:
                sock.close()
                setattr(self, name, None)
        ctx = getattr(self, "context", None)
        if ctx is not None:
            ctx.term()
            self.context = None
        if task is not None and not task.done():
            import asyncio
            try:
                current = asyncio.current_task()
            except RuntimeError:
                current = None
            if task is not current:
                task.cancel()


CURRENT = {}   # client_id -> the DataSource this process serves with (set by run_data_source; introspection / tests)


async def run_data_source(client_id: int, dataset: str, name: str, simulate: bool = False):
    """Process entry with the reference's signature (data_source.py:224-226)."""
    CURRENT[client_id] = data_source = DataSource(client_id, dataset, name, simulate=simulate)
    await data_source.start()
