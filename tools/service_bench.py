"""Service-level throughput of the data-source loop on one MI355X.

    python tools/service_bench.py [rows] [windows_ms ...]          (default: 10_000_000 rows, windows 0.2 0.5 2)

The reference's harness drives concurrency through HTTP (`run_benchmark.py --parallel N`, run_benchmark.py:90-112); every request
reaches a data source as ONE message `{"id", "query", "embedding": [768 floats]}` (http_server.py:205-209) and is answered with
`{"query_id", "client_id", "name", "indices", "docs", "scores", "duration"}` (data_source.py:123-131).  Here C closed-loop
clients (C = 1, 32, 256: each sends its next request when the previous reply is in) call `DataSource.handle_query` of the
MedRAG-shaped source in process — JSON-encoding every request and reply like the wire does (tests/stub_zmq.py) — against a
synthetic 10M x 768 fp16 corpus, for each batcher window `RAGROUTE_BATCH_WINDOW_MS`.  Reported: requests per second, p50 / p99
latency, mean queries per GPU search (how full the batcher's windows are), next to the raw `FlatIndex.search` (numpy in / out,
no service) at batch 1 / 32 / 256 — what the service could reach if it cost nothing.

Per request the service does what the reference's does (data_source.py:165-194): the search, `metadatas[i]` for the k rows,
the JSONL text lookup per row; metadata and texts are synthetic (columnar metadata as load_faiss_index builds it).
Round 4: the batcher runs the search and the reply building of consecutive windows on two threads (QueryBatcher search= / finish=)."""
import asyncio
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ragroute_amd import config
from ragroute_amd.data_source import DataSource
from ragroute_amd.flat_index import FlatIndex


def synthetic_meta(n):
    """metadatas.jsonl stand-in, columnar as DataSource.load_faiss_index builds it (MedragMetadata): row -> {"index": row % 4096,
    "source": f"chunk{row % 8}"} (data_source.py:73, 169-170)."""
    from ragroute_amd.data_source import MedragMetadata
    rows = np.arange(n, dtype=np.int64)
    return MedragMetadata(rows % 4096, (rows % 8).astype(np.int32), [f"chunk{c}" for c in range(8)])


def make_corpus(n, d, dev):
    idx = FlatIndex(d, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    xb = torch.zeros((n, idx.dim), dtype=torch.float16, device=dev)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        x = torch.randn((e - s, d), generator=g, device=dev)
        xb[s:e, :d] = (x / x.norm(dim=1, keepdim=True)).half()
    idx.adopt(xb)
    return idx


async def closed_loop(ds, queries, clients, seconds, wire="json"):
    """wire = "json": every request and reply is JSON-encoded and decoded in this (the event loop's) thread, as pyzmq's
    send_json / recv_json do in the reference; "objects": the same messages passed as Python objects (embedding = float32 array) -
    what the batcher and the data source sustain when the transport costs nothing."""
    lat, done = [], 0
    stop_at = time.perf_counter() + seconds

    async def client(c):
        nonlocal done
        i = c
        while time.perf_counter() < stop_at:
            msg = {"id": f"q{c}_{i}", "query": "synthetic", "embedding": queries[i % len(queries)]}
            t0 = time.perf_counter()
            if wire == "json":
                msg = json.loads(json.dumps(msg))               # http_server.py:205-209 -> data_source.py:102
            reply = await ds.handle_query(msg)
            if wire == "json":
                json.dumps(reply)                               # the reply crosses the wire as JSON (data_source.py:132)
            lat.append(time.perf_counter() - t0)
            done += 1
            i += clients

    t0 = time.perf_counter()
    await asyncio.gather(*[client(c) for c in range(clients)])
    dt = time.perf_counter() - t0
    lat.sort()
    b = ds._batcher
    return {"clients": clients, "wire": wire, "requests": done, "requests_per_s": round(done / dt, 1),
            # the SEARCH thread's share per window (decode, GPU search, copy-out) and the REPLY thread's (metadata + text lookups);
            # the two overlap (QueryBatcher's two-stage form), so the search thread's share is what gates the GPU
            "worker_ms_per_search": round(b.search_seconds / max(1, b.batches_run) * 1e3, 3),
            "reply_build_ms_per_window": round(b.finish_seconds / max(1, b.batches_run) * 1e3, 3),
            "p50_ms": round(lat[len(lat) // 2] * 1e3, 3), "p99_ms": round(lat[min(len(lat) - 1, int(len(lat) * 0.99))] * 1e3, 3),
            "queries_per_search": round(b.items_run / max(1, b.batches_run), 1), "searches": b.batches_run}


def raw_search(idx, queries, batch, seconds):
    q = np.asarray(queries[:batch], np.float32)
    idx.search(q, 32)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        idx.search(q, 32)
        n += 1
    dt = time.perf_counter() - t0
    return {"batch": batch, "queries_per_s": round(n * batch / dt, 1), "ms_per_search": round(dt / n * 1e3, 3)}


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    windows = [float(v) for v in sys.argv[2:]] or [0.2, 1.0]
    dev = torch.device("cuda:0")
    d, k = 768, config.K["medrag"]
    idx = make_corpus(rows, d, dev)
    rng = np.random.default_rng(4321)
    q = rng.standard_normal((1024, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    queries = [row.tolist() for row in q]                        # the wire carries Python float lists (router.py:317-319)
    arrays = [row.copy() for row in q]
    out = {"workload": f"{rows} x {d} fp16 rows, medrag-shaped data source (k = {k}), closed-loop clients calling DataSource.handle_query "
                       "in process, JSON-encoded requests and replies", "raw_FlatIndex_search": [raw_search(idx, q, b, 1.5) for b in (1, 32, 256)],
           "service": []}
    meta = synthetic_meta(rows)
    DataSource.tune_runtime()     # what DataSource.start() does in a service process (gc.freeze + young-generation threshold)
    for wire in ("json", "objects"):
        for w in windows:
            for clients in (1, 32, 256, 512):
                ds = DataSource(0, "medrag", "pubmed")
                ds.batch_window_ms = w
                ds.set_index(idx, meta)
                ds.cache_jsonl = {f"chunk{c}": [json.dumps({"id": f"c{c}_{i}", "title": f"title {i}", "content": "x" * 200}) for i in range(4096)]
                                  for c in range(8)}
                res = asyncio.run(closed_loop(ds, queries if wire == "json" else arrays, clients, 3.0, wire))
                res["batch_window_ms"] = w
                out["service"].append(res)
                print(json.dumps(res), file=sys.stderr, flush=True)
    for wire in ("json", "objects"):
        best = max((r for r in out["service"] if r["clients"] == 256 and r["wire"] == wire), key=lambda r: r["requests_per_s"])
        out[f"best_window_ms_at_256_clients_{wire}"] = best["batch_window_ms"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
