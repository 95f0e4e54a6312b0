"""GPU: BASELINE config 1 plumbing (MIRAGE/medrag, --routing all|ragroute, --disable-llm) replayed in process.

The reference's front-end (ragroute/http_server.py:168-293) does, per query: router reply -> forward the embedding to every
selected data source -> gather (indices, docs, scores) -> concatenate in client order (280-286) -> rerank_medrag (289).
Transport (ZeroMQ/aiohttp) is out of scope; this test wires the same steps with the drop-in classes and checks the final
documents against the oracle chain (oracle flat search per source + oracle rerank)."""
import json

import numpy as np
import pytest

from tests.util import int_data, synth_router_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("routing", ["all", "ragroute", "none"])
def test_medrag_query_flow(gpu, routing):
    from oracle import oracle as O
    from ragroute_amd import config as C
    from ragroute_amd.data_source import DataSource
    from ragroute_amd.rerank import rerank_medrag
    from ragroute_amd.router import Router
    rng = np.random.default_rng(2)
    sources = C.DATA_SOURCES["medrag"]
    K = C.K["medrag"]
    corpora, data_sources = {}, {}
    for cid, name in enumerate(sources):
        n = [3000, 12_000, 700, 30_000][cid]
        xb = int_data(rng, n, 768)
        metadatas = [{"index": i, "source": f"{name}_file"} for i in range(n)]
        ds = DataSource(cid, "medrag", name)
        ds.set_index(xb, metadatas)
        ds.cache_jsonl = {f"{name}_file": [json.dumps({"id": f"{name}-{i}", "title": f"t{i}", "content": f"{name} chunk {i}"}) for i in range(n)]}
        corpora[name], data_sources[name] = xb, ds
    case = synth_router_case("medrag", 11)
    router = Router("medrag", sources, routing)
    mean, scale = case["scaler"]
    router.set_router(case["sd"], case["centroids"], mean, scale)
    model = router.model_names[0]
    for qi in range(6):
        emb = int_data(rng, 1, 768)[0]
        selected = router.select_relevant_sources({model: emb})                         # router.py:221-239
        if routing == "all":
            assert selected == sources
        if routing == "none":
            assert selected == []
        all_docs, all_scores, want_docs, want_scores = [], [], [], []
        for name in sources:                                                             # http_server.py:198-209, 280-286
            if name not in selected:
                continue
            ids, docs, scores = data_sources[name].retrieve_docs_medrag(np.asarray(emb, np.float32).reshape(1, -1), K)
            all_docs.extend(docs)
            all_scores.extend(scores)
            Dr, Ir = O.flat_search_ip(corpora[name], emb[None, :], K)
            want_docs.extend({"id": f"{name}-{i}", "title": f"t{i}", "content": f"{name} chunk {i}"} for i in Ir[0])
            want_scores.extend(float(s) for s in Dr[0])
        got_docs, got_scores = rerank_medrag(all_docs, all_scores, K)                    # http_server.py:289
        exp_docs, exp_scores = O.rerank_medrag(want_docs, want_scores, K)
        assert got_scores == exp_scores
        assert got_docs == exp_docs
        assert len(got_docs) == (min(K, len(all_docs)))
