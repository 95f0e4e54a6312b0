"""The process entries `run_router` / `run_data_source` (reference ragroute/router.py:343-346, data_source.py:224-226, spawned
by ragroute/ragroute.py:10-16, 43-54) driven with the exact messages the front-end sends (http_server.py:153-156 to the router,
205-209 to a data source) over an in-test stand-in for pyzmq (tests/stub_zmq.py), and their replies' fields (router.py:324-330,
data_source.py:123-131) and contents checked against the ORACLE chain — not against the object's own single-query method."""
import asyncio
import json
import os
import pickle

import numpy as np
import pytest

from tests import stub_zmq
from tests.util import int_data, synth_router_case


@pytest.fixture
def hub():
    h = stub_zmq.install()
    yield h
    stub_zmq.uninstall()


def test_simulate_mode_wire_format_and_shutdown(hub, monkeypatch):
    """--simulate (main.py:17): no models, no indexes, no GPU — canned documents after DATA_SOURCE_DELAY, every source
    selected, random embeddings, ROUTER_DELAY before the router's reply (router.py:321-322); stop() closes both sockets and
    terminates the context (router.py:335-341, data_source.py:217-222)."""
    from ragroute_amd import config as C
    from ragroute_amd import data_source as DS
    from ragroute_amd import router as R
    monkeypatch.setattr(C, "ROUTER_DELAY", 0.2)
    monkeypatch.setattr(C, "DATA_SOURCE_DELAY", 0.1)
    sources = C.DATA_SOURCES["medrag"]

    async def go():
        rt = asyncio.ensure_future(R.run_router("medrag", sources, "ragroute", simulate=True))
        dt = asyncio.ensure_future(DS.run_data_source(2, "medrag", "textbooks", simulate=True))
        await asyncio.sleep(0.05)
        loop = asyncio.get_running_loop()
        t0 = loop.time()
        await hub.send(C.SERVER_ROUTER_PORT, {"id": "q-1", "query": "what is aspirin?"})          # http_server.py:153-156
        rep = await hub.recv(C.ROUTER_SERVER_PORT)
        assert loop.time() - t0 >= 0.2                                                             # the simulate delay
        assert set(rep) == {"query_id", "data_sources", "embeddings", "embedding_time", "selection_time"}
        assert rep["query_id"] == "q-1" and rep["data_sources"] == sources
        model = C.EMBEDDING_MODELS_PER_DATA_SOURCE["medrag"]["textbooks"][0]
        assert list(rep["embeddings"]) == [model] and len(rep["embeddings"][model]) == 768
        await hub.send(C.SERVER_CLIENT_BASE_PORT + 2, {"id": "q-1", "query": "what is aspirin?",   # http_server.py:205-209
                                                       "embedding": rep["embeddings"][model]})
        drep = await hub.recv(C.CLIENT_SERVER_BASE_PORT + 2)
        assert set(drep) == {"query_id", "client_id", "name", "indices", "docs", "scores", "duration"}
        assert (drep["query_id"], drep["client_id"], drep["name"]) == ("q-1", 2, "textbooks")
        assert drep["indices"] == ["doc1", "doc2", "doc3"] and drep["scores"] == [0.9, 0.85, 0.8] and drep["duration"] >= 0.1
        # shutdown: what ragroute.py:92-131 does to its children ends the loops and closes the transport
        rt.cancel()
        dt.cancel()
        await asyncio.gather(rt, dt, return_exceptions=True)
    asyncio.run(go())
    assert len(hub.sockets) == 4 and all(s.closed for s in hub.sockets)
    assert len(hub.contexts) == 2 and all(c.terminated for c in hub.contexts)


def test_stop_wakes_a_parked_service_loop(hub):
    """stop() from outside the loop task (the reference calls it from its signal handler path) must end start()."""
    from ragroute_amd import config as C
    from ragroute_amd.router import Router

    async def go():
        r = Router("medrag", C.DATA_SOURCES["medrag"], "all", simulate=True)
        t = asyncio.ensure_future(r.start())
        await asyncio.sleep(0.05)
        assert r.running and not t.done()
        r.stop()
        await asyncio.wait_for(t, 5.0)
        assert not r.running
    asyncio.run(go())
    assert all(s.closed for s in hub.sockets) and all(c.terminated for c in hub.contexts)


def _write_medrag_deployment(tmp_path, case, corpora):
    """The files a medrag deployment holds where config.py:16-24 points: router weights / scaler / stats, and per source a
    flat index, the row -> {index, source} map and the chunk files."""
    import torch
    from sklearn.preprocessing import StandardScaler
    from ragroute_amd.data_source import write_faiss_flat_index
    routing = tmp_path / "MedRAG" / "routing"
    os.makedirs(routing)
    torch.save({k: torch.from_numpy(v) for k, v in case["sd"].items()}, routing / "best_model.pth")
    sc = StandardScaler()
    sc.mean_, sc.scale_ = case["scaler"]
    sc.var_, sc.n_features_in_ = sc.scale_ ** 2, len(sc.mean_)
    pickle.dump(("X_train", "X_test", "y", sc, "extra"), open(routing / "preprocessed_data.pkl", "wb"))
    for name, cen in case["centroids"].items():
        json.dump({"centroid": cen.tolist()}, open(routing / f"{name}_stats.json", "w"))
    for name, xb in corpora.items():
        idx_dir = tmp_path / "MedRAG" / "corpus" / name / "index" / "ncbi/MedCPT-Article-Encoder"
        os.makedirs(idx_dir)
        os.makedirs(tmp_path / "MedRAG" / "corpus" / name / "chunk")
        write_faiss_flat_index(str(idx_dir / "faiss.index"), xb)
        half = len(xb) // 2
        metas = [{"index": i % half, "source": f"{name}_{'a' if i < half else 'b'}"} for i in range(len(xb))]
        open(idx_dir / "metadatas.jsonl", "w").write("\n".join(json.dumps(m) for m in metas))
        for part in "ab":
            lines = [json.dumps({"id": f"{name}_{part}_{j}", "title": f"t{j}", "content": f"{name} {part} chunk {j}"}) for j in range(len(xb))]
            open(tmp_path / "MedRAG" / "corpus" / name / "chunk" / f"{name}_{part}.jsonl", "w").write("\n".join(lines))


@pytest.mark.gpu
def test_run_router_and_run_data_source_replies_match_the_oracle_chain(gpu, hub, tmp_path, monkeypatch):
    from oracle import oracle as O
    from ragroute_amd import config as C
    from ragroute_amd import data_source as DS
    from ragroute_amd import router as R
    sources = C.DATA_SOURCES["medrag"]
    model = C.EMBEDDING_MODELS_PER_DATA_SOURCE["medrag"]["pubmed"][0]
    case = synth_router_case("medrag", 77)
    rng = np.random.default_rng(9)
    served = {2: "textbooks", 0: "pubmed"}                          # client_id -> name, as ragroute.py:48-54 numbers them
    corpora = {name: int_data(rng, n, 768) for name, n in (("textbooks", 5000), ("pubmed", 12_000))}
    _write_medrag_deployment(tmp_path, case, corpora)
    monkeypatch.setattr(C, "USR_DIR", str(tmp_path))
    monkeypatch.setattr(C, "MODELS_USR_DIR", str(tmp_path))
    monkeypatch.setattr(C, "MEDRAG_DIR", str(tmp_path / "MedRAG" / "corpus"))
    nq = 48
    texts = [f"question {i}" for i in range(nq)]
    embs = {t: int_data(rng, 1, 768)[0] for t in texts}             # the query encoder is out of scope: a lookup table
    R.set_encoder_factory(lambda dataset: (lambda query: {model: embs[query]}))
    try:
        async def go():
            tasks = [asyncio.ensure_future(R.run_router("medrag", sources, "ragroute"))]
            tasks += [asyncio.ensure_future(DS.run_data_source(cid, "medrag", name)) for cid, name in served.items()]
            await asyncio.sleep(0.05)
            # all queries at once, as run_benchmark.py --parallel does: the services must batch them and still answer each
            for i, t in enumerate(texts):
                await hub.send(C.SERVER_ROUTER_PORT, {"id": f"q{i}", "query": t})
            rreps = {}
            for _ in texts:
                rep = await hub.recv(C.ROUTER_SERVER_PORT, 120.0)
                rreps[rep["query_id"]] = rep
            for i, t in enumerate(texts):
                for cid in served:
                    await hub.send(C.SERVER_CLIENT_BASE_PORT + cid, {"id": f"q{i}", "query": t, "embedding": rreps[f"q{i}"]["embeddings"][model]})
            dreps = {cid: {} for cid in served}
            for cid in served:
                for _ in texts:
                    rep = await hub.recv(C.CLIENT_SERVER_BASE_PORT + cid, 120.0)
                    dreps[cid][rep["query_id"]] = rep
            for t in tasks:
                t.cancel()
            await asyncio.gather(*tasks, return_exceptions=True)
            # the 48 concurrent requests were served in a few windows, not one launch each (router.py:207-219 is serial)
            assert R.CURRENT._batcher.items_run == nq and R.CURRENT._batcher.batches_run <= 8
            for cid in served:
                assert DS.CURRENT[cid]._batcher.items_run == nq and DS.CURRENT[cid]._batcher.batches_run <= 8
            return rreps, dreps
        rreps, dreps = asyncio.run(go())
    finally:
        R.set_encoder_factory(None)
    mean, scale = case["scaler"]
    flips = 0
    for i, t in enumerate(texts):
        rep = rreps[f"q{i}"]
        assert set(rep) == {"query_id", "data_sources", "embeddings", "embedding_time", "selection_time"}
        assert rep["embeddings"] == {model: embs[t].tolist()}
        assert rep["embedding_time"] >= 0 and rep["selection_time"] >= 0
        logits = O.router_logits("medrag", sources, {s: model for s in sources}, C.MEDRAG_SOURCE_TO_ID, 768, {model: embs[t]},
                                 {s: np.pad(case["centroids"][s], (0, 768 - len(case["centroids"][s]))) for s in sources},
                                 case["sd"], mean, scale)
        want = O.router_select("medrag", sources, logits)
        if rep["data_sources"] != want:   # only a logit within 2e-4 of the boundary may flip (folded f32 sums reassociate)
            boundary = np.log(0.4924 / (1 - 0.4924))
            diff = set(rep["data_sources"]) ^ set(want)
            assert all(abs(logits[sources.index(s)] - boundary) < 2e-4 for s in diff)
            flips += 1
        for cid, name in served.items():
            d = dreps[cid][f"q{i}"]
            assert set(d) == {"query_id", "client_id", "name", "indices", "docs", "scores", "duration"}
            assert (d["client_id"], d["name"]) == (cid, name) and d["duration"] >= 0
            Dr, Ir = O.flat_search_ip(corpora[name], embs[t][None, :], C.K["medrag"])
            half = len(corpora[name]) // 2
            want_idx = [{"index": int(r) % half, "source": f"{name}_{'a' if r < half else 'b'}"} for r in Ir[0]]
            assert d["indices"] == want_idx                                   # data_source.py:190
            assert d["scores"] == [float(s) for s in Dr[0]]                   # data_source.py:187 (exact: integer data)
            assert d["docs"] == [{"id": f"{m['source']}_{m['index']}", "title": f"t{m['index']}",
                                  "content": f"{name} {m['source'][-1]} chunk {m['index']}"} for m in want_idx]   # 166-183
    assert flips <= 2
    assert all(s.closed for s in hub.sockets) and all(c.terminated for c in hub.contexts)
