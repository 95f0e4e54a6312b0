"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

    PYTHONPATH=/root/repo python tests/golden/make_golden.py

Imports (never copies) /root/reference/ragroute/{rerank,router,data_source}.py.  Only transport and
encoder imports are stubbed (zmq, sentence_transformers: no arithmetic in them).  For data_source.py the
`faiss` import has to be stubbed too — faiss is not installed — so that fixture pins only the GLUE
(row -> metadata -> text lookups and the returned tuple layout, data_source.py:143-215); the stub's search
is the CPU oracle on integer-valued data (exact scores).  The reference never travels to the GPU box:
only the JSON files written here do.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tests.util import flat_golden_inputs, synth_medrag_corpus, synth_router_case  # noqa: E402


def _stub_transport():
    zmq = types.ModuleType("zmq")
    zmq.asyncio = types.ModuleType("zmq.asyncio")
    zmq.PULL, zmq.PUSH = 0, 1

    class _Ctx:
        def socket(self, *a):
            return None
    zmq.asyncio.Context = _Ctx
    sys.modules["zmq"] = zmq
    sys.modules["zmq.asyncio"] = zmq.asyncio
    st = types.ModuleType("sentence_transformers")
    stm = types.ModuleType("sentence_transformers.models")
    st.SentenceTransformer = type("SentenceTransformer", (), {})
    stm.Transformer = stm.Pooling = type("_T", (), {})
    st.models = stm
    sys.modules["sentence_transformers"] = st
    sys.modules["sentence_transformers.models"] = stm


def golden_rerank():
    from ragroute import rerank as R
    rng = np.random.default_rng(2024)
    cases = []
    for n, k in [(128, 32), (100, 10), (40, 10), (7, 10), (256, 32), (1, 1), (0, 5)]:
        scores = [float(x) for x in rng.permutation(n * 4)[:n] / 7.0 - 20.0]  # tie-free
        docs = [f"doc{i}" for i in range(n)]
        md, ms = R.rerank_medrag(docs, scores, k)
        wd, ws = R.rerank_wikipedia(docs, scores, k)
        cases.append({"docs": docs, "scores": scores, "k": k, "medrag": [md, [float(s) for s in ms]],
                      "wikipedia": [wd, [float(s) for s in ws]]})
    # scores that differ only BEYOND float32 (the reference sorts the float64 list as given, rerank.py:5, 30): tie-free in f64,
    # pairwise equal once rounded to f32
    f64_cases = []
    for n, k in [(64, 16), (10, 10), (33, 5)]:
        base = [float(np.float32(x)) for x in rng.permutation(n * 4)[: n // 2 + 1] / 7.0 - 3.0]
        scores = []
        for i in range(n):
            scores.append(base[i // 2] * (1.0 + (1e-12 if i % 2 else 0.0)) + (3e-13 if i % 2 else 0.0))
        order = rng.permutation(n)
        scores = [scores[i] for i in order]
        assert len(set(scores)) == n and len({float(np.float32(s)) for s in scores}) < n
        docs = [f"doc{i}" for i in range(n)]
        md, ms = R.rerank_medrag(docs, scores, k)
        wd, ws = R.rerank_wikipedia(docs, scores, k)
        f64_cases.append({"docs": docs, "scores": scores, "k": k, "medrag": [md, [float(s) for s in ms]], "wikipedia": [wd, [float(s) for s in ws]]})
    # ties: only the score multiset is defined by the reference (numpy's argsort order is not)
    tie_scores = [0.1, 0.9, 0.5, 0.9, 0.3, 0.2, 0.7, 0.5]
    md, ms = R.rerank_medrag(list("abcdefgh"), tie_scores, 4)
    ties = {"docs": list("abcdefgh"), "scores": tie_scores, "k": 4, "medrag_scores": [float(s) for s in ms]}
    rel = {"q1": [["d3", "2"], ["d1", "1"], ["d9", "3"], ["d7", "0"]], "q2": []}
    feb = []
    for ids, qid, k in [(["d1", "d2", "d3", "d4", "d7"], "q1", 3), (["d4", "d2"], "q1", 5), (["d1", "d2"], "q2", 2),
                        (["d1"], "zzz", 1), ([], "q1", 3)]:
        docs = [f"text of {i}" for i in ids]
        od, oi = R.rerank_feb4rag(ids, docs, qid, k, {q: [tuple(x) for x in v] for q, v in rel.items()})
        feb.append({"ids": ids, "docs": docs, "query_id": qid, "k": k, "out_docs": od, "out_ids": oi})
    json.dump({"cases": cases, "f64_cases": f64_cases, "ties": ties, "feb4rag": {"relevance": rel, "cases": feb}},
              open(os.path.join(HERE, "rerank.json"), "w"), indent=1)


def golden_router():
    import torch
    from sklearn.preprocessing import StandardScaler
    from ragroute import router as RR
    out = {}
    for dataset, seed in [("medrag", 11), ("feb4rag", 12), ("wikipedia", 13)]:
        case = synth_router_case(dataset, seed)
        r = RR.Router.__new__(RR.Router)
        r.dataset, r.data_sources, r.routing_strategy, r.simulate, r.device = dataset, case["sources"], "ragroute", False, "cpu"
        net = RR.CorpusRoutingNN(case["sd"]["fc1.weight"].shape[1])
        net.load_state_dict({k: torch.from_numpy(v) for k, v in case["sd"].items()})
        net.eval()
        r.router = net
        d_max = case["d_max"]
        r.centroids = {c: np.pad(v, (0, d_max - len(v))) for c, v in case["centroids"].items()}  # router.py:149-151
        if case["scaler"] is not None:
            sc = StandardScaler()
            sc.mean_, sc.scale_ = case["scaler"]
            sc.var_ = sc.scale_ ** 2
            sc.n_features_in_ = len(sc.mean_)
            r.scaler = sc
        # capture the logits the reference computes inside select_relevant_sources_ragroute
        logits, selected = [], []
        orig = net.forward

        def spy(x, _orig=orig, _log=logits):
            y = _orig(x)
            _log.append(y.detach().view(-1).numpy().copy())
            return y
        net.forward = spy
        for q in case["queries"]:
            selected.append(r.select_relevant_sources(q))
        out[dataset] = {"seed": seed, "logits": [[float(v) for v in l] for l in logits], "selected": selected}
    json.dump(out, open(os.path.join(HERE, "router.json"), "w"), indent=1)


def golden_data_source_glue():
    from oracle import oracle as O
    faiss = types.ModuleType("faiss")

    class _FakeIndex:  # stand-in for the absent faiss wheel: the arithmetic here is the oracle, NOT the reference
        def __init__(self, xb):
            self.xb = xb

        def search(self, xq, k):
            return O.flat_search_ip(self.xb, np.asarray(xq, np.float32), k)
    store = {}
    faiss.read_index = lambda path: _FakeIndex(store[path])
    faiss.normalize_L2 = O.normalize_L2
    sys.modules["faiss"] = faiss
    from ragroute import config as RC
    from ragroute import data_source as DS
    xb, metadatas, chunks = synth_medrag_corpus(5)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        DS.MEDRAG_DIR = RC.MEDRAG_DIR = tmp
        ds = DS.DataSource(0, "medrag", "textbooks")
        os.makedirs(ds.index_dir)
        os.makedirs(os.path.join(tmp, "textbooks", "chunk"))
        with open(ds.doc_ids_path, "w") as f:
            f.write("\n".join(json.dumps(m) for m in metadatas))
        for b, lines in chunks.items():
            with open(os.path.join(tmp, "textbooks", "chunk", f"{b}.jsonl"), "w") as f:
                f.write("\n".join(json.dumps(l) for l in lines))
        store[ds.index_path] = xb
        ds.load_faiss_index()
        rng = np.random.default_rng(99)
        queries = rng.integers(-2, 3, size=(3, 768)).astype(np.float32)
        res = []
        for q in queries:
            indices, docs, scores = ds.retrieve_docs_medrag(q.reshape(1, -1), 32)
            res.append({"indices": indices, "docs": docs, "scores": [float(s) for s in scores]})
        out["medrag"] = {"corpus_seed": 5, "query_seed": 99, "k": 32, "results": res}
        # feb4rag: ids are docids strings, docs come from corpus.jsonl keyed by _id, NO scores (data_source.py:143-163)
        DS.FEB4RAG_DIR = RC.FEB4RAG_DIR = tmp
        fs = DS.DataSource(1, "feb4rag", "scifact")
        os.makedirs(fs.index_dir)
        docids = [f"doc-{i * 7 % 1000}-{i}" for i in range(xb.shape[0])]
        json.dump(docids, open(fs.doc_ids_path, "w"))
        cdir = os.path.join(tmp, "dataset_creation/original_dataset", "scifact", "scifact")
        os.makedirs(cdir)
        with open(os.path.join(cdir, "corpus.jsonl"), "w") as f:
            for i, did in enumerate(docids):
                if i % 5 != 0:  # some docids are missing from the corpus file -> None entries
                    f.write(json.dumps({"_id": did, "title": f"t{i}", "text": f"body {i}"}) + "\n")
        store[fs.index_path] = xb
        fs.load_faiss_index()
        res = []
        for q in queries:
            ids, docs, scores = fs.retrieve_docs_fed4rag(q.reshape(1, -1), 10)
            res.append({"ids": ids, "docs": docs, "scores": scores})
        out["feb4rag"] = {"k": 10, "results": res}
        # wikipedia: normalize_L2 on the query, ids are the rows, docs are (title, text)
        DS.WIKIPEDIA_DIR = tmp
        split = os.path.join(tmp, "faiss_clusters", "split_texts_titles")
        os.makedirs(split)
        n = xb.shape[0]
        open(os.path.join(split, "titles_3.txt"), "w").write("\n".join(f"title {i}" for i in range(n)))
        open(os.path.join(split, "texts_3.txt"), "w").write("\n".join(f"text {i}" for i in range(n)))
        ws = DS.DataSource(3, "wikipedia", "3")
        xbn = xb.copy()
        xbn[:, 0] += 3  # avoid zero rows
        O.normalize_L2(xbn)
        import torch
        xbn = torch.from_numpy(xbn).to(torch.float16).to(torch.float32).numpy()  # what an fp16 index stores
        store[ws.index_path] = xbn
        ws.load_faiss_index()
        res = []
        for q in queries:
            ids, docs, scores = ws.retrieve_docs_wikipedia(q.copy().reshape(1, -1), 10)
            res.append({"ids": [int(i) for i in ids], "docs": [list(d) for d in docs], "scores": [float(s) for s in scores]})
        out["wikipedia"] = {"k": 10, "results": res}
    json.dump(out, open(os.path.join(HERE, "data_source_glue.json"), "w"), indent=1)


def golden_flat_search():
    """index.search itself is third-party faiss (absent here), so there is no reference output to record: this fixture is an
    INDEPENDENT statement of the flat-IP semantics (f64 numpy matmul, (score desc, id asc) order) that both the C oracle and
    the HIP path are checked against.  It does not pin faiss's f32 summation order ("parity unpinned")."""
    k, out = 32, {}
    for case in ("gauss", "int", "cosine"):
        xb, xq = flat_golden_inputs(case)
        S = xq.astype(np.float64) @ xb.astype(np.float64).T
        ids = np.tile(np.arange(S.shape[1]), (S.shape[0], 1))
        order = np.lexsort((ids, -S), axis=1)[:, :k]
        out[case + "_I"] = order.astype(np.int64)
        out[case + "_D"] = np.take_along_axis(S, order, axis=1)
        out[case + "_gap"] = np.partition(S, -k - 1, axis=1)[:, -k - 1]   # (k+1)-th best score: how well separated the set is
    np.savez(os.path.join(HERE, "flat_search.npz"), **out)


if __name__ == "__main__":
    golden_flat_search()
    _stub_transport()
    golden_rerank()
    golden_router()
    golden_data_source_glue()
    print("golden vectors written to", HERE)
