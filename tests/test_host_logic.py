"""CPU: host-side logic that needs no device — fc1 folding algebra, shard ids, candidate layout,
the flat-index file layout, config constants."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests.util import synth_router_case


@pytest.mark.parametrize("dataset", ["medrag", "feb4rag", "wikipedia"])
def test_fold_weights_reproduces_unfolded_router(dataset):
    """fc1(scaler(features)) == q @ w1q + c1[c] (float64 fold), so folded logits match the golden logits."""
    from ragroute_amd import config as C
    from ragroute_amd.router import fold_weights
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "router.json")))[dataset]
    case = synth_router_case(dataset, g["seed"])
    d_max, sources, sd = case["d_max"], case["sources"], case["sd"]
    cen = np.stack([np.pad(case["centroids"][c], (0, d_max - len(case["centroids"][c]))) for c in sources])
    if dataset == "medrag":
        ids, n1 = [C.MEDRAG_SOURCE_TO_ID[c] for c in sources], 4
    elif dataset == "feb4rag":
        ids, n1 = [C.FEB4RAG_SOURCE_TO_ID[c] for c in sources], 13
    else:
        ids, n1 = [int(c) for c in sources], 10
    mean, scale = case["scaler"] if case["scaler"] is not None else (None, None)
    w1q, c1 = fold_weights(sd, cen, ids, n1, d_max, mean, scale)
    assert w1q.shape == (d_max, 256) and c1.shape == (len(sources), 256)
    for q, want in zip(case["queries"], g["logits"]):
        rows = []
        for ci, c in enumerate(sources):
            e = q[C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][c][0]]
            rows.append(np.pad(e, (0, d_max - len(e))).astype(np.float64) @ w1q + c1[ci])
        pre = np.stack(rows).astype(np.float32)
        h = np.maximum(O.layer_norm(pre, sd["ln1.weight"], sd["ln1.bias"]), 0).astype(np.float32)
        h = np.maximum(O.layer_norm(h @ sd["fc2.weight"].T + sd["fc2.bias"], sd["ln2.weight"], sd["ln2.bias"]), 0).astype(np.float32)
        logits = (h @ sd["fc3.weight"].T + sd["fc3.bias"]).reshape(-1)
        assert np.allclose(logits, want, atol=5e-5, rtol=0)


def test_global_ids_and_mask():
    import torch
    from ragroute_amd import sharded as S
    gid = S.global_id(5, 123456789)
    assert S.split_global_id(gid) == (5, 123456789) and gid < 2 ** 63
    D = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    I = torch.arange(6, dtype=torch.int64).reshape(2, 3)
    Dm, Im = S.apply_route_mask(D, I, torch.tensor([True, False]))
    assert torch.equal(Dm[0], D[0]) and torch.isinf(Dm[1]).all() and (Im[1] == -1).all()
    Dg, Ig = S.gather_candidates(D, I)  # no process group: identity
    assert Dg is D and Ig is I


def test_flat_index_file_round_trip(tmp_path):
    from ragroute_amd.data_source import read_faiss_flat_index, write_faiss_flat_index
    xb = np.random.default_rng(0).standard_normal((37, 24)).astype(np.float32)
    p = str(tmp_path / "x.index")
    write_faiss_flat_index(p, xb, "ip")
    got, metric = read_faiss_flat_index(p)
    assert metric == "ip" and np.array_equal(got, xb) and isinstance(got, np.memmap)
    got2, _ = read_faiss_flat_index(p, mmap=False)
    assert np.array_equal(got2, xb) and not isinstance(got2, np.memmap)
    write_faiss_flat_index(p, xb, "l2")
    assert read_faiss_flat_index(p)[1] == "l2"
    open(p, "r+b").write(b"IwFl")
    with pytest.raises(ValueError):
        read_faiss_flat_index(p)


def test_data_source_paths_follow_reference(monkeypatch):
    from ragroute_amd import config as C
    from ragroute_amd.data_source import DataSource
    ds = DataSource(2, "medrag", "textbooks")
    assert ds.recv_port == C.SERVER_CLIENT_BASE_PORT + 2 and ds.send_port == C.CLIENT_SERVER_BASE_PORT + 2
    assert ds.index_path.endswith("textbooks/index/ncbi/MedCPT-Article-Encoder/faiss.index")
    fs = DataSource(0, "feb4rag", "scifact")
    assert fs.index_path.endswith("embeddings/scifact/scifact_gte-base.faiss")
    assert fs.doc_ids_path.endswith("scifact_gte-base.docids.json")
    with pytest.raises(ValueError):
        DataSource(0, "nope", "x")


def test_data_source_processes_spread_over_the_visible_gpus(monkeypatch):
    """One process per source (ragroute.py:10-16): client_id mod device_count, RAGROUTE_DEVICE overrides, one GPU = current device."""
    import torch
    from ragroute_amd.data_source import DataSource
    monkeypatch.delenv("RAGROUTE_DEVICE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert [str(DataSource(c, "feb4rag", "msmarco").pick_device()) for c in (0, 5, 8, 12)] == ["cuda:0", "cuda:5", "cuda:0", "cuda:4"]
    monkeypatch.setenv("RAGROUTE_DEVICE", "3")
    assert str(DataSource(0, "medrag", "pubmed").pick_device()) == "cuda:3"
    monkeypatch.setenv("RAGROUTE_DEVICE", "cuda:6")
    assert str(DataSource(1, "medrag", "pubmed").pick_device()) == "cuda:6"
    monkeypatch.delenv("RAGROUTE_DEVICE")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert DataSource(2, "medrag", "pubmed").pick_device() is None


def test_router_strategies_without_weights():
    from ragroute_amd import config as C
    from ragroute_amd.router import Router
    src = C.DATA_SOURCES["medrag"]
    assert Router("medrag", src, "all").select_relevant_sources({}) == src
    assert Router("medrag", src, "none").select_relevant_sources({}) == []
    assert len(Router("medrag", src, "random").select_relevant_sources({})) == 2
    assert len(Router("feb4rag", C.DATA_SOURCES["feb4rag"], "random").select_relevant_sources({})) == 9
    assert Router("medrag", src, "ragroute", simulate=True).select_relevant_sources({}) == src
    with pytest.raises(ValueError):
        Router("medrag", src, "bogus").select_relevant_sources({})
    emb = Router("feb4rag", C.DATA_SOURCES["feb4rag"], "all", simulate=True).encode_query("q")
    assert len(emb) == 8 and all(v.shape == (4096,) for v in emb.values())
    r = Router("feb4rag", C.DATA_SOURCES["feb4rag"], "ragroute")
    x = r.pack_queries({m: np.ones(8, np.float32) for m in r.model_names})
    assert tuple(x.shape) == (1, 8, 4096) and float(x.sum()) == 64.0


def test_rerank_feb4rag_host_path_matches_golden():
    from ragroute_amd.rerank import rerank_feb4rag
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rerank.json")))["feb4rag"]
    rel = {q: [tuple(x) for x in v] for q, v in g["relevance"].items()}
    for c in g["cases"]:
        d, i = rerank_feb4rag(c["ids"], c["docs"], c["query_id"], c["k"], rel)
        assert d == c["out_docs"] and i == c["out_ids"]


def test_load_router_reads_the_reference_file_formats(tmp_path):
    """router.py:106-151: state_dict via torch.load, medrag scaler = 4th element of a 5-tuple pickle, wikipedia scaler = bare
    pickle, centroid = stats JSON ["centroid"] zero padded to the dataset's max length (wikipedia: list indexed by cluster)."""
    import pickle
    import torch
    from sklearn.preprocessing import StandardScaler
    from ragroute_amd import config as C
    from ragroute_amd.router import Router
    case = synth_router_case("medrag", 31)
    model = tmp_path / "best_model.pth"
    torch.save({k: torch.from_numpy(v) for k, v in case["sd"].items()}, model)
    sc = StandardScaler()
    sc.mean_, sc.scale_ = case["scaler"]
    sc.var_, sc.n_features_in_ = sc.scale_ ** 2, len(sc.mean_)
    pk = tmp_path / "preprocessed_data.pkl"
    pickle.dump(("X_train", "X_test", "y", sc, "extra"), open(pk, "wb"))
    stats = {}
    for c, v in case["centroids"].items():
        f = tmp_path / f"{c}_stats.json"
        json.dump({"centroid": v[:700].tolist(), "n": 5}, open(f, "w"))   # shorter than 768 -> must be zero padded
        stats[c] = str(f)
    r = Router("medrag", case["sources"], "ragroute")
    r.load_router(model_path=str(model), scaler_path=str(pk), stats_files=stats)
    assert np.array_equal(r.router.state_dict()["fc1.weight"], case["sd"]["fc1.weight"])
    assert np.array_equal(r.scaler.mean_, case["scaler"][0])
    for c in case["sources"]:
        assert r.centroids[c].shape == (768,) and (r.centroids[c][700:] == 0).all()
        assert np.array_equal(r.centroids[c][:700], case["centroids"][c][:700])
    # wikipedia: bare scaler pickle and one stats file holding a list indexed by cluster id
    wcase = synth_router_case("wikipedia", 32)
    torch.save({k: torch.from_numpy(v) for k, v in wcase["sd"].items()}, model)
    wsc = StandardScaler()
    wsc.mean_, wsc.scale_ = wcase["scaler"]
    pickle.dump(wsc, open(pk, "wb"))
    f = tmp_path / "cluster_stats.json"
    json.dump([{"centroid": wcase["centroids"][str(i)].tolist()} for i in range(10)], open(f, "w"))
    w = Router("wikipedia", wcase["sources"], "ragroute")
    w.load_router(model_path=str(model), scaler_path=str(pk), stats_files={c: str(f) for c in wcase["sources"]})
    assert np.array_equal(w.centroids["7"], wcase["centroids"]["7"]) and np.array_equal(w.scaler.scale_, wcase["scaler"][1])
    with pytest.raises(KeyError):
        bad = dict(wcase["sd"])
        del bad["ln2.bias"]
        w.router.load_state_dict(bad)


def test_compat_shim_resolves_hot_path_modules_to_ragroute_amd():
    """compat/ragroute: router/data_source/rerank come from ragroute_amd; other modules fall through to the reference checkout
    (only checked where one is mounted: it never travels to the GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = "/root/reference"
    code = ("import ragroute.rerank as r, ragroute.router as ro, ragroute.data_source as ds\n"
            "assert r.rerank_medrag.__module__ == 'ragroute_amd.rerank'\n"
            "assert ro.Router.__module__ == 'ragroute_amd.router' and ds.DataSource.__module__ == 'ragroute_amd.data_source'\n"
            "import os\n"
            "if os.environ.get('RAGROUTE_REFERENCE_DIR'):\n"
            "    import ragroute.config as c, ragroute.queue_manager as q\n"
            "    assert c.__file__.startswith(os.environ['RAGROUTE_REFERENCE_DIR']) and c.K['medrag'] == 32\n"
            "    assert q.QueryQueue.__module__ == 'ragroute.queue_manager'\n"
            "print('shim ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "compat") + os.pathsep + root)
    if os.path.isdir(os.path.join(ref, "ragroute")):
        env["RAGROUTE_REFERENCE_DIR"] = ref
    else:
        env.pop("RAGROUTE_REFERENCE_DIR", None)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "shim ok" in res.stdout, res.stderr[-1500:]


# ---- list-shaped merges run on the HOST, like the reference's (rerank.py:3-9, 28-34; called from http_server.py:288-293) ----
def test_rerank_functions_match_reference_golden_on_the_host():
    from ragroute_amd import rerank as R
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rerank.json")))
    for c in g["cases"]:
        assert list(map(list, R.rerank_medrag(c["docs"], c["scores"], c["k"]))) == c["medrag"]
        assert list(map(list, R.rerank_wikipedia(c["docs"], c["scores"], c["k"]))) == c["wikipedia"]
    for c in g["f64_cases"]:     # scores that differ only beyond float32: the reference sorts the float64 list as given
        assert list(map(list, R.rerank_medrag(c["docs"], c["scores"], c["k"]))) == c["medrag"]
        assert list(map(list, R.rerank_wikipedia(c["docs"], c["scores"], c["k"]))) == c["wikipedia"]
    t = g["ties"]
    docs, scores = R.rerank_medrag(t["docs"], t["scores"], t["k"])
    assert scores == t["medrag_scores"] and docs[:2] == ["b", "d"]
    assert R.rerank_medrag([], [], 5) == ([], []) and R.rerank_wikipedia([], [], 5) == ([], [])


def test_rerank_nan_and_long_lists_follow_numpy():
    """rerank.py:5 is `np.argsort(scores)[::-1][:k]`, rerank.py:30 `np.argsort(scores)[:k]`: numpy sorts NaN to the END of the
    ascending order, so rerank_medrag ranks a NaN-scored document FIRST and rerank_wikipedia keeps it last.  Lists of any length."""
    from ragroute_amd.rerank import rerank_medrag, rerank_wikipedia
    rng = np.random.default_rng(12)
    scores = rng.permutation(40).astype(np.float64).tolist()       # tie-free
    scores[17] = float("nan")
    docs = [f"d{i}" for i in range(40)]
    for k in (1, 5, 39, 40, 64):
        order = np.argsort(scores)[::-1][:k]
        d, s = rerank_medrag(docs, scores, k)
        assert d == [docs[i] for i in order]
        assert [x for x in s if x == x] == [scores[i] for i in order if scores[i] == scores[i]] and (s[0] != s[0])
        order = np.argsort(scores)[:k]
        d, s = rerank_wikipedia(docs, scores, k)
        assert d == [docs[i] for i in order]
    n = 20_000
    scores = rng.permutation(n).astype(np.float64).tolist()
    docs = list(range(n))
    d, s = rerank_medrag(docs, scores, 32)
    order = np.argsort(scores)[::-1][:32]
    assert d == order.tolist() and s == [scores[i] for i in order]
    d, s = rerank_wikipedia(docs, scores, 100)
    assert d == np.argsort(scores)[:100].tolist()


def test_host_rerank_uses_the_device_merges_total_order():
    """Same candidates, same order as `rr_merge_topk` defines it (restated by the oracle's merge): f32 compare, the earlier
    candidate wins a tie, -0.0 == +0.0; the scores come back as the caller's own objects."""
    from oracle import oracle as O
    from ragroute_amd.rerank import rerank_medrag, rerank_wikipedia
    rng = np.random.default_rng(0)
    for trial in range(200):
        n, k = int(rng.integers(1, 300)), int(rng.integers(1, 140))
        sc = (np.round(rng.standard_normal(n) * 3) / 3).tolist()    # many ties, and both zeros
        docs = list(range(n))
        for desc, f in ((True, rerank_medrag), (False, rerank_wikipedia)):
            Dr, Ir = O.merge_topk(np.asarray(sc, np.float32)[None], np.arange(n, dtype=np.int64)[None], min(k, n), desc)
            d, s = f(docs, sc, k)
            assert d == Ir[0].tolist() and s == [sc[i] for i in d]


def test_list_rerank_needs_no_gpu():
    """The front-end process (http_server.py:25) imports these and must not need a HIP context for a 128-element sort."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "torch.cuda.is_available = lambda: (_ for _ in ()).throw(AssertionError('the list rerank touched the GPU'))\n"
            "torch.Tensor.cuda = None\n"
            "from ragroute_amd.rerank import rerank_medrag, rerank_wikipedia\n"
            "assert rerank_medrag(list('abcdefgh'), [.1,.9,.5,.9,.3,.2,.7,.5], 4) == (['b','d','g','c'], [0.9,0.9,0.7,0.5])\n"
            "assert rerank_wikipedia(list('abcdefgh'), [.1,.9,.5,.9,.3,.2,.7,.5], 4) == (['a','f','e','c'], [0.1,0.2,0.3,0.5])\n") % ROOT
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1"))


def test_tie_report_of_the_faiss_probe():
    """The report the faiss probe prints (tests/test_faiss_parity_gpu.py; skipped without faiss) on a hand-made case: same id
    set, different order inside a run of equal scores."""
    from tests.util import tie_report
    Df = np.array([[5, 4, 4, 4, 1], [3, 3, 2, 1, 0]], np.float32)
    If = np.array([[9, 7, 2, 5, 1], [4, 6, 0, 1, 2]], np.int64)         # "faiss": first run not in ascending id order
    I = np.array([[9, 2, 5, 7, 1], [4, 6, 0, 1, 2]], np.int64)          # this build: ascending ids inside ties
    r = tie_report(I, If, Df)
    assert r == {"ties_set_identical": True, "ties_order_identical": False, "tie_runs_in_faiss_result": 2,
                 "faiss_tie_runs_in_ascending_id_order": 1}
