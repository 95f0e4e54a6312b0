"""CPU: pin the oracle against the golden vectors produced by the imported reference
(tests/golden/make_golden.py) and against an independent numpy statement."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import half_round, int_data, synth_router_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_rerank_matches_reference_golden():
    g = json.load(open(os.path.join(GOLD, "rerank.json")))
    for c in g["cases"] + g["f64_cases"]:   # (f64_cases: scores that differ only beyond float32)
        assert list(map(list, O.rerank_medrag(c["docs"], c["scores"], c["k"]))) == c["medrag"]
        assert list(map(list, O.rerank_wikipedia(c["docs"], c["scores"], c["k"]))) == c["wikipedia"]
    t = g["ties"]
    assert O.rerank_medrag(t["docs"], t["scores"], t["k"])[1] == t["medrag_scores"]  # tie ORDER is unspecified upstream
    rel = {q: [tuple(x) for x in v] for q, v in g["feb4rag"]["relevance"].items()}
    for c in g["feb4rag"]["cases"]:
        d, i = O.rerank_feb4rag(c["ids"], c["docs"], c["query_id"], c["k"], rel)
        assert d == c["out_docs"] and i == c["out_ids"]


@pytest.mark.parametrize("dataset", ["medrag", "feb4rag", "wikipedia"])
def test_router_matches_reference_golden(dataset):
    from ragroute_amd import config as C
    g = json.load(open(os.path.join(GOLD, "router.json")))[dataset]
    case = synth_router_case(dataset, g["seed"])
    d_max = case["d_max"]
    cen = {c: np.pad(v, (0, d_max - len(v))) for c, v in case["centroids"].items()}
    mos = {s: C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][s][0] for s in case["sources"]}
    s2i = {"medrag": C.MEDRAG_SOURCE_TO_ID, "feb4rag": C.FEB4RAG_SOURCE_TO_ID, "wikipedia": None}[dataset]
    mean, scale = case["scaler"] if case["scaler"] is not None else (None, None)
    for q, want_logits, want_sel in zip(case["queries"], g["logits"], g["selected"]):
        logits = O.router_logits(dataset, case["sources"], mos, s2i, d_max, q, cen, case["sd"], mean, scale)
        assert np.allclose(logits, want_logits, atol=2e-5, rtol=0)
        sel = O.router_select(dataset, case["sources"], np.asarray(want_logits, np.float32))
        assert sel == want_sel


def test_flat_search_c_vs_numpy():
    rng = np.random.default_rng(1)
    xb = half_round(rng.standard_normal((3000, 96)).astype(np.float32))
    xq = half_round(rng.standard_normal((11, 96)).astype(np.float32))
    D, I = O.flat_search_ip(xb, xq, 16)
    D2, I2 = O.flat_search_ip_numpy(xb, xq, 16)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)


def test_flat_search_ties_padding_nan():
    xb = int_data(np.random.default_rng(2), 50, 8, 0, 2)
    xq = int_data(np.random.default_rng(3), 4, 8, 0, 2)
    D, I = O.flat_search_ip(xb, xq, 64)
    D2, I2 = O.flat_search_ip_numpy(xb, xq, 64)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)
    assert (I[:, 50:] == -1).all() and np.isneginf(D[:, 50:]).all()
    for q in range(4):  # ties by ascending id
        for j in range(49):
            assert D[q, j] > D[q, j + 1] or (D[q, j] == D[q, j + 1] and I[q, j] < I[q, j + 1])
    xb[7, 0] = np.nan
    _, I3 = O.flat_search_ip(xb, xq, 50)
    assert 7 not in I3 and (I3[:, -1] == -1).all()
    D0, I0 = O.flat_search_ip(np.zeros((0, 8), np.float32), xq, 3)
    assert (I0 == -1).all() and np.isneginf(D0).all()


def test_single_query_form_agrees():
    rng = np.random.default_rng(4)
    xb = rng.standard_normal((20000, 64)).astype(np.float32)
    xq = rng.standard_normal((3, 64)).astype(np.float32)
    D, I = O.flat_search_ip(xb, xq, 10)
    for q in range(3):
        D1, I1 = O.flat_search_ip_single(xb, xq[q], 10)
        assert np.array_equal(I1[0], I[q]) and np.allclose(D1[0], D[q], atol=1e-4)


def test_normalize_l2():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((10, 33)).astype(np.float32) * 5
    x[3] = 0
    y = x.copy()
    O.normalize_L2(y)
    assert np.allclose(np.linalg.norm(np.delete(y, 3, 0), axis=1), 1, atol=1e-6)
    assert (y[3] == 0).all()
    assert np.allclose(y[0], x[0] / np.linalg.norm(x[0]), atol=1e-6)


def test_merge_topk_oracle():
    rng = np.random.default_rng(6)
    D = rng.standard_normal((5, 40)).astype(np.float32)
    I = rng.permutation(1000)[:200].reshape(5, 40).astype(np.int64)
    I[0, :5] = -1
    Do, Io = O.merge_topk(D, I, 8, True)
    for q in range(5):
        valid = I[q] >= 0
        order = np.lexsort((I[q][valid], -D[q][valid].astype(np.float64)))[:8]
        assert np.array_equal(Io[q], I[q][valid][order]) and np.array_equal(Do[q], D[q][valid][order])
    Da, Ia = O.merge_topk(D, I, 8, False)
    assert np.all(np.diff(Da, axis=1) >= 0)
    Dp, Ip = O.merge_topk(D[:, :3], np.full((5, 3), -1, np.int64), 4, True)
    assert (Ip == -1).all() and np.isneginf(Dp).all()


@pytest.mark.parametrize("case", ["gauss", "int", "cosine"])
def test_flat_search_matches_committed_fixture(case):
    """tests/golden/flat_search.npz: f64 numpy statement of flat-IP top-k on seeded fp16-rounded inputs."""
    from tests.util import flat_golden_inputs
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "flat_search.npz"))
    xb, xq = flat_golden_inputs(case)
    D, I = O.flat_search_ip(xb, xq, 32)
    if case == "int":
        assert np.array_equal(I, G["int_I"]) and np.array_equal(D.astype(np.float64), G["int_D"])
    else:
        assert np.array_equal(I, G[case + "_I"])           # f32 summation error << the score gaps of this fixture
        assert np.allclose(D, G[case + "_D"], atol=1e-3, rtol=0)
