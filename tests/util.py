"""Shared helpers for the parity tests (oracle = checker, ragroute_amd = thing under test)."""
import numpy as np


def half_round(x, dtype="fp16"):
    """Round f32 values to the storage dtype and back, so oracle and GPU see identical inputs."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.to(torch.float16 if dtype == "fp16" else torch.bfloat16).to(torch.float32)
    return t.numpy()


def int_data(rng, n, d, lo=-2, hi=3):
    """Small-integer embeddings: every dot product is an exact f32 integer in any summation order,
    so GPU scores must equal the oracle's bit for bit and ties are plentiful."""
    return rng.integers(lo, hi, size=(n, d)).astype(np.float32)


def assert_topk_close(D, I, Dref, Iref, S=None, tol=1e-3):
    """Tolerance-aware parity: scores within tol position by position; ids identical wherever the
    oracle's ranking is unambiguous at tol (neighbouring gaps > 2*tol), and otherwise every returned
    id must be a legitimate member (its oracle score within tol of the position's score)."""
    assert D.shape == Dref.shape and I.shape == Iref.shape
    fin = np.isfinite(Dref)
    assert np.array_equal(np.isfinite(D), fin)
    assert np.allclose(D[fin], Dref[fin], atol=tol, rtol=0), float(np.abs(D[fin] - Dref[fin]).max())
    assert np.array_equal(I[~fin], Iref[~fin])
    nq, k = I.shape
    bad = 0
    for q in range(nq):
        if np.array_equal(I[q], Iref[q]):
            continue
        for j in range(k):
            if I[q, j] == Iref[q, j]:
                continue
            # ambiguous only if a neighbour of the oracle's position j is within 2*tol
            lo = Dref[q, j - 1] if j > 0 else np.inf
            hi = Dref[q, j + 1] if j + 1 < k else (-np.inf if S is None else np.partition(S[q], -k - 1)[-k - 1] if S.shape[1] > k else -np.inf)
            near = (lo - Dref[q, j] <= 2 * tol) or (Dref[q, j] - hi <= 2 * tol)
            assert near, f"query {q} pos {j}: id {I[q, j]} != {Iref[q, j]} with unambiguous oracle gap"
            if S is not None:
                assert abs(S[q, I[q, j]] - D[q, j]) <= tol
            bad += 1
    return bad


def tie_report(I, If, Df):
    """How FAISS orders EQUAL scores against this build's rule (ascending row id, include/ragroute_hip.h rr_flat_search): per query, the
    returned id SET (which rows make it past a tie at the k-th score) and the ORDER inside runs of equal scores."""
    set_same = all(set(a.tolist()) == set(b.tolist()) for a, b in zip(I, If))
    order_same = bool(np.array_equal(I, If))
    runs = asc = 0
    for q in range(If.shape[0]):
        j = 0
        while j < If.shape[1]:
            e = j
            while e + 1 < If.shape[1] and Df[q, e + 1] == Df[q, j]:
                e += 1
            if e > j:
                runs += 1
                asc += int(np.all(np.diff(If[q, j:e + 1]) > 0))
            j = e + 1
    return {"ties_set_identical": bool(set_same), "ties_order_identical": order_same, "tie_runs_in_faiss_result": runs,
            "faiss_tie_runs_in_ascending_id_order": asc}



# ---- synthetic router cases (shared by tests/golden/make_golden.py and the tests) -----------------
def synth_router_case(dataset, seed, n_queries=8):
    """Seeded synthetic router: state_dict, centroids, scaler statistics and query embeddings with the
    reference's shapes (router.py:32-34, config.py:32-101).  numpy's PCG64 stream is version-stable, so
    only the seed and the expected outputs need to be stored as fixtures."""
    from ragroute_amd import config as C
    rng = np.random.default_rng(seed)
    d_max = C.EMBEDDING_MAX_LENGTH[dataset]
    sources = C.DATA_SOURCES[dataset]
    n_in = C.ROUTER_INPUT_DIMENSION[dataset]
    u = lambda shape, b: rng.uniform(-b, b, size=shape).astype(np.float32)  # noqa: E731
    sd = {"fc1.weight": u((256, n_in), 3.0 / np.sqrt(n_in)), "fc1.bias": u((256,), 0.1),
          "ln1.weight": (1 + 0.2 * rng.standard_normal(256)).astype(np.float32), "ln1.bias": u((256,), 0.2),
          "fc2.weight": u((128, 256), 2.0 / 16), "fc2.bias": u((128,), 0.1),
          "ln2.weight": (1 + 0.2 * rng.standard_normal(128)).astype(np.float32), "ln2.bias": u((128,), 0.2),
          "fc3.weight": u((1, 128), 0.3), "fc3.bias": u((1,), 0.05)}
    model_dims = {}
    for s in sources:
        m = C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][s][0]
        if m not in model_dims:
            model_dims[m] = d_max if dataset != "feb4rag" else int(rng.choice([768, 1024, 4096]))
    centroids = {}
    for s in sources:
        dm = model_dims[C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][s][0]]
        centroids[s] = (0.5 * rng.standard_normal(dm)).astype(np.float32)
    scaler = None
    if dataset in ("medrag", "wikipedia"):
        scaler = (0.1 * rng.standard_normal(n_in), rng.uniform(0.5, 2.0, size=n_in))
    queries = [{m: rng.standard_normal(dm).astype(np.float32) for m, dm in model_dims.items()} for _ in range(n_queries)]
    return {"dataset": dataset, "sources": sources, "sd": sd, "centroids": centroids, "scaler": scaler, "queries": queries,
            "d_max": d_max}


def synth_medrag_corpus(seed, n=600, d=768):
    """Small integer-valued 'pubmed'-shaped data source: rows, metadatas (row -> {index, source}) and chunk files."""
    rng = np.random.default_rng(seed)
    xb = int_data(rng, n, d)
    books = ["bookA", "bookB", "bookC"]
    per = n // len(books)
    metadatas = [{"index": i % per, "source": books[min(i // per, len(books) - 1)]} for i in range(n)]
    chunks = {b: [{"id": f"{b}_{j}", "title": f"title {b} {j}", "content": f"content of {b} chunk {j}"} for j in range(per + n)] for b in books}
    return xb, metadatas, chunks


def flat_golden_inputs(case):
    """Seeded inputs of the flat-search fixture tests/golden/flat_search.npz (N=4096, d=768, B=8, k=32,
    fp16-rounded).  Only the expected outputs are stored; the inputs are regenerated from the seed (numpy's PCG64 streams
    are stable across versions)."""
    rng = np.random.default_rng({"gauss": 101, "int": 102, "cosine": 103}[case])
    n, d, nq = 4096, 768, 8
    if case == "int":
        return int_data(rng, n, d), int_data(rng, nq, d)
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    if case == "cosine":   # normalise in f64, then round to the stored precision: what the index holds
        xb = (xb / np.linalg.norm(xb.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
        xq = (xq / np.linalg.norm(xq.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    else:
        xb /= np.float32(np.sqrt(d))
    return half_round(xb), half_round(xq)
