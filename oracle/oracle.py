"""oracle/oracle.py — TEST INFRASTRUCTURE, not product code.

CPU restatement (numpy + the plain-C library built from flat_oracle.c) of the reference's numeric
call sites on the retrieval hot path.  Each function cites the reference file:line it follows
(paths relative to the reference repository root).

Pinning status
  * merge (rerank_*)  : PINNED — checked against the imported reference `ragroute/rerank.py` via the
                        golden vectors in tests/golden/rerank_*.npz (tests/golden/make_golden.py).
  * router            : PINNED — checked against the imported reference `ragroute/router.py`
                        (CorpusRoutingNN + Router.select_relevant_sources_ragroute, transport and
                        encoder imports stubbed, no arithmetic in the stubs) via tests/golden/router_*.npz.
  * flat search / normalize_L2 : PARITY UNPINNED — the arithmetic lives in the third-party wheel
                        faiss-cpu==1.7.4 (environment.yml:55), absent from /root/reference and from this
                        image; the reference has no tests or golden vectors.  Restated from FAISS's
                        published IndexFlatIP / normalize_L2 contract and the reference's call sites.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/flat_oracle.c -> oracle/liboracle.so (gcc)."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        try:
            lib = ctypes.CDLL(path)
        except OSError:
            os.remove(path)
            build()
            lib = ctypes.CDLL(path)
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        lib.oracle_flat_search_ip.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, f32p, i64p]
        lib.oracle_flat_search_ip.restype = None
        lib.oracle_flat_search_l2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, f32p, i64p]
        lib.oracle_flat_search_l2.restype = None
        lib.oracle_flat_search_ip_f32_single.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, f32p, i64p]
        lib.oracle_flat_search_ip_f32_single.restype = None
        lib.oracle_normalize_l2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64]
        lib.oracle_normalize_l2.restype = None
        lib.oracle_merge_topk.argtypes = [f32p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, i64p]
        lib.oracle_merge_topk.restype = None
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_set_threads.restype = None
        _LIB = lib
    return _LIB


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def num_threads():
    return int(_lib().oracle_num_threads())


def set_threads(n):
    """Cap the OpenMP team (bench.py sizes it to the CPUs this process may actually use)."""
    _lib().oracle_set_threads(int(n))


# ---- a2: index.search — data_source.py:158, 186, 203 -----------------------------------------
def flat_search_ip(xb, xq, k):
    """(D f32[nq,k], I i64[nq,k]) of an exact inner-product flat index; ties by ascending id,
    (-inf, -1) padding when k > ntotal.  Scores are f32 roundings of f64 dot products.
    The order INSIDE a run of equal scores (ascending id) is this build's definition: FAISS 1.7.4's is unverified here
    (parity unpinned, see the module header); tests/test_faiss_parity_gpu.py reports it where faiss is importable."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    xq = np.ascontiguousarray(xq, dtype=np.float32).reshape(-1, xb.shape[1] if xb.ndim == 2 and xb.shape[0] else xq.shape[-1])
    nq, d = xq.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    _lib().oracle_flat_search_ip(_f32(xb), xb.shape[0], d, _f32(xq), nq, k, _f32(D), _i64(I))
    return D, I


def flat_search_l2(xb, xq, k):
    """faiss.IndexFlatL2 contract: (D = squared L2 distances f32[nq,k] ascending, I i64[nq,k]); ties by ascending id;
    (+inf, -1) padding.  Not used by the reference's code directly (the index FILE decides the metric, data_source.py:71)."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    nq, d = xq.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    _lib().oracle_flat_search_l2(_f32(xb), xb.shape[0], d, _f32(xq), nq, k, _f32(D), _i64(I))
    return D, I


def flat_search_ip_single(xb, xq_row, k):
    """The reference's call shape: ONE f32 query per call (data_source.py:113-114), f32 arithmetic,
    rows split over the host cores.  Timed by bench.py as the CPU baseline."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    q = np.ascontiguousarray(xq_row, dtype=np.float32).reshape(-1)
    D = np.empty((1, k), np.float32)
    I = np.empty((1, k), np.int64)
    _lib().oracle_flat_search_ip_f32_single(_f32(xb), xb.shape[0], xb.shape[1], _f32(q), k, _f32(D), _i64(I))
    return D, I


def flat_search_ip_numpy(xb, xq, k):
    """Independent numpy statement of the same contract (cross-checks the C version)."""
    xb = np.asarray(xb, np.float32)
    xq = np.asarray(xq, np.float32)
    S = (xq.astype(np.float64) @ xb.astype(np.float64).T).astype(np.float32)
    n = xb.shape[0]
    D = np.full((xq.shape[0], k), -np.inf, np.float32)
    I = np.full((xq.shape[0], k), -1, np.int64)
    for q in range(xq.shape[0]):
        order = np.lexsort((np.arange(n), -S[q].astype(np.float64)))  # score desc, id asc
        order = order[~np.isnan(S[q][order])][:k]
        D[q, : len(order)] = S[q][order]
        I[q, : len(order)] = order
    return D, I


# ---- a3: faiss.normalize_L2 — data_source.py:198-199 -------------------------------------------
def normalize_L2(x):
    """In place; f32; zero-norm rows unchanged."""
    assert x.dtype == np.float32 and x.flags.c_contiguous and x.ndim == 2
    _lib().oracle_normalize_l2(_f32(x), x.shape[0], x.shape[1])


# ---- a11/a12: score merge — rerank.py:3-9, 28-34; caller http_server.py:280-293 ------------------
def merge_topk(D, I, k, descending=True):
    D = np.ascontiguousarray(D, np.float32)
    I = np.ascontiguousarray(I, np.int64)
    nq, m = D.shape
    Do = np.empty((nq, k), np.float32)
    Io = np.empty((nq, k), np.int64)
    _lib().oracle_merge_topk(_f32(D), _i64(I), nq, m, k, int(bool(descending)), _f32(Do), _i64(Io))
    return Do, Io


def rerank_medrag(docs, scores, k):
    """rerank.py:3-9: k highest scores (ties: first occurrence first — numpy leaves them unspecified)."""
    if len(scores) == 0:
        return [], []
    order = np.lexsort((np.arange(len(scores)), -np.asarray(scores, np.float64)))[:k]
    return [docs[i] for i in order], [scores[i] for i in order]


def rerank_wikipedia(docs, scores, k):
    """rerank.py:28-34: k LOWEST scores (ascending argsort, as the reference has it)."""
    if len(scores) == 0:
        return [], []
    order = np.lexsort((np.arange(len(scores)), np.asarray(scores, np.float64)))[:k]
    return [docs[i] for i in order], [scores[i] for i in order]


def rerank_feb4rag(ids, docs, query_id, k, relevance_data):
    """rerank.py:12-25: order by qrels grade (desc, stable), unknown docs last in input order;
    returns (docs, ids)."""
    rel = relevance_data.get(query_id, [])
    order = [docid for docid, _ in sorted(rel, key=lambda x: -int(x[1]))]
    rank = {docid: i for i, docid in enumerate(order)}
    data = sorted(zip(ids, docs), key=lambda x: rank.get(x[0], float("inf")))
    return [d for _, d in data][:k], [i for i, _ in data][:k]


# ---- a7/a9: router — router.py:37-55, 241-283 ----------------------------------------------------
def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)  # biased, as torch.nn.LayerNorm
    return (x - mu) / np.sqrt(var + eps) * g + b


def corpus_routing_nn(x, sd):
    """CorpusRoutingNN.forward in eval mode (router.py:50-55); sd = state_dict of f32 numpy arrays."""
    x = np.asarray(x, np.float32)
    h = np.maximum(layer_norm(x @ sd["fc1.weight"].T + sd["fc1.bias"], sd["ln1.weight"], sd["ln1.bias"]), 0).astype(np.float32)
    h = np.maximum(layer_norm(h @ sd["fc2.weight"].T + sd["fc2.bias"], sd["ln2.weight"], sd["ln2.bias"]), 0).astype(np.float32)
    return (h @ sd["fc3.weight"].T + sd["fc3.bias"]).astype(np.float32)


def router_features(dataset, data_sources, model_of_source, source_to_id, d_max, query_embeddings, centroids):
    """router.py:245-267: per corpus [pad(q_model) || centroid || onehot] (float64, as numpy promotes)."""
    rows = []
    for corpus in data_sources:
        q = np.asarray(query_embeddings[model_of_source[corpus]])
        q = np.pad(q, (0, d_max - len(q)))
        if dataset == "wikipedia":
            onehot = np.eye(len(data_sources))[int(corpus)]
        else:
            onehot = np.eye(len(source_to_id))[source_to_id[corpus]]
        rows.append(np.concatenate([q, centroids[corpus], onehot]))
    return np.stack(rows)


def router_logits(dataset, data_sources, model_of_source, source_to_id, d_max, query_embeddings, centroids, sd,
                  scaler_mean=None, scaler_scale=None):
    """router.py:241-275 up to the logits: features -> StandardScaler (medrag, wikipedia) -> f32 -> MLP."""
    x = router_features(dataset, data_sources, model_of_source, source_to_id, d_max, query_embeddings, centroids)
    if scaler_mean is not None:
        x = (x - scaler_mean) / scaler_scale  # sklearn StandardScaler.transform, f64
    return corpus_routing_nn(x.astype(np.float32), sd).reshape(-1)


def router_select(dataset, data_sources, logits):
    """router.py:276-282: sigmoid, > 0.4924 (medrag) or > 0.5, names in data_sources order."""
    p = (1.0 / (1.0 + np.exp(-logits.astype(np.float32)))).astype(np.float32)
    thr = np.float32(0.4924) if dataset == "medrag" else np.float32(0.5)
    return [c for c, keep in zip(data_sources, p > thr) if keep]
