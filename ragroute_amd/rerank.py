"""Cross-source merge — mirror of reference ragroute/rerank.py with the same names and return shapes.

`rerank_medrag` / `rerank_wikipedia` are HOST functions, as in the reference (its aiohttp front-end calls them on Python
lists of <= S*k floats, http_server.py:288-293; `north_star`: "host-side rerank merge"): numpy, no HIP context needed.  They
compare the scores AS GIVEN (float64, what numpy makes of the reference's Python list: rerank.py:5, 30); the earlier candidate
wins a tie (numpy's argsort leaves tie order unspecified).  For scores that came from float32 — every score a data source
returns (faiss D is f32, data_source.py:187) — this is the total order the device merge defines.  `rerank_feb4rag` orders by ground-truth qrels, not by scores (rerank.py:12-25): a host
dictionary sort, kept as such.
`merge_topk` (C ABI `rr_merge_topk`, csrc/select.hip) is the batched DEVICE form, used where the candidates already are on
the device: after the multi-GPU candidate all-gather (sharded.py, pipeline.py)."""
import numpy as np
import torch

from ._lib import check, lib


MERGE_MAX = 8192   # candidates per query one rr_merge_topk launch sorts in LDS (csrc/rr_common.h kSelectCap)


def merge_topk(D, I, k, descending=True):
    """D f32 [nq,m], I i64 [nq,m] CUDA tensors -> (D [nq,k], I [nq,k]) best-first; ties by ascending id;
    id < 0 marks padding.  Enqueued on the current stream.

    Contract beyond the reference's one-query list merge (rerank.py:3-9, 28-34):
      * NaN-scored candidates are dropped (treated as padding).  The scan kernels never emit one (a NaN score is never
        selected), so after the candidate exchange there is nothing to drop; the list-shaped host functions `rerank_medrag`
        / `rerank_wikipedia` below place NaNs where numpy does.
      * m > 8192 is merged in rounds (per-slice top-k, then a merge of the survivors), which needs k <= 4096."""
    if D.shape != I.shape or D.dim() != 2:
        raise ValueError("merge_topk needs D and I of the same [nq,m] shape")
    D = D.to(torch.float32).contiguous()
    I = I.to(torch.int64).contiguous()
    nq, m = D.shape
    if m > MERGE_MAX:
        if k > MERGE_MAX // 2:
            raise ValueError(f"merge_topk: {m} candidates per query need k <= {MERGE_MAX // 2}, got k={k}")
        parts = [merge_topk(D[:, s:s + MERGE_MAX], I[:, s:s + MERGE_MAX], k, descending) for s in range(0, m, MERGE_MAX)]
        return merge_topk(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k, descending)
    Do = torch.empty((nq, k), dtype=torch.float32, device=D.device)
    Io = torch.empty((nq, k), dtype=torch.int64, device=D.device)
    check(lib().rr_merge_topk(D.data_ptr(), I.data_ptr(), nq, m, k, int(bool(descending)), Do.data_ptr(), Io.data_ptr(),
                              torch.cuda.current_stream().cuda_stream), "rr_merge_topk")
    return Do, Io


def _rerank_by_score(docs, scores, k, descending):
    """The reference sorts the float64 list with numpy on the HOST (rerank.py:5, 30; called from the aiohttp front-end,
    http_server.py:288-293), and so does this: a <= 128-element sort needs no HIP context in the front-end process.
    Scores are compared as given (float64, like np.argsort on the reference's list; two scores that differ only beyond float32
    keep the reference's order — tests/golden/rerank.json "f64_cases"), candidate position is the tie-break (earlier first;
    numpy's default sort leaves ties unspecified), and NaN scores are placed where numpy's argsort puts them — at the END of the
    ascending order: `[::-1]` therefore ranks them FIRST for rerank_medrag, rerank_wikipedia keeps them last.  On float32-born
    scores this is exactly `rr_merge_topk`'s order."""
    n = len(scores)
    if n == 0:
        return [], []
    if len(docs) != n:
        raise ValueError("docs and scores must have the same length")
    if k <= 0:
        return [], []
    s64 = np.asarray(scores, np.float64)
    nan = np.isnan(s64)
    nan_pos = np.flatnonzero(nan)
    valid = np.flatnonzero(~nan)
    v = s64[valid]
    ranked = valid[np.lexsort((valid, -v if descending else v))]      # IEEE compare (-0.0 == 0.0); the earlier candidate wins a tie
    order = (list(nan_pos[::-1]) + list(ranked)) if descending else (list(ranked) + list(nan_pos))
    order = [int(i) for i in order[:k]]
    return [docs[i] for i in order], [scores[i] for i in order]


def rerank_medrag(docs, scores, k):
    """k highest-scoring docs, best first (rerank.py:3-9)."""
    return _rerank_by_score(docs, scores, k, True)


def rerank_wikipedia(docs, scores, k):
    """k LOWEST-scoring docs, ascending — exactly what the reference does (rerank.py:28-34)."""
    return _rerank_by_score(docs, scores, k, False)


def rerank_feb4rag(ids, docs, query_id, k, relevance_data):
    """Order candidates by qrels grade, unknown docs last in input order; returns (docs, ids) (rerank.py:12-25)."""
    rel = relevance_data.get(query_id, [])
    rel_order = [docid for docid, _ in sorted(rel, key=lambda x: -int(x[1]))]
    sort_key = {docid: i for i, docid in enumerate(rel_order)}
    data = sorted(zip(ids, docs), key=lambda x: sort_key.get(x[0], float("inf")))
    sorted_ids, sorted_docs = zip(*data) if data else ([], [])
    return list(sorted_docs[:k]), list(sorted_ids[:k])
