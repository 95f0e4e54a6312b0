"""Cross-source merge — mirror of reference ragroute/rerank.py with the same names and return shapes.

`rerank_medrag` / `rerank_wikipedia` select on the GPU (C ABI `rr_merge_topk`, csrc/select.hip); the
candidate position is the id, so ties keep the earlier candidate (numpy's argsort leaves tie order
unspecified, rerank.py:5,30).  `rerank_feb4rag` orders by ground-truth qrels, not by scores
(rerank.py:12-25): a host dictionary sort, kept as such.
`merge_topk` is the batched device form used after the multi-GPU candidate all-gather."""
import numpy as np
import torch

from ._lib import check, lib


def merge_topk(D, I, k, descending=True):
    """D f32 [nq,m], I i64 [nq,m] CUDA tensors -> (D [nq,k], I [nq,k]) best-first; ties by ascending id;
    id < 0 marks padding.  Enqueued on the current stream."""
    if D.shape != I.shape or D.dim() != 2:
        raise ValueError("merge_topk needs D and I of the same [nq,m] shape")
    D = D.to(torch.float32).contiguous()
    I = I.to(torch.int64).contiguous()
    nq, m = D.shape
    Do = torch.empty((nq, k), dtype=torch.float32, device=D.device)
    Io = torch.empty((nq, k), dtype=torch.int64, device=D.device)
    check(lib().rr_merge_topk(D.data_ptr(), I.data_ptr(), nq, m, k, int(bool(descending)), Do.data_ptr(), Io.data_ptr(),
                              torch.cuda.current_stream().cuda_stream), "rr_merge_topk")
    return Do, Io


def _rerank_by_score(docs, scores, k, descending):
    n = len(scores)
    if n == 0:
        return [], []
    if len(docs) != n:
        raise ValueError("docs and scores must have the same length")
    D = torch.tensor(np.asarray(scores, np.float64).astype(np.float32)[None, :], device="cuda")
    I = torch.arange(n, dtype=torch.int64, device="cuda")[None, :]
    _, order = merge_topk(D, I, min(k, n), descending)
    order = order[0].cpu().tolist()
    order = [i for i in order if i >= 0]
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
