"""Device-side glue of the whole hot path: router output feeds the shard scans and the merge
without the JSON float-list hops of the reference (router.py:317-319 -> http_server.py:205-209 -> data_source.py:113-114).

    queries f32 [B, d] on device
      -> router MLP (K3)                   mask [B, C]
      -> per local shard: convert (K0) + scan/top-k (K1/K2) with the mask column folded in
      -> [N>1] RCCL all_gather of candidates
      -> merge (K4)                        D f32 [B, k], I i64 [B, k] (global ids: shard << 40 | row)
Everything is enqueued on the current stream; nothing synchronises with the host."""
import torch

from .rerank import merge_topk
from .sharded import SHARD_SHIFT, alloc_packed, gather_candidates, gather_packed


class RetrievalPipeline:
    def __init__(self, shards, shard_ids, router=None, group=None):
        """shards: FlatIndex objects local to this rank; shard_ids: their global source ids (columns of the router mask);
        router: FoldedRouter (or None = routing strategy "all")."""
        self.shards = list(shards)
        self.shard_ids = [int(s) for s in shard_ids]
        self.router = router
        self.group = group
        self._packed = {}

    def route(self, xq_models):
        """xq_models: f32 [B, n_models, d_max] -> (logits, bool mask [B, C]) on device, or (None, None) for 'all'."""
        if self.router is None:
            return None, None
        return self.router.run(xq_models)

    def search(self, xq, k, xq_models=None):
        """xq: f32 CUDA [B, d] query embeddings for the shards, or a dict {shard_id: [B, d_shard]} when sources use
        different encoders (FeB4RAG: 768 / 1024 / 4096 wide, config.py:44-58, http_server.py:201-209 picks the embedding of
        each source's model); xq_models: router input [B, n_models, d_max] (defaults to xq as the single model)."""
        if isinstance(xq, dict):
            return self._search_per_shard(xq, k, xq_models)
        _, mask = self.route(xq[:, None, :].contiguous() if xq_models is None else xq_models)
        B = xq.shape[0]
        if len(self.shards) == 1:
            # one shard per rank (the benchmark layout): results land in a packed buffer, ONE collective moves it
            key = (B, k)
            if key not in self._packed:
                self._packed = {key: alloc_packed(B, k, xq.device)}
            buf, D, I = self._packed[key]
            idx, sid = self.shards[0], self.shard_ids[0]
            idx.search_prepared(idx.prepare_queries(xq), k, id_offset=sid << SHARD_SHIFT, out=(D, I),
                                route_mask=None if mask is None else mask[:, sid])
            Dg, Ig = gather_packed(buf, B, k, self.group)
            return merge_topk(Dg, Ig, k, True)
        Ds, Is = [], []
        for idx, sid in zip(self.shards, self.shard_ids):
            D, I = idx.search_prepared(idx.prepare_queries(xq), k, id_offset=sid << SHARD_SHIFT,
                                       route_mask=None if mask is None else mask[:, sid])
            Ds.append(D)
            Is.append(I)
        Dg, Ig = gather_candidates(torch.cat(Ds, 1), torch.cat(Is, 1), self.group)
        return merge_topk(Dg, Ig, k, True)

    def _search_per_shard(self, xq_by_shard, k, xq_models):
        _, mask = (None, None) if xq_models is None else self.route(xq_models)
        Ds, Is = [], []
        for idx, sid in zip(self.shards, self.shard_ids):
            xq = xq_by_shard[sid]
            D, I = idx.search_prepared(idx.prepare_queries(xq), k, id_offset=sid << SHARD_SHIFT,
                                       route_mask=None if mask is None else mask[:, sid])
            Ds.append(D)
            Is.append(I)
        Dg, Ig = gather_candidates(torch.cat(Ds, 1), torch.cat(Is, 1), self.group)
        return merge_topk(Dg, Ig, k, True)
