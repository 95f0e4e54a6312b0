"""Device-side glue of the whole hot path: router output feeds the shard scans and the merge
without the JSON float-list hops of the reference (router.py:317-319 -> http_server.py:205-209 -> data_source.py:113-114).

    queries f32 [B, d] on device
      -> router MLP (K3)                   mask [B, C]
      -> per local shard: convert (K0) + scan/top-k (K1/K2) with the mask column folded in,
         results written straight into this rank's packed candidate buffer (one slot per local shard)
      -> [N>1] ONE RCCL all_gather of the packed buffer
      -> merge (K4)                        D f32 [B, k], I i64 [B, k] (global ids: shard << 40 | row)
Everything is enqueued on the current stream; nothing synchronises with the host.

The reference's counterpart is the fan-out / gather / concat / rerank of http_server.py:198-209, 227-257, 280-293: one
message per selected source, replies collected in arrival order, flat candidate lists merged by score."""
from .sharded import SHARD_SHIFT, alloc_packed, exchange_packed, max_over_ranks, merge_gathered


class RetrievalPipeline:
    def __init__(self, shards, shard_ids, router=None, group=None, slots=None):
        """shards: FlatIndex objects local to this rank (any mix of widths: FeB4RAG's sources are 768 / 1024 / 4096 wide,
        config.py:45-57); shard_ids: their global source ids (columns of the router mask); router: FoldedRouter (or None =
        routing strategy "all"); slots: candidate slots per rank in the exchange = the largest local shard count over the
        ranks (default: agreed on by one all_reduce at the first search)."""
        if len(shards) != len(shard_ids):
            raise ValueError("one shard id per shard")
        self.shards = list(shards)
        self.shard_ids = [int(s) for s in shard_ids]
        self.router = router
        self.group = group
        self.slots = slots
        self._packed = {}
        self.stage_events = None   # measurement aid (bench.py): [(exchange start, exchange end = merge start, merge end)] per search

    def time_stages(self, on=True):
        """Bracket the exchange and the merge of every following search with events on the current stream (bench.py's
        `exchange_ms` / `merge_ms`); stage_ms() reads them back.  Off by default: three event records per search."""
        self.stage_events = [] if on else None

    def stage_ms(self):
        """(mean exchange ms, mean merge ms) over the searches since time_stages(); synchronises on the last event."""
        ev = self.stage_events or []
        if not ev:
            return None, None
        ev[-1][2].synchronize()
        return (sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev), sum(b.elapsed_time(c) for _, b, c in ev) / len(ev))

    def route(self, xq_models):
        """xq_models: f32 [B, n_models, d_max] -> (logits, bool mask [B, C]) on device, or (None, None) for 'all'."""
        if self.router is None:
            return None, None
        return self.router.run(xq_models)

    def _buffers(self, B, k, device):
        if self.slots is None:
            self.slots = max(1, max_over_ranks(len(self.shards), device, self.group))
        if len(self.shards) > self.slots:
            raise ValueError(f"{len(self.shards)} local shards but only {self.slots} exchange slots")
        key = (B, k)
        if key not in self._packed:
            buf, D, I = alloc_packed(B, k, device, self.slots)
            if self.slots == 1:
                D, I = D[None], I[None]
            self._packed = {key: (buf, D, I)}
        return self._packed[key]

    def search(self, xq, k, xq_models=None):
        """xq: f32 CUDA [B, d] query embeddings for the shards, or a dict {shard_id: [B, d_shard]} when sources use
        different encoders (FeB4RAG: http_server.py:201-209 picks the embedding of each source's model);
        xq_models: router input [B, n_models, d_max] (defaults to xq as the single model; required for routing when xq is a
        dict).  Returns (D f32 [B,k], I i64 [B,k]) on device, best first, ids = shard << 40 | row."""
        per_shard = isinstance(xq, dict)
        if per_shard:
            _, mask = (None, None) if xq_models is None else self.route(xq_models)
            first = next(iter(xq.values()))
        else:
            _, mask = self.route(xq[:, None, :].contiguous() if xq_models is None else xq_models)
            first = xq
        B = first.shape[0]
        buf, D, I = self._buffers(B, k, first.device)
        for slot, (idx, sid) in enumerate(zip(self.shards, self.shard_ids)):
            q = xq[sid] if per_shard else xq
            idx.search_prepared(idx.prepare_queries(q), k, id_offset=sid << SHARD_SHIFT, out=(D[slot], I[slot]),
                                route_mask=None if mask is None else mask[:, sid])
        if self.stage_events is None:
            out = exchange_packed(buf, self.group)                       # C1: the ONE collective
            return merge_gathered(out, B, k, self.slots, k, True)        # K4, reading the gathered buffer where it lies
        import torch
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = exchange_packed(buf, self.group)
        e1.record()
        res = merge_gathered(out, B, k, self.slots, k, True)
        e2.record()
        self.stage_events.append((e0, e1, e2))
        return res
