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
from .sharded import SHARD_SHIFT, alloc_gathered, alloc_packed, exchange_packed, max_over_ranks, merge_gathered


class RetrievalPipeline:
    def __init__(self, shards, shard_ids, router=None, group=None, slots=None, share_queries=None, units=None):
        """shards: FlatIndex objects local to this rank (any mix of widths: FeB4RAG's sources are 768 / 1024 / 4096 wide,
        config.py:45-57); shard_ids: their global source ids (columns of the router mask); router: FoldedRouter (or None =
        routing strategy "all"); slots: candidate slots per rank in the exchange = the largest local unit count over the
        ranks (default: agreed on by one all_reduce at the first search).
        share_queries: optional list, one hashable key per shard — shards with the same key receive the SAME query
        embeddings (the key is the encoder: config.py:37-71; MedRAG's four sources all use MedCPT, five FeB4RAG sources
        UAE-Large-V1).  Local shards that share a key, dimension, metric and dtype are packed into one SegmentedIndex and
        searched in ONE pass (`rr_flat_search_segments`) whose result is already their merged top-k: one exchange slot per
        group instead of one per shard.  The shard objects are re-pointed at their slices of the packed matrix.
        units: alternatively, the search units themselves — a list of (FlatIndex | SegmentedIndex, [source ids]) built by the
        caller (a SegmentedIndex filled in place never holds its sources twice)."""
        if len(shards) != len(shard_ids):
            raise ValueError("one shard id per shard")
        self.shards = list(shards)
        self.shard_ids = [int(s) for s in shard_ids]
        self.router = router
        self.group = group
        self.slots = slots
        self._packed = {}
        # units: what one scan call serves and one exchange slot carries: ("shard", FlatIndex, [sid]) or ("segments", SegmentedIndex, [sids])
        # a "shard" unit may be a row slice of its source (placement.py): its ids start at (sid << 40) + row_begin
        self.units = []
        if units is not None:
            from .flat_index import SegmentedIndex
            for u in units:
                obj, sids = u[0], [int(s) for s in u[1]]
                seg = isinstance(obj, SegmentedIndex)
                self.units.append(("segments" if seg else "shard", obj, sids, None if seg else int(u[2]) if len(u) > 2 else sids[0] << SHARD_SHIFT))
        elif share_queries is None:
            self.units = [("shard", idx, [sid], sid << SHARD_SHIFT) for idx, sid in zip(self.shards, self.shard_ids)]
        else:
            if len(share_queries) != len(self.shards):
                raise ValueError("one share_queries key per shard")
            from .flat_index import SegmentedIndex
            groups = {}
            for idx, sid, key in zip(self.shards, self.shard_ids, share_queries):
                if getattr(idx, "metric", None) in ("ip", "cosine"):
                    groups.setdefault((key, idx.d, idx.metric, idx.dtype), []).append((sid, idx))
                else:
                    self.units.append(("shard", idx, [sid], sid << SHARD_SHIFT))
            for members in groups.values():
                members.sort(key=lambda m: m[0])      # ascending source id = ascending id offset: ties by ascending global id
                if len(members) == 1:
                    self.units.append(("shard", members[0][1], [members[0][0]], members[0][0] << SHARD_SHIFT))
                else:
                    sids = [m[0] for m in members]
                    seg = SegmentedIndex.from_indexes([m[1] for m in members], id_offsets=[s << SHARD_SHIFT for s in sids], mask_cols=sids)
                    self.units.append(("segments", seg, sids, None))
            self.units.sort(key=lambda u: u[2][0])
        self.stage_events = None   # measurement aid (bench.py): [(exchange start, exchange end = merge start, merge end)] per search

    @classmethod
    def from_placement(cls, placement, rank, rows_f32=None, fill_half=None, router=None, group=None, device=None):
        """This rank's search units of a placement.Placement (row slices of the federation's sources, balanced over the GPUs).
        The pieces of one encoder group become ONE SegmentedIndex (segment = slice: id offset (sid << 40) + row_begin, mask
        column sid), a lone piece a FlatIndex whose ids start at the slice's offset.  Rows come from exactly one of
          rows_f32(sid, row_begin, row_end) -> float32 [n, d] (numpy or tensor): ingested like FlatIndex.add / SegmentedIndex.fill;
          fill_half(source, row_slice, out): writes the slice's rows into `out`, a device view [n_rows, padded dim] of the
            storage dtype, zero padded (synthetic corpora generated in place: nothing is held twice).
        Exchange slots = placement.slots on every rank (known from the plan: no collective at set-up)."""
        import torch
        from .flat_index import _TORCH_DTYPE, FlatIndex, SegmentedIndex
        from ._lib import RR_MAX_SEGMENTS
        if (rows_f32 is None) == (fill_half is None):
            raise ValueError("give exactly one of rows_f32 / fill_half")
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        units = []
        for unit in placement.ranks[rank]:
            src0 = placement.sources[unit.slices[0].sid]
            for c in range(0, len(unit.slices), RR_MAX_SEGMENTS):      # a segmented search takes 32 segments at most
                slices = unit.slices[c: c + RR_MAX_SEGMENTS]
                if len(slices) == 1:
                    sl = slices[0]
                    idx = FlatIndex(src0.dim, metric=src0.metric, dtype=src0.dtype, device=dev)
                    if fill_half is not None:
                        xb = torch.zeros((max(1, sl.n_rows), idx.dim), dtype=_TORCH_DTYPE[src0.dtype], device=dev)
                        fill_half(placement.sources[sl.sid], sl, xb[: sl.n_rows])
                        idx.adopt(xb, ntotal=sl.n_rows)
                    elif sl.n_rows:
                        idx.add(rows_f32(sl.sid, sl.row_begin, sl.row_begin + sl.n_rows))
                    units.append((idx, [sl.sid], sl.id_offset))
                else:
                    seg = SegmentedIndex(src0.dim, [sl.n_rows for sl in slices], id_offsets=[sl.id_offset for sl in slices],
                                         mask_cols=[sl.sid for sl in slices], metric=src0.metric, dtype=src0.dtype, device=dev)
                    for j, sl in enumerate(slices):
                        if fill_half is not None:
                            fill_half(placement.sources[sl.sid], sl, seg.rows_of(j))
                        elif sl.n_rows:
                            seg.fill(j, rows_f32(sl.sid, sl.row_begin, sl.row_begin + sl.n_rows))
                    units.append((seg, [sl.sid for sl in slices]))
        slots = max(1, max(sum(-(-len(u.slices) // RR_MAX_SEGMENTS) for u in units_r) for units_r in placement.ranks))
        pipe = cls([], [], router=router, group=group, slots=slots, units=units)
        pipe.placement, pipe.rank = placement, rank
        return pipe

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
        return (sum(e[0].elapsed_time(e[1]) for e in ev) / len(ev), sum(e[1].elapsed_time(e[2]) for e in ev) / len(ev))

    def local_ms(self):
        """Mean ms of THIS rank's own work per search (router + query conversion + every local scan, up to the exchange) over
        the searches since time_stages(): the figure the placement balances (bench.py's `per_rank_local_ms`)."""
        ev = self.stage_events or []
        if not ev:
            return None
        ev[-1][2].synchronize()
        return sum(e[3].elapsed_time(e[0]) for e in ev) / len(ev)

    def route(self, xq_models):
        """xq_models: f32 [B, n_models, d_max] -> (logits, bool mask [B, C]) on device, or (None, None) for 'all'."""
        if self.router is None:
            return None, None
        return self.router.run(xq_models)

    def _buffers(self, B, k, device):
        if self.slots is None:
            self.slots = max(1, max_over_ranks(len(self.units), device, self.group))
        if len(self.units) > self.slots:
            raise ValueError(f"{len(self.units)} local search units but only {self.slots} exchange slots")
        key = (B, k)
        if key not in self._packed:
            buf, D, I = alloc_packed(B, k, device, self.slots)
            if self.slots == 1:
                D, I = D[None], I[None]
            self._packed = {key: (buf, D, I)}
            self._gathered = alloc_gathered(buf, self.group)     # the exchange's destination, reused by every search of this (B, k)
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
        if self.stage_events is not None:
            import torch
            e_start = torch.cuda.Event(enable_timing=True)
            e_start.record()
        for slot, (kind, idx, sids, id_offset) in enumerate(self.units):
            q = xq[sids[0]] if per_shard else xq          # (a group shares one encoder: any member's embedding)
            if kind == "segments":
                idx.search_prepared(idx.prepare_queries(q), k, route_mask=mask, out=(D[slot], I[slot]))
            else:
                idx.search_prepared(idx.prepare_queries(q), k, id_offset=id_offset, out=(D[slot], I[slot]),
                                    route_mask=None if mask is None else mask[:, sids[0]])
        if self.stage_events is None:
            out = exchange_packed(buf, self.group, self._gathered)       # C1: the ONE collective
            return merge_gathered(out, B, k, self.slots, k, True)        # K4, reading the gathered buffer where it lies
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = exchange_packed(buf, self.group, self._gathered)
        e1.record()
        res = merge_gathered(out, B, k, self.slots, k, True)
        e2.record()
        self.stage_events.append((e0, e1, e2, e_start))
        return res
