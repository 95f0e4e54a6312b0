"""Federated corpora sharded one (or more) per GPU: local scan, candidate exchange, merge.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).
Replaces the reference's ZeroMQ gather of per-source replies (ragroute/http_server.py:227-257, 280-286)
by ONE all_gather of packed [B,k] (score f32, id i64) candidates, followed by the device merge
(rr_merge_topk) — ragroute/rerank.py:3-9.  The payload is B*k*12 bytes per rank (98 KB at B=256, k=32),
i.e. latency-bound on xGMI, so a single collective per batch and no host synchronisation is the design.

Global ids: row r of shard s is  (s << SHARD_SHIFT) | r, so the host can still map an id to
(source, metadata) as data_source.py:190 does.
"""
import torch
import torch.distributed as dist

SHARD_SHIFT = 40


def global_id(shard, row):
    return (int(shard) << SHARD_SHIFT) | int(row)


def split_global_id(gid):
    return int(gid) >> SHARD_SHIFT, int(gid) & ((1 << SHARD_SHIFT) - 1)


def apply_route_mask(D, I, mask_col):
    """Routing: candidates of a (query, shard) pair the router did not select become padding
    (-inf, -1) before the merge.  mask_col: bool [B] for this shard."""
    keep = mask_col.to(torch.bool)[:, None]
    return torch.where(keep, D, torch.full_like(D, float("-inf"))), torch.where(keep, I, torch.full_like(I, -1))


def gather_candidates(D, I, group=None):
    """all_gather of this rank's [B,k] candidates -> ([B, G*k] scores, [B, G*k] ids), rank-major columns.
    Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return D, I
    B, k = D.shape
    if D.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on one box: stage through the host
        Dc, Ic = gather_candidates(D.cpu(), I.cpu(), group)
        return Dc.to(D.device), Ic.to(I.device)
    Dg = torch.empty((world * B, k), dtype=D.dtype, device=D.device)
    Ig = torch.empty((world * B, k), dtype=I.dtype, device=I.device)
    dist.all_gather_into_tensor(Dg, D.contiguous(), group=group)
    dist.all_gather_into_tensor(Ig, I.contiguous(), group=group)
    return (Dg.view(world, B, k).permute(1, 0, 2).reshape(B, world * k),
            Ig.view(world, B, k).permute(1, 0, 2).reshape(B, world * k))


def alloc_packed(B, k, device, slots=1):
    """One allocation holding D f32[slots,B,k] followed by I i64[slots,B,k] (one slot per local shard): the scan kernels
    write straight into the views and the exchange moves the whole buffer with ONE collective.  Every slot starts as
    padding (-inf, -1), so a rank with fewer shards than `slots` contributes nothing for the unused ones.
    Returns (buf, D, I); with slots == 1 the views are [B,k]."""
    nD = slots * B * k * 4
    pad = (-nD) % 8
    buf = torch.empty(nD + pad + slots * B * k * 8, dtype=torch.uint8, device=device)
    D = buf[:nD].view(torch.float32).view(slots, B, k)
    I = buf[nD + pad:].view(torch.int64).view(slots, B, k)
    D.fill_(float("-inf"))
    I.fill_(-1)
    return (buf, D[0], I[0]) if slots == 1 else (buf, D, I)


def packed_layout(B, k, slots=1):
    """(bytes of the D region, offset of the I region, bytes per rank) of a packed candidate buffer (alloc_packed)."""
    nD = slots * B * k * 4
    pad = (-nD) % 8
    return nD, nD + pad, nD + pad + slots * B * k * 8


def exchange_packed(buf, group=None, out=None):
    """THE collective of the path: all_gather of every rank's packed candidate buffer -> uint8 [world, bytes per rank], left
    exactly as the collective wrote it (`merge_gathered` reads it in place).  One rank: a view of `buf`, nothing is copied.
    out: optional uint8 tensor of world x buf.numel() bytes on buf's device to gather into (the callers keep one per (B, k):
    the merge that follows writes fresh result tensors, so the gathered buffer is free again when the next search's collective
    is enqueued behind it on the stream)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return buf.view(1, buf.numel())
    if out is None or out.numel() != world * buf.numel() or out.device != buf.device:
        out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
    out = out.view(-1)
    if buf.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on one box: stage through the host
        host = torch.empty(world * buf.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        out.copy_(host)
        return out.view(world, buf.numel())
    dist.all_gather_into_tensor(out, buf, group=group)
    return out.view(world, buf.numel())


def alloc_gathered(buf, group=None):
    """The reusable destination of exchange_packed for this packed buffer (None on one rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return None if world == 1 else torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)


def unpack_gathered(out, B, k, slots=1):
    """Views / copies of an exchanged buffer as ([B, G*slots*k] scores, [B, G*slots*k] ids); columns are rank-major, then
    slot-major: rank r's shard slot s occupies columns (r*slots + s)*k .. +k.  This is the layout statement the tests check
    the in-place device merge against; the product path does not call it on the GPU (it costs four copy kernels)."""
    world = out.shape[0]
    nD, ioff, _ = packed_layout(B, k, slots)
    if world == 1 and slots == 1:
        return out[0, :nD].view(torch.float32).view(B, k), out[0, ioff:].view(torch.int64).view(B, k)
    Dg = out[:, :nD].contiguous().view(torch.float32).view(world * slots, B, k).permute(1, 0, 2).reshape(B, world * slots * k)
    Ig = out[:, ioff:].contiguous().view(torch.int64).view(world * slots, B, k).permute(1, 0, 2).reshape(B, world * slots * k)
    return Dg, Ig


def gather_packed(buf, B, k, group=None, slots=1):
    """exchange_packed + unpack_gathered (see there)."""
    return unpack_gathered(exchange_packed(buf, group), B, k, slots)


def merge_gathered(out, B, k_in, slots, k, descending=True):
    """Cross-source merge (rerank.py:3-9 over the concatenation of http_server.py:280-286) of an exchanged buffer, read in
    place by `rr_merge_topk_gathered`: no repacking between the collective and the merge.  Fresh (D f32 [B,k], I i64 [B,k])."""
    from ._lib import check, lib
    world = out.shape[0]
    _, ioff, per_rank = packed_layout(B, k_in, slots)
    if world * slots * k_in > 8192:   # beyond one LDS sort: fall back to the staged merge of the unpacked view
        from .rerank import merge_topk
        Dg, Ig = unpack_gathered(out, B, k_in, slots)
        return merge_topk(Dg, Ig, k, descending)
    Do = torch.empty((B, k), dtype=torch.float32, device=out.device)
    Io = torch.empty((B, k), dtype=torch.int64, device=out.device)
    check(lib().rr_merge_topk_gathered(out.data_ptr(), world, per_rank, ioff, slots, B, k_in, k, int(bool(descending)),
                                       Do.data_ptr(), Io.data_ptr(), torch.cuda.current_stream().cuda_stream), "rr_merge_topk_gathered")
    return Do, Io


def max_over_ranks(value, device, group=None):
    """max of a small integer over the ranks (used once, to size the packed buffer to the largest local shard count)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(value)
    on_dev = dist.get_backend(group) != "gloo"
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


class ShardedFlatSearch:
    """This rank's shards (FlatIndex objects) + the exchange/merge step, for queries already in the scan format
    (all shards of one width).  RetrievalPipeline is the general form (router, per-shard encoders)."""

    def __init__(self, shards, shard_ids, group=None, slots=None):
        if len(shards) != len(shard_ids):
            raise ValueError("one shard id per shard")
        self.shards = list(shards)
        self.shard_ids = [int(s) for s in shard_ids]
        self.group = group
        self.slots = slots
        self._packed = {}
        self._gathered = None

    def local_candidates(self, xq_half, k, route_mask=None):
        """Scan every local shard into this rank's packed buffer; returns (buf, D [slots,B,k], I [slots,B,k]) with global ids.
        route_mask: optional bool [B, n_total_shards] (router output)."""
        B = xq_half.shape[0]
        if self.slots is None:
            self.slots = max(1, max_over_ranks(len(self.shards), xq_half.device, self.group))
        if (B, k) not in self._packed:
            buf, D, I = alloc_packed(B, k, xq_half.device, self.slots)
            self._packed = {(B, k): (buf, D[None], I[None]) if self.slots == 1 else (buf, D, I)}
            self._gathered = alloc_gathered(buf, self.group)
        buf, D, I = self._packed[(B, k)]
        for slot, (idx, sid) in enumerate(zip(self.shards, self.shard_ids)):
            idx.search_prepared(xq_half, k, id_offset=sid << SHARD_SHIFT, out=(D[slot], I[slot]),
                                route_mask=None if route_mask is None else route_mask[:, sid])
        return buf, D, I

    def search(self, xq_half, k, route_mask=None):
        """Full federated step on device: local scans -> ONE all_gather -> merge.  Every rank gets the result."""
        buf, _, _ = self.local_candidates(xq_half, k, route_mask)
        out = exchange_packed(buf, self.group, self._gathered)
        # always through the merge: it writes fresh tensors (the packed buffer is reused by the next search, so returning
        # views of it would let search N+1 overwrite what the caller still holds of search N)
        return merge_gathered(out, xq_half.shape[0], k, self.slots, k, True)
