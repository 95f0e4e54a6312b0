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


def alloc_packed(B, k, device):
    """One allocation holding D f32[B,k] followed by I i64[B,k]: the kernels write straight into the two views and the
    exchange moves the whole buffer with ONE collective."""
    nD = B * k * 4
    pad = (-nD) % 8
    buf = torch.empty(nD + pad + B * k * 8, dtype=torch.uint8, device=device)
    D = buf[:nD].view(torch.float32).view(B, k)
    I = buf[nD + pad:].view(torch.int64).view(B, k)
    return buf, D, I


def gather_packed(buf, B, k, group=None):
    """all_gather of packed candidate buffers (see alloc_packed) -> ([B, G*k] scores, [B, G*k] ids), rank-major columns."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nD = B * k * 4
    pad = (-nD) % 8
    if world == 1:
        return buf[:nD].view(torch.float32).view(B, k), buf[nD + pad:].view(torch.int64).view(B, k)
    if buf.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on one box: stage through the host
        out = torch.empty(world * buf.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, buf.cpu(), group=group)
        out = out.to(buf.device)
    else:
        out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
        dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(world, buf.numel())
    Dg = out[:, :nD].contiguous().view(torch.float32).view(world, B, k).permute(1, 0, 2).reshape(B, world * k)
    Ig = out[:, nD + pad:].contiguous().view(torch.int64).view(world, B, k).permute(1, 0, 2).reshape(B, world * k)
    return Dg, Ig


class ShardedFlatSearch:
    """This rank's shards (FlatIndex objects) + the exchange/merge step."""

    def __init__(self, shards, shard_ids, group=None):
        if len(shards) != len(shard_ids):
            raise ValueError("one shard id per shard")
        self.shards = list(shards)
        self.shard_ids = [int(s) for s in shard_ids]
        self.group = group

    def local_candidates(self, xq_half, k, route_mask=None):
        """Scan every local shard; returns [B, S_local*k] candidates with global ids.
        route_mask: optional bool [B, n_total_shards] (router output)."""
        Ds, Is = [], []
        for idx, sid in zip(self.shards, self.shard_ids):
            D, I = idx.search_prepared(xq_half, k, id_offset=sid << SHARD_SHIFT,
                                       route_mask=None if route_mask is None else route_mask[:, sid])
            Ds.append(D)
            Is.append(I)
        return (Ds[0], Is[0]) if len(Ds) == 1 else (torch.cat(Ds, 1), torch.cat(Is, 1))

    def search(self, xq_half, k, route_mask=None):
        """Full federated step on device: local scans -> all_gather -> merge.  Every rank gets the result."""
        from .rerank import merge_topk
        D, I = self.local_candidates(xq_half, k, route_mask)
        Dg, Ig = gather_candidates(D, I, self.group)
        if Dg.shape[1] == k and route_mask is None:
            return Dg, Ig
        return merge_topk(Dg, Ig, k, True)
