"""Synthetic federations at the reference's real shapes (measurement infrastructure for bench.py --workload and
tools/config34.py; not part of the product path).

  federation("medrag" | "feb4rag")   ragroute_amd.placement.federation: sources / order / encoders from config.py:32-71,
                                     public row counts, encoder widths.
  Row r of source s is a pure function of (s, r): i.i.d. N(0,1), L2-normalised, cast to fp16 — generated on device in blocks
  of 2^18 source rows seeded by (s, block), so ANY placement of the federation (whole sources, row slices, one GPU or eight)
  holds the same corpus and must return the same ids.
  Queries: one embedding per encoder (seed 4321), normalised; the router input packs them [B, n_models, d_max] as
  Router.pack_queries does (router.py:241-265)."""
import numpy as np
import torch

from ragroute_amd import config as C

BLOCK = 1 << 18


def fill_half(src, sl, out):
    """RetrievalPipeline.from_placement's `fill_half`: rows [sl.row_begin, +sl.n_rows) of source src into `out` (storage dtype)."""
    dev = out.device
    g = torch.Generator(device=dev)
    r0, r1 = sl.row_begin, sl.row_begin + sl.n_rows
    for b in range(r0 // BLOCK, -(-r1 // BLOCK) if r1 > r0 else r0 // BLOCK):
        g.manual_seed(1234 + (src.sid << 20) + b)
        lo, hi = b * BLOCK, min((b + 1) * BLOCK, src.rows)
        x = torch.randn((hi - lo, src.dim), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        a, e = max(lo, r0), min(hi, r1)
        out[a - r0: e - r0, : src.dim] = x[a - lo: e - lo].to(out.dtype)


def query_embeddings(fed, B, dev, seed=4321):
    """{encoder: f32 [B, width]} — the same on every rank."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    emb = {}
    for enc in sorted({s.encoder for s in fed}):
        w = next(s.dim for s in fed if s.encoder == enc)
        x = torch.randn((B, w), generator=g, device=dev)
        emb[enc] = x / x.norm(dim=1, keepdim=True)
    return emb


def router_for(dataset, fed, centroids, dev):
    """FoldedRouter over the federation: CorpusRoutingNN default init seed 0 (router.py:37-55), the dataset's one-hot ids and
    threshold (config.py:72-90, router.py:276-280), centroids f32 [C, d_max]."""
    from ragroute_amd.router import CorpusRoutingNN, FoldedRouter
    models = sorted({s.encoder for s in fed})
    names = [s.name for s in fed]
    onehot = [C.FEB4RAG_SOURCE_TO_ID[n] for n in names] if dataset == "feb4rag" else [C.MEDRAG_SOURCE_TO_ID[n] for n in names]
    net = CorpusRoutingNN(C.ROUTER_INPUT_DIMENSION[dataset], seed=0)
    return FoldedRouter.fold(net.state_dict(), np.asarray(centroids, np.float32), onehot, len(fed), C.EMBEDDING_MAX_LENGTH[dataset],
                             [models.index(s.encoder) for s in fed], C.ROUTER_THRESHOLD[dataset], device=dev)


def pack_router_input(dataset, fed, emb, dev):
    models = sorted({s.encoder for s in fed})
    B = next(iter(emb.values())).shape[0]
    xq_models = torch.zeros((B, len(models), C.EMBEDDING_MAX_LENGTH[dataset]), device=dev)
    for j, m in enumerate(models):
        xq_models[:, j, : emb[m].shape[1]] = emb[m]
    return xq_models


def local_centroids(dataset, fed, pipe, dev, sample_rows=100_000):
    """f32 [C, d_max]: for every source whose FIRST slice lives on this rank, the mean of (up to) its first 100k rows
    (SURVEY §8d); zero rows for the others — the ranks' matrices are summed once at set-up."""
    from ragroute_amd.flat_index import SegmentedIndex
    d_max = C.EMBEDDING_MAX_LENGTH[dataset]
    cen = torch.zeros((len(fed), d_max), dtype=torch.float32, device=dev)
    pos = {s.sid: i for i, s in enumerate(fed)}
    for unit, (_, obj, _, _) in zip(pipe.placement.ranks[pipe.rank], pipe.units):
        for j, sl in enumerate(unit.slices):
            if sl.row_begin != 0 or sl.n_rows == 0:
                continue
            rows = obj.rows_of(j) if isinstance(obj, SegmentedIndex) else obj.xb
            w = pipe.placement.sources[sl.sid].dim
            cen[pos[sl.sid], :w] = rows[: min(sl.n_rows, sample_rows), :w].float().mean(0)
    return cen


def queries_by_source(fed, emb):
    return {s.sid: emb[s.encoder] for s in fed}
