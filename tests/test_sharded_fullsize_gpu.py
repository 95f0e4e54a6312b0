"""GPU: the federated step on one device (several shards, fake world of 1) and size-independent
properties at BASELINE.json's full shape (10M x 768 fp16, B=256, k=32)."""
import numpy as np
import pytest
import torch

from tests.util import int_data

pytestmark = pytest.mark.gpu


def test_sharded_equals_unsharded(gpu):
    """Merging per-shard top-k (global ids) == top-k of the concatenated corpus, bit for bit on exact data."""
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.sharded import SHARD_SHIFT, ShardedFlatSearch
    rng = np.random.default_rng(8)
    sizes = [30_000, 9_000, 70_000, 500]
    parts = [int_data(rng, n, 768) for n in sizes]
    xq = int_data(rng, 40, 768)
    shards = []
    for p in parts:
        idx = FlatIndex(768, device=gpu)
        idx.add(p)
        shards.append(idx)
    fed = ShardedFlatSearch(shards, list(range(len(parts))))
    xqh = shards[0].prepare_queries(xq)
    D, I = fed.search(xqh, 32)
    Dr, Ir = O.flat_search_ip(np.concatenate(parts), xq, 32)
    base = np.cumsum([0] + sizes)
    gids = np.concatenate([np.arange(n) + (s << SHARD_SHIFT) for s, n in enumerate(sizes)])
    assert np.array_equal(D.cpu().numpy(), Dr)
    assert np.array_equal(I.cpu().numpy(), gids[Ir])
    # routing mask: a masked (query, shard) pair contributes nothing
    mask = torch.ones((40, 4), dtype=torch.bool, device=gpu)
    mask[:, 2] = False
    Dm, Im = fed.search(xqh, 32, route_mask=mask)
    keep = np.concatenate([parts[0], parts[1], parts[3]])
    gk = np.concatenate([gids[base[0]:base[1]], gids[base[1]:base[2]], gids[base[3]:base[4]]])
    Dk, Ik = O.flat_search_ip(keep, xq, 32)
    assert np.array_equal(Dm.cpu().numpy(), Dk) and np.array_equal(Im.cpu().numpy(), gk[Ik])


def test_full_size_planted_neighbours(gpu):
    """10M x 768 fp16, B=256, k=32 (the headline shape).  The oracle cannot scan 10M rows in seconds, so the
    check is a planted-answer property: for every query, 32 rows at known random positions are set to
    (1 + j/64) * query, which makes them the exact top-32 in a known order; everything else is N(0,1)/sqrt(d)."""
    from ragroute_amd.flat_index import FlatIndex
    n, d, nq, k = 10_000_000, 768, 256, 32
    g = torch.Generator(device=gpu)
    g.manual_seed(1234)
    xb = torch.empty((n, d), dtype=torch.float16, device=gpu)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        xb[s:e] = (torch.randn((e - s, d), generator=g, device=gpu) / d ** 0.5).to(torch.float16)
    xq = torch.randn((nq, d), generator=g, device=gpu)
    xq = (xq / xq.norm(dim=1, keepdim=True)).to(torch.float16)
    pos = torch.randperm(n, generator=g, device=gpu)[: nq * k].reshape(nq, k)
    scale = (1 + torch.arange(k, device=gpu, dtype=torch.float32) / 64)        # exact in fp16 products' ordering
    planted = (xq.float()[:, None, :] * scale[None, :, None]).to(torch.float16)  # [nq,k,d]
    xb[pos.reshape(-1)] = planted.reshape(-1, d)
    idx = FlatIndex(d, device=gpu)
    idx.adopt(xb)
    D, I = idx.search_prepared(xq, k)
    want_scores = (planted.float() * xq.float()[:, None, :]).sum(-1)            # [nq,k]
    order = torch.argsort(want_scores, dim=1, descending=True, stable=True)
    assert torch.equal(I, torch.gather(pos, 1, order))
    assert torch.allclose(D, torch.gather(want_scores, 1, order), atol=1e-3, rtol=0)
    # idempotence / determinism: a second run returns bit-identical results
    D2, I2 = idx.search_prepared(xq, k)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    # sub-batch property: the first 7 queries alone give the same rows
    D3, I3 = idx.search_prepared(xq[:7].contiguous(), k)
    assert torch.equal(I3, I[:7]) and torch.equal(D3, D[:7])


@pytest.mark.parametrize("dataset", ["feb4rag", "medrag"])
def test_sliced_placement_is_placement_independent_at_scale(gpu, dataset):
    """BASELINE configs 4 / 3 at 1/8 of the federations' real row counts (same widths, encoders and source order; synthetic rows
    that are a pure function of (source, row), tools/workloads.py): the 8-GPU placement of placement.plan, every rank's units built
    and searched in turn on this device and merged, against the one-GPU search of the same federation - the row slices, the
    per-rank segmented units and the 40 exchange candidates per query must not change one id or score.  (The oracle cannot scan
    millions of rows in seconds: this is the size-independent property; small cases against the oracle: test_placement_gpu.py.)"""
    from ragroute_amd import config as C
    from ragroute_amd import placement as P
    from ragroute_amd.pipeline import RetrievalPipeline
    from ragroute_amd.rerank import merge_topk
    from tools import workloads as W
    fed = P.federation(dataset, rows={name: max(64, n // 8) for name, n in P.ROWS[dataset].items()})
    k, B = C.K[dataset], 256
    emb = W.query_embeddings(fed, B, gpu)
    xq = W.queries_by_source(fed, emb)
    xq_models = W.pack_router_input(dataset, fed, emb, gpu)
    g = torch.Generator(device=gpu)
    g.manual_seed(5)
    mask = torch.rand((B, len(fed)), generator=g, device=gpu) < 0.7        # a fixed random route mask, some queries routed nowhere
    mask[:4] = False

    class Fixed:
        def run(self, _):
            return None, mask

    def run(plan):
        out = []
        for r in range(len(plan.ranks)):
            pipe = RetrievalPipeline.from_placement(plan, r, fill_half=W.fill_half, router=Fixed(), device=gpu)
            D, I = pipe.search(xq, k, xq_models=xq_models)
            out.append((D.clone(), I.clone()))
            del pipe
            torch.cuda.empty_cache()
        return out

    D1, I1 = run(P.plan(fed, 1))[0]
    plan8 = P.plan(fed, 8, cost=P.CostModel(fixed_ms=0.01, segment_ms=0.00125))   # fixed costs scaled with the rows: the full-size plan's cuts
    assert sum(len(plan8.slices_of(s.sid)) - 1 for s in fed) >= 4
    parts = run(plan8)
    Dm, Im = merge_topk(torch.cat([d for d, _ in parts], 1), torch.cat([i for _, i in parts], 1), k, True)
    assert torch.equal(Im, I1) and torch.equal(Dm, D1)
    assert bool((I1[:4] == -1).all()) and bool((I1[4:, 0] >= 0).any())
    # ... and the whole-source layout (source s -> GPU s mod 8) gives the same answer
    whole = run(P.whole_source_plan(fed, 8))
    Dw, Iw = merge_topk(torch.cat([d for d, _ in whole], 1), torch.cat([i for _, i in whole], 1), k, True)
    assert torch.equal(Iw, I1) and torch.equal(Dw, D1)
