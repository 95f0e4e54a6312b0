"""GPU parity of the segmented search (`rr_flat_search_segments`, SegmentedIndex): ONE pass over several data sources that
receive the same query embeddings, against the oracle CHAIN the reference's flow defines — every selected source answers with
its own top-k (`index.search`, data_source.py:158, 186, 203), the front-end concatenates the replies and keeps the k best
(http_server.py:280-293, rerank.py:3-9).  Integer-valued embeddings make every score exact, so ids and scores must match bit
for bit, ties included (ties: ascending global id = (source order, row))."""
import numpy as np
import pytest
import torch

from tests.util import half_round, int_data

pytestmark = pytest.mark.gpu
SHIFT = 40


def oracle_chain(parts, xq, k, mask, sids=None):
    """per-source oracle top-k -> concat -> oracle merge; mask bool [nq, C] or None; sids: global source id (mask column) per part"""
    from oracle import oracle as O
    nq = xq.shape[0]
    sids = list(range(len(parts))) if sids is None else sids
    Ds, Is = [], []
    for p, sid in zip(parts, sids):
        if p.shape[0] == 0:
            D, I = np.full((nq, k), -np.inf, np.float32), np.full((nq, k), -1, np.int64)
        else:
            D, I = O.flat_search_ip(p, xq, k)
            I = np.where(I >= 0, I + (sid << SHIFT), -1)
        if mask is not None:
            keep = mask[:, sid][:, None]
            D, I = np.where(keep, D, -np.inf).astype(np.float32), np.where(keep, I, -1)
        Ds.append(D)
        Is.append(I)
    return O.merge_topk(np.concatenate(Ds, 1), np.concatenate(Is, 1), k, True)


def build(parts, d, gpu, dtype="fp16", metric="ip", sids=None, poison=True):
    from ragroute_amd.flat_index import SegmentedIndex
    sids = list(range(len(parts))) if sids is None else sids
    seg = SegmentedIndex(d, [p.shape[0] for p in parts], id_offsets=[s << SHIFT for s in sids], mask_cols=sids, metric=metric,
                         dtype=dtype, device=gpu)
    if poison and seg._xb.shape[0] > 0:
        seg._xb.fill_(float("nan"))          # alignment gaps hold garbage: scanned, never returned, never counted
        seg._xb[::7] = 60000.0
    for s, p in enumerate(parts):
        if p.shape[0]:
            seg.fill(s, p)
    return seg


CASES = [
    # (d, rows per source, nq, k)
    (768, [40_000, 9_000, 0, 70_001], 16, 32),          # medrag-shaped: scan16 kernel, an empty source, ragged ends
    (768, [300, 5, 4100, 77], 256, 32),                  # under 8192 rows in total: the dense path
    (768, [9_000, 31, 257, 120_000, 1], 256, 10),        # a one-row source; 256 queries
    (1024, [30_000, 3_633, 8_674, 25_000], 100, 10),     # feb4rag's UAE-Large-V1 group, <= 128 queries: half-resident kernel
    (1024, [30_000, 3_633, 8_674, 25_000], 256, 10),     # ... 256 queries: row-split wide-row kernel (flat_scan_wide_rs_kernel)
    (1024, [30_000, 3_633, 8_674, 25_000], 200, 10),     # ... 200 queries: 8-wave wide-row kernel
    (4096, [9_000, 2_000, 14_000], 256, 10),             # wide rows: row-split kernel, K rotation per 256-row group
    (4096, [9_000, 2_000, 14_000], 180, 10),             # ... 180 queries: 4-wave kernel (flat_scan_wide_pd_kernel)
    (2048, [9_000, 2_000, 14_000], 256, 300),            # row-split kernel with k beyond one candidate-buffer generation (round 4: any k)
    (128, [50_000, 50_000], 37, 100),
    (768, [20_000, 5_000, 9_000], 8, 300),               # k beyond one candidate-buffer generation
    (256, [700 + 13 * i for i in range(32)], 40, 10),   # RR_MAX_SEGMENTS sources
]


@pytest.mark.parametrize("d,rows,nq,k", CASES)
def test_segmented_search_equals_the_oracle_chain(gpu, d, rows, nq, k):
    rng = np.random.default_rng(sum(rows) + d + nq)
    parts = [int_data(rng, n, d) for n in rows]
    xq = int_data(rng, nq, d)
    mask = rng.integers(0, 2, size=(nq, len(rows))).astype(bool)
    mask[0] = False                    # a query routed nowhere: all padding
    if nq > 1:
        mask[1] = True
    seg = build(parts, d, gpu)
    for m in (mask, None):
        D, I = seg.search(xq, k, route_mask=m)
        Dr, Ir = oracle_chain(parts, xq, k, m)
        assert np.array_equal(I, Ir), (np.argwhere(I != Ir)[:5], I[I != Ir][:5], Ir[I != Ir][:5])
        assert np.array_equal(D, Dr)
    assert (I[1] >= 0).sum() == min(k, sum(rows))


def test_segmented_search_mask_columns_and_id_offsets_are_the_callers(gpu):
    """Sources 2, 5 and 11 of a 13-source federation (FeB4RAG's column layout): mask columns = global source ids, ids = sid << 40."""
    rng = np.random.default_rng(3)
    sids = [2, 5, 11]
    parts = [int_data(rng, n, 768) for n in (20_000, 700, 33_000)]
    xq = int_data(rng, 64, 768)
    mask = rng.integers(0, 2, size=(64, 13)).astype(bool)
    seg = build(parts, 768, gpu, sids=sids)
    D, I = seg.search(xq, 10, route_mask=mask)
    Dr, Ir = oracle_chain(parts, xq, 10, mask, sids)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


def test_segmented_search_all_ties(gpu):
    """Every row scores the same: the k lowest global ids of the selected sources, whatever the scan order."""
    parts = [np.ones((5000, 256), np.float32), np.ones((300, 256), np.float32), np.ones((12_000, 256), np.float32)]
    xq = np.ones((20, 256), np.float32)
    mask = np.ones((20, 3), bool)
    mask[3, 0] = False
    mask[4, :2] = False
    seg = build(parts, 256, gpu)
    D, I = seg.search(xq, 32, route_mask=mask)
    Dr, Ir = oracle_chain(parts, xq, 32, mask)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    assert I[3, 0] == (1 << SHIFT) and I[4, 0] == (2 << SHIFT) and I[0, 0] == 0


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_segmented_search_gaussian_cosine(gpu, dtype):
    """Real-valued rows (cosine): scores within 1e-3 of the oracle chain, ids identical where the ranking is unambiguous."""
    from tests.util import assert_topk_close
    rng = np.random.default_rng(8)
    rows, d, nq, k = [60_000, 2_000, 45_000], 768, 48, 32
    raw = [rng.standard_normal((n, d)).astype(np.float32) for n in rows]
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    norm = lambda x: x / np.linalg.norm(x, axis=1, keepdims=True)  # noqa: E731
    parts = [half_round(norm(p), dtype) for p in raw]
    mask = rng.integers(0, 2, size=(nq, 3)).astype(bool)
    seg = build(raw, d, gpu, dtype=dtype, metric="cosine")
    D, I = seg.search(xq, k, route_mask=mask)
    Dr, Ir = oracle_chain(parts, half_round(norm(xq), dtype), k, mask)
    assert_topk_close(D, I, Dr, Ir, tol=1e-3)


def test_segment_slices_serve_the_per_source_surface_on_the_same_memory(gpu):
    """SegmentedIndex.source(s) is a FlatIndex on the slice (no copy): the per-source `index.search` the DataSource mirror calls
    (data_source.py:158, 186, 203) and the one-pass search agree through the chain."""
    from oracle import oracle as O
    from ragroute_amd.rerank import merge_topk
    rng = np.random.default_rng(12)
    parts = [int_data(rng, n, 768) for n in (15_000, 333, 21_000)]
    xq = int_data(rng, 32, 768)
    seg = build(parts, 768, gpu)
    Ds, Is = [], []
    for s, p in enumerate(parts):
        idx = seg.source(s)
        assert idx.xb.data_ptr() == seg.rows_of(s).data_ptr()
        D, I = idx.search_prepared(idx.prepare_queries(xq), 32, id_offset=s << SHIFT)
        Dr, Ir = O.flat_search_ip(p, xq, 32)
        assert np.array_equal(I.cpu().numpy(), np.where(Ir >= 0, Ir + (s << SHIFT), -1)) and np.array_equal(D.cpu().numpy(), Dr)
        Ds.append(D)
        Is.append(I)
    Dm, Im = merge_topk(torch.cat(Ds, 1), torch.cat(Is, 1), 32, True)
    D1, I1 = seg.search_prepared(seg.prepare_queries(xq), 32)
    assert torch.equal(I1, Im) and torch.equal(D1, Dm)


def test_pipeline_with_shared_encoders_equals_the_per_source_pipeline(gpu):
    """RetrievalPipeline(share_queries=encoder per source) packs the sources of one encoder into one segmented search; the result
    equals the pipeline that searches every source separately (and the oracle chain), with a router mask in between."""
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.pipeline import RetrievalPipeline
    rng = np.random.default_rng(21)
    enc = ["uae", "mpnet", "uae", "e5", "mpnet", "uae"]            # feb4rag-like: three sources share one encoder, two another
    width = {"uae": 1024, "mpnet": 768, "e5": 1024}
    rows = [12_000, 5_000, 3_633, 9_000, 700, 20_000]
    parts = [int_data(rng, n, width[e]) for n, e in zip(rows, enc)]
    nq, k = 40, 10
    emb = {e: int_data(rng, nq, w) for e, w in width.items()}
    mask = torch.from_numpy(rng.integers(0, 2, size=(nq, 6)).astype(bool)).to(gpu)

    class FixedRouter:
        def run(self, xq_models):
            return None, mask

    def make():
        out = []
        for p, e in zip(parts, enc):
            idx = FlatIndex(width[e], device=gpu)
            idx.add(p)
            out.append(idx)
        return out

    xq = {s: torch.from_numpy(emb[e]).to(gpu) for s, e in enumerate(enc)}
    plain = RetrievalPipeline(make(), list(range(6)), router=FixedRouter())
    packed = RetrievalPipeline(make(), list(range(6)), router=FixedRouter(), share_queries=enc)
    assert [u[0] for u in packed.units].count("segments") == 2 and len(packed.units) == 3
    D0, I0 = plain.search(xq, k, xq_models=torch.zeros(nq, 1, 1, device=gpu))
    D1, I1 = packed.search(xq, k, xq_models=torch.zeros(nq, 1, 1, device=gpu))
    assert packed.slots == 3 and plain.slots == 6
    assert torch.equal(I0, I1) and torch.equal(D0, D1)
    # and the chain itself, source by source with its own embedding
    from oracle import oracle as O
    Ds, Is = [], []
    for s, (p, e) in enumerate(zip(parts, enc)):
        D, I = O.flat_search_ip(p, emb[e], k)
        keep = mask[:, s].cpu().numpy()[:, None]
        Ds.append(np.where(keep, D, -np.inf).astype(np.float32))
        Is.append(np.where(keep & (I >= 0), I + (s << SHIFT), -1))
    Dr, Ir = O.merge_topk(np.concatenate(Ds, 1), np.concatenate(Is, 1), k, True)
    assert np.array_equal(I1.cpu().numpy(), Ir) and np.array_equal(D1.cpu().numpy(), Dr)


def test_config2_exact_size_one_million_rows_bit_exact(gpu):
    """BASELINE config 2 at its own size: 1 000 000 x 768 fp16, 256 queries, k = 32, exact arithmetic (integer-valued rows), ids and
    scores bit for bit against the oracle.  (~20 s of oracle time on the box's host cores.)"""
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    rng = np.random.default_rng(2)
    xb = rng.integers(-2, 3, size=(1_000_000, 768), dtype=np.int8)
    xq = int_data(rng, 256, 768)
    idx = FlatIndex(768, device=gpu)
    for s in range(0, xb.shape[0], 250_000):
        idx.add(xb[s:s + 250_000].astype(np.float32))
    D, I = idx.search(xq, 32)
    Dr, Ir = O.flat_search_ip(xb.astype(np.float32), xq, 32)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


def test_torch_ops_reach_the_segmented_search_and_the_in_place_merge(gpu):
    """torch.ops.ragroute.flat_topk_segments / merge_gathered are the same entry points as SegmentedIndex / sharded.merge_gathered."""
    import ragroute_amd.torch_ops  # noqa: F401  (registers the ops)
    from ragroute_amd import sharded as S
    rng = np.random.default_rng(31)
    parts = [int_data(rng, n, 768) for n in (9_000, 100, 30_000)]
    xq = int_data(rng, 24, 768)
    mask = rng.integers(0, 2, size=(24, 3)).astype(bool)
    seg = build(parts, 768, gpu)
    q = seg.prepare_queries(xq)
    m = torch.from_numpy(mask).to(gpu)
    D, I = torch.ops.ragroute.flat_topk_segments(seg._xb, seg.begins, seg.rows, seg.id_offsets, seg.mask_cols, q, 10, m)
    Dr, Ir = oracle_chain(parts, xq, 10, mask)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
    buf, Dp, Ip = S.alloc_packed(24, 10, gpu)
    Dp.copy_(D)
    Ip.copy_(I)
    out = torch.stack([buf, buf])
    D2, I2 = torch.ops.ragroute.merge_gathered(out, 1, 24, 10, 10)
    from oracle import oracle as O
    Dw, Iw = O.merge_topk(np.concatenate([Dr, Dr], 1), np.concatenate([Ir, Ir], 1), 10, True)
    assert np.array_equal(I2.cpu().numpy(), Iw) and np.array_equal(D2.cpu().numpy(), Dw)


def test_workspace_size_by_width(gpu):
    """Rows up to 768 wide need 4 candidate buffers per workgroup and query, wider rows 8 (the row-split kernel; since round 4 at
    every k): rr_flat_search_workspace_bytes_for sizes for one width, rr_flat_search_workspace_bytes for any."""
    from ragroute_amd import _lib
    L = _lib.lib()
    a, b, c = L.rr_flat_search_workspace_bytes_for(32, 768), L.rr_flat_search_workspace_bytes_for(32, 1024), L.rr_flat_search_workspace_bytes(32)
    assert 0 < a < b == c
    assert L.rr_flat_search_workspace_bytes_for(32, 1000) == 0
    assert L.rr_flat_search_workspace_bytes_for(300, 768) < L.rr_flat_search_workspace_bytes_for(300, 1024) == L.rr_flat_search_workspace_bytes(300)
