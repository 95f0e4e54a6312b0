"""CPU: the placement planner (ragroute_amd/placement.py) — row slices of the federation's sources balanced over G GPUs —
and, over gloo with two ranks, the slice arithmetic end to end: a source cut in three slices over two ranks, route mask,
ties, against the reference's flow on the WHOLE sources (one top-k per selected source, data_source.py:158-163; concatenate
and keep the k best, http_server.py:280-293 + rerank.py:3-9).  The per-slice scan is stood in for by the oracle (the HIP
scan needs a GPU; tests/test_placement_gpu.py runs the same check through the C ABI)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ragroute_amd import placement as P


def _covered(pl):
    """Every source's slices tile [0, rows) exactly once, cuts on 256-row boundaries."""
    for sid, src in pl.sources.items():
        sl = pl.slices_of(sid)
        assert sl, f"source {sid} is placed nowhere"
        pos = 0
        for s in sl:
            assert s.row_begin == pos and s.n_rows >= 0
            assert s.row_begin % P.SLICE_ALIGN == 0
            pos += s.n_rows
        assert pos == src.rows


def test_padded_dim_matches_the_library():
    from ragroute_amd._lib import lib
    for d in list(range(1, 1700, 37)) + [768, 769, 896, 1024, 1025, 1536, 1537, 2048, 4096, 8192]:
        assert P.padded_dim(d) == lib().rr_padded_dim(d), d


@pytest.mark.parametrize("dataset", ["feb4rag", "medrag"])
@pytest.mark.parametrize("G", [1, 2, 3, 4, 8])
def test_real_federations_balance(dataset, G):
    fed = P.federation(dataset)
    pl = P.plan(fed, G)
    _covered(pl)
    assert len(pl.ranks) == G
    assert pl.imbalance <= 1.02, pl.predicted_ms
    whole = P.whole_source_plan(fed, G)
    assert max(pl.predicted_ms) <= max(whole.predicted_ms) + 1e-9
    if G == 8:   # the point of slicing: s mod G predicts 3.75x for FeB4RAG (msmarco alone is a quarter of the bytes)
        one = max(P.plan(fed, 1).predicted_ms)
        assert one / max(pl.predicted_ms) >= 6.5
        assert one / max(whole.predicted_ms) < 4.5


def test_same_encoder_pieces_share_a_unit():
    pl = P.plan(P.federation("feb4rag"), 8)
    for units in pl.ranks:
        groups = [u.group for u in units]
        assert len(groups) == len(set(groups)), "two units of one encoder group on one rank"
        for u in units:
            assert len({pl.sources[s.sid].encoder for s in u.slices}) == 1
            offs = [s.id_offset for s in u.slices]
            assert offs == sorted(offs)       # ascending id offsets: the segmented search's tie order = ascending id
    # at most G - 1 cuts
    assert sum(len(pl.slices_of(sid)) - 1 for sid in pl.sources) <= 7


def test_plan_is_deterministic_and_rank_independent():
    a = P.plan(P.federation("feb4rag"), 8).describe()
    b = P.plan(list(reversed(P.federation("feb4rag"))), 8).describe()
    assert a == b


def test_small_and_degenerate_federations():
    # fewer sources than GPUs, a zero-row source, an L2 source (never grouped), tiny sources (never cut)
    fed = [P.Source(0, 1000, 64, "a"), P.Source(1, 0, 64, "a"), P.Source(2, 3_000_000, 64, "b", "l2"), P.Source(3, 500, 64, "b", "l2")]
    pl = P.plan(fed, 4)
    _covered(pl)
    for units in pl.ranks:
        for u in units:
            if u.group[2] == "l2":
                assert len(u.slices) == 1
    assert len(pl.slices_of(2)) >= 3            # the one large source is what gets cut
    assert len(pl.slices_of(0)) == 1
    one = P.plan([P.Source(7, 10, 32)], 8)      # one tiny source on 8 GPUs: seven idle ranks
    _covered(one)
    assert sum(len(u) for u in one.ranks) == 1 and one.slots == 1
    with pytest.raises(ValueError):
        P.plan([P.Source(0, 10, 32), P.Source(0, 10, 32)], 2)
    with pytest.raises(ValueError):
        P.plan([P.Source(0, 1 << 40, 32)], 2)


def test_min_slice_rows_is_respected():
    pl = P.plan([P.Source(0, 300_000, 768, "a"), P.Source(1, 290_000, 768, "a"), P.Source(2, 310_000, 768, "a")], 4)
    _covered(pl)
    for sid in pl.sources:
        for s in pl.slices_of(sid):
            assert s.n_rows >= P.MIN_SLICE_ROWS


# ---- world 2 over gloo: slices as search units ---------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleSlice:
    """FlatIndex / SegmentedIndex stand-in over row slices, backed by the oracle (test infrastructure)."""

    def __init__(self, pieces, segmented):
        self.pieces, self.segmented = pieces, segmented      # [(rows f32, id_offset, mask_col)]

    def prepare_queries(self, xq):
        return xq

    def search_prepared(self, xq, k, id_offset=0, out=None, route_mask=None):
        from oracle import oracle as O
        xq = xq.numpy()
        if not self.segmented:
            (xb, _, _), = self.pieces
            D, I = O.flat_search_ip(xb, xq, k)
            I = np.where(I >= 0, I + id_offset, -1)
            if route_mask is not None:
                keep = route_mask.numpy().astype(bool)[:, None]
                D, I = np.where(keep, D, -np.inf).astype(np.float32), np.where(keep, I, -1)
        else:   # rr_flat_search_segments' contract: the merged top-k over the selected segments, ties by ascending id
            mask = None if route_mask is None else route_mask.numpy().astype(bool)
            Ds, Is = [], []
            for xb, off, col in self.pieces:
                d_, i_ = O.flat_search_ip(xb, xq, k)
                i_ = np.where(i_ >= 0, i_ + off, -1)
                if mask is not None:
                    keep = mask[:, col][:, None]
                    d_, i_ = np.where(keep, d_, -np.inf).astype(np.float32), np.where(keep, i_, -1)
                Ds.append(d_)
                Is.append(i_)
            D, I = O.merge_topk(np.concatenate(Ds, 1), np.concatenate(Is, 1), k, True)
        out[0].copy_(torch.from_numpy(np.ascontiguousarray(D)))
        out[1].copy_(torch.from_numpy(np.ascontiguousarray(I)))
        return out


class _FixedRouter:
    def __init__(self, mask):
        self.mask = mask

    def run(self, xq_models):
        return None, self.mask


def slice_case(seed=3, big_rows=1500):
    """Three sources (two share an encoder), the large one cut by hand in THREE slices over two ranks."""
    from tests.util import int_data
    rng = np.random.default_rng(seed)
    fed = [P.Source(0, big_rows, 64, "enc-a", name="big"), P.Source(1, 300, 64, "enc-a", name="small"), P.Source(2, 777, 128, "enc-b", name="other")]
    corpora = {s.sid: int_data(rng, s.rows, s.dim) for s in fed}
    corpora[0][700:740] = corpora[0][100:140]      # equal rows in different slices of one source: ties across the cuts
    corpora[1][5:25] = corpora[0][100:120]         # ... and across sources of one encoder
    nq, k = 9, 12
    emb = {"enc-a": int_data(rng, nq, 64), "enc-b": int_data(rng, nq, 128)}
    mask = rng.integers(0, 2, size=(nq, 3)).astype(bool)
    mask[0] = False
    mask[1] = True
    mask[2] = [True, False, False]
    by = {s.sid: s for s in fed}
    ranks = [[P.Unit(fed[0].group, (P.RowSlice(0, 0, 512), P.RowSlice(0, 1024, big_rows - 1024), P.RowSlice(1, 0, 300)))],
             [P.Unit(fed[0].group, (P.RowSlice(0, 512, 512),)), P.Unit(fed[2].group, (P.RowSlice(2, 0, 777),))]]
    return fed, corpora, emb, mask, nq, k, P.Placement(ranks, [0.0, 0.0], by)


def expected_chain(O, fed, corpora, emb, mask, nq, k):
    """The reference's flow on the WHOLE sources."""
    want_D, want_I = [], []
    for q in range(nq):
        cand = []
        for s in fed:
            if mask[q, s.sid]:
                Ds, Is = O.flat_search_ip(corpora[s.sid], emb[s.encoder][q:q + 1], k)
                cand += [(-float(d), (s.sid << P.SHARD_SHIFT) + int(i)) for d, i in zip(Ds[0], Is[0]) if i >= 0]
        cand.sort()
        want_I.append([i for _, i in cand[:k]] + [-1] * (k - len(cand[:k])))
        want_D.append([-d for d, _ in cand[:k]] + [-np.inf] * (k - len(cand[:k])))
    return want_D, want_I


def _slice_worker(rank, world, port, out_dir, planned):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from ragroute_amd import pipeline as PL
        from ragroute_amd.sharded import unpack_gathered

        def merge_gathered_cpu(out, B, k_in, slots, k, descending=True):   # the device merge needs a GPU (see test_sharded_gloo.py)
            Dg, Ig = unpack_gathered(out, B, k_in, slots)
            return tuple(torch.from_numpy(a) for a in O.merge_topk(Dg.numpy(), Ig.numpy(), k, descending))
        PL.merge_gathered = merge_gathered_cpu
        fed, corpora, emb, mask, nq, k, pl = slice_case(big_rows=3000 if planned else 1500)
        if planned:   # the planner's own cut of the same federation (min slice and fixed cost lowered to the toy sizes)
            pl = P.plan(fed, world, cost=P.CostModel(fixed_ms=1e-7, segment_ms=0.0), min_slice_rows=256)
            assert len(pl.slices_of(0)) >= 2
        units = []
        for u in pl.ranks[rank]:
            pieces = [(corpora[s.sid][s.row_begin: s.row_begin + s.n_rows], s.id_offset, s.sid) for s in u.slices]
            obj = _OracleSlice(pieces, len(pieces) > 1)
            units.append(("segments" if len(pieces) > 1 else "shard", obj, [s.sid for s in u.slices], None if len(pieces) > 1 else pieces[0][1]))
        pipe = PL.RetrievalPipeline([], [], router=_FixedRouter(torch.from_numpy(mask)), slots=pl.slots)
        pipe.units = units
        xq = {s.sid: torch.from_numpy(emb[s.encoder]) for s in fed}
        D, I = pipe.search(xq, k, xq_models=torch.zeros(nq, 1, 1))
        want_D, want_I = expected_chain(O, fed, corpora, emb, mask, nq, k)
        assert I.numpy().tolist() == want_I
        assert D.numpy().tolist() == want_D
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("planned", [False, True])
def test_one_source_in_three_slices_over_two_ranks(tmp_path, planned):
    mp.spawn(_slice_worker, args=(2, _free_port(), str(tmp_path), planned), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


# ---- world 8 over gloo: the planner's own FeB4RAG-shaped plan, every rank's units, uneven slots -------------------------------
def _feb4rag_small():
    """FeB4RAG's thirteen sources at 1/1000 of their row counts (encoders and order of config.py:32-57), widths scaled 1/16."""
    fed = []
    for s in P.federation("feb4rag"):
        fed.append(P.Source(s.sid, max(3, s.rows // 1000), s.dim // 16, s.encoder, name=s.name))
    return fed


def _eight_rank_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from oracle import oracle as O
        from ragroute_amd import pipeline as PL
        from ragroute_amd.sharded import unpack_gathered
        from tests.util import int_data
        O.set_threads(1)

        def merge_gathered_cpu(out, B, k_in, slots, k, descending=True):
            Dg, Ig = unpack_gathered(out, B, k_in, slots)
            return tuple(torch.from_numpy(a) for a in O.merge_topk(Dg.numpy(), Ig.numpy(), k, descending))
        PL.merge_gathered = merge_gathered_cpu
        fed = _feb4rag_small()
        pl = P.plan(fed, world, cost=P.CostModel(fixed_ms=1e-6, segment_ms=0.0), min_slice_rows=256)
        assert sum(len(pl.slices_of(s.sid)) - 1 for s in fed) >= 4, "the plan should cut the large sources"
        rng = np.random.default_rng(77)                      # the same stream on every rank
        corpora = {s.sid: int_data(rng, s.rows, s.dim) for s in fed}
        nq, k = 11, 10
        emb = {e: int_data(rng, nq, next(s.dim for s in fed if s.encoder == e)) for e in sorted({s.encoder for s in fed})}
        mask = rng.integers(0, 2, size=(nq, len(fed))).astype(bool)
        mask[0] = False
        mask[1] = True
        units = []
        for u in pl.ranks[rank]:
            pieces = [(corpora[s.sid][s.row_begin: s.row_begin + s.n_rows], s.id_offset, s.sid) for s in u.slices]
            units.append(("segments" if len(pieces) > 1 else "shard", _OracleSlice(pieces, len(pieces) > 1), [s.sid for s in u.slices],
                          None if len(pieces) > 1 else pieces[0][1]))
        pipe = PL.RetrievalPipeline([], [], router=_FixedRouter(torch.from_numpy(mask)), slots=pl.slots)
        pipe.units = units
        xq = {s.sid: torch.from_numpy(emb[s.encoder]) for s in fed}
        D, I = pipe.search(xq, k, xq_models=torch.zeros(nq, 1, 1))
        D2, I2 = pipe.search(xq, k, xq_models=torch.zeros(nq, 1, 1))          # the reused packed / gathered buffers
        want_D, want_I = expected_chain(O, fed, corpora, emb, mask, nq, k)
        assert I.numpy().tolist() == want_I and D.numpy().tolist() == want_D
        assert torch.equal(I, I2) and torch.equal(D, D2)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write(f"{len(units)} {pl.slots}")
    finally:
        dist.destroy_process_group()


def test_eight_ranks_feb4rag_shaped_plan(tmp_path):
    """World 8 on the CPU (gloo): the planner's plan of a FeB4RAG-shaped federation - 13 sources, 8 encoders, three widths, several cuts,
    ranks with different unit counts (unused exchange slots stay padding) - every rank searching its own units, ONE packed exchange,
    merge: identical to the reference's flow on the whole sources for every query, on every rank."""
    mp.spawn(_eight_rank_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    got = [open(tmp_path / f"ok{r}").read().split() for r in range(8)]
    assert len({g[1] for g in got}) == 1 and max(int(g[0]) for g in got) == int(got[0][1])
    assert len({g[0] for g in got}) > 1, "the ranks should hold different unit counts"


# ---- property test: random federations -------------------------------------------------------------------------------------------
def test_planner_invariants_on_random_federations():
    from hypothesis import given, settings, strategies as st
    dims = st.sampled_from([64, 100, 384, 768, 1024, 1536, 2048, 4096])
    source = st.tuples(st.integers(0, 40_000_000), dims, st.integers(0, 5), st.sampled_from(["ip", "ip", "ip", "l2"]))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(source, min_size=1, max_size=18), st.integers(1, 16))
    def check(specs, G):
        fed = [P.Source(i, rows, dim, f"enc{enc}-{dim}", metric) for i, (rows, dim, enc, metric) in enumerate(specs)]
        pl = P.plan(fed, G)
        assert len(pl.ranks) == G
        _covered(pl)
        cuts = 0
        for sid, src in pl.sources.items():
            sl = pl.slices_of(sid)
            cuts += len(sl) - 1
            if len(sl) > 1:
                assert all(s.n_rows >= P.MIN_SLICE_ROWS for s in sl)
        assert cuts <= G - 1
        for units in pl.ranks:
            keys = [u.group for u in units if u.group[2] != "l2"]
            assert len(keys) == len(set(keys))                      # the pieces of one group on a rank are ONE unit
            for u in units:
                assert len({(pl.sources[s.sid].encoder, P.padded_dim(pl.sources[s.sid].dim), pl.sources[s.sid].metric) for s in u.slices}) == 1
                assert [s.id_offset for s in u.slices] == sorted(s.id_offset for s in u.slices)
                if u.group[2] == "l2":
                    assert len(u.slices) == 1
        # never worse than everything on one rank, and within the model's granularity of the ideal share
        cost = P.CostModel()
        one = P.plan(fed, 1).predicted_ms[0]
        assert max(pl.predicted_ms) <= one + 1e-6
        biggest_uncuttable = max((cost.row_ms(s) * min(s.rows, 2 * P.MIN_SLICE_ROWS) for s in fed), default=0.0)
        assert max(pl.predicted_ms) <= one / G + biggest_uncuttable + cost.fixed_ms * (len(fed) + 1) + 1e-6
        assert P.plan(fed, G).describe() == pl.describe()           # deterministic

    check()


def test_synthetic_federation_rows_do_not_depend_on_the_slicing():
    """tools/workloads.fill_half: row r of source s is a pure function of (s, r) - whatever slices hold it (what makes
    bench.py --workload's result checksums comparable across placements and GPU counts)."""
    from tools import workloads as W
    src = P.Source(3, 2 * W.BLOCK + 1234, 32, "enc")
    whole = torch.zeros((src.rows, 128), dtype=torch.float16)
    W.fill_half(src, P.RowSlice(3, 0, src.rows), whole)
    cuts = [0, 256 * 7, W.BLOCK - 256, W.BLOCK + 512, src.rows]
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = torch.zeros((b - a, 128), dtype=torch.float16)
        W.fill_half(src, P.RowSlice(3, a, b - a), part)
        assert torch.equal(part, whole[a:b])
    assert float(whole[:, 32:].abs().max()) == 0.0 and float(whole[:, :32].float().norm(dim=1).sub(1).abs().max()) < 2e-3
    other = torch.zeros((16, 128), dtype=torch.float16)
    W.fill_half(P.Source(4, 16, 32, "enc"), P.RowSlice(4, 0, 16), other)
    assert not torch.equal(other, whole[:16])
