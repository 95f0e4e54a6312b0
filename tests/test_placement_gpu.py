"""GPU: row slices as search units, through the C ABI.

Two PROCESSES on the one device (gloo group, exchange staged through the host where RCCL has one device only), real HIP scans
built by `RetrievalPipeline.from_placement`: one source cut in THREE slices over the two ranks (the pieces of one encoder on
a rank are one segmented search, a lone piece a plain search with the slice's id offset), route mask, ties across the cuts —
bit-exact against the reference's flow on the WHOLE sources (tests/test_placement.py::expected_chain: one top-k per selected
source, data_source.py:158-163; concatenate, keep the k best, http_server.py:280-293 + rerank.py:3-9)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FixedRouter:
    def __init__(self, mask):
        self.mask = mask

    def run(self, xq_models):
        return None, self.mask


def _rank(rank, world, port, out_dir, planned, half_fill):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from ragroute_amd import placement as P
        from ragroute_amd.flat_index import SegmentedIndex
        from ragroute_amd.pipeline import RetrievalPipeline
        from tests.test_placement import expected_chain, slice_case
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        fed, corpora, emb, mask, nq, k, pl = slice_case(seed=17 + int(planned), big_rows=3000 if planned else 1500)
        if planned:
            pl = P.plan(fed, world, cost=P.CostModel(fixed_ms=1e-7, segment_ms=0.0), min_slice_rows=256)
            assert len(pl.slices_of(0)) >= 2
        router = _FixedRouter(torch.from_numpy(mask).to(dev))
        if half_fill:   # rows written in the storage type straight into the index's matrix (bench.py's synthetic corpora)
            def fill(src, sl, out):
                out[:, : src.dim] = torch.from_numpy(corpora[sl.sid][sl.row_begin: sl.row_begin + sl.n_rows]).to(dev).half()
            pipe = RetrievalPipeline.from_placement(pl, rank, fill_half=fill, router=router, device=dev)
        else:
            pipe = RetrievalPipeline.from_placement(pl, rank, rows_f32=lambda sid, a, b: corpora[sid][a:b], router=router, device=dev)
        assert pipe.slots == pl.slots
        if not planned:
            kinds = [type(u[1]).__name__ for u in pipe.units]
            assert kinds == (["SegmentedIndex"] if rank == 0 else ["FlatIndex", "FlatIndex"])
            if rank == 0:
                assert isinstance(pipe.units[0][1], SegmentedIndex) and pipe.units[0][1].rows == [512, 476, 300]
        xq = {s.sid: torch.from_numpy(emb[s.encoder]).to(dev) for s in fed}
        D, I = pipe.search(xq, k, xq_models=torch.zeros(nq, 1, 1, device=dev))
        want_D, want_I = expected_chain(O, fed, corpora, emb, mask, nq, k)
        assert I.cpu().numpy().tolist() == want_I
        assert D.cpu().numpy().tolist() == want_D
        # a second search through the reused packed buffer gives the same answer
        D2, I2 = pipe.search(xq, k, xq_models=torch.zeros(nq, 1, 1, device=dev))
        assert torch.equal(I, I2) and torch.equal(D, D2)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("planned,half_fill", [(False, False), (True, True)])
def test_one_source_in_three_slices_over_two_ranks_on_one_device(gpu, tmp_path, planned, half_fill):
    import torch.multiprocessing as mp
    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path), planned, half_fill), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_sliced_plan_equals_whole_source_plan_on_one_rank(gpu):
    """Same federation, one rank: the planner's units (segmented where encoders are shared) against one search per whole
    source — identical ids and scores, with a route mask, k above a small source's row count, and planted ties."""
    from ragroute_amd import placement as P
    from ragroute_amd.pipeline import RetrievalPipeline
    from tests.util import int_data
    rng = np.random.default_rng(23)
    fed = [P.Source(0, 40_000, 96, "a"), P.Source(1, 7, 96, "a"), P.Source(2, 9_000, 1024, "b"), P.Source(3, 0, 96, "a")]
    corpora = {s.sid: int_data(rng, s.rows, s.dim) for s in fed}
    corpora[0][30_000:30_050] = corpora[0][10:60]
    nq, k = 33, 16
    emb = {e: int_data(rng, nq, d) for e, d in (("a", 96), ("b", 1024))}
    mask = torch.from_numpy(rng.integers(0, 2, size=(nq, 4)).astype(bool)).to(gpu)
    mask[0] = False
    xq = {s.sid: torch.from_numpy(emb[s.encoder]).to(gpu) for s in fed}
    zeros = torch.zeros(nq, 1, 1, device=gpu)
    res = []
    for pl in (P.whole_source_plan(fed, 1), P.plan(fed, 1),
               P.Placement([[P.Unit(fed[0].group, (P.RowSlice(0, 0, 10_240), P.RowSlice(0, 10_240, 20_480), P.RowSlice(0, 30_720, 9_280),
                                                  P.RowSlice(1, 0, 7), P.RowSlice(3, 0, 0))),
                             P.Unit(fed[2].group, (P.RowSlice(2, 0, 4_096), P.RowSlice(2, 4_096, 4_904)))]], [0.0], {s.sid: s for s in fed})):
        pipe = RetrievalPipeline.from_placement(pl, 0, rows_f32=lambda sid, a, b: corpora[sid][a:b], router=_FixedRouter(mask), device=gpu)
        D, I = pipe.search(xq, k, xq_models=zeros)
        res.append((D.cpu(), I.cpu()))
    for D, I in res[1:]:
        assert torch.equal(I, res[0][1]) and torch.equal(D, res[0][0])
