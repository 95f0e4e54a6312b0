"""GPU: the BASELINE.json configurations that are not the bench line.

  config 4  FeB4RAG corpora sharded several-per-GPU, mixed encoder widths, router mask, candidate exchange:
            two PROCESSES on the one device (gloo group; the exchange is staged through the host where RCCL is not
            available), real HIP scans, checked against the oracle chain on the union of the selected sources.
  config 5  80M x 768 bf16 per GPU (HBM-resident, 122.9 GB), k = 100, B = 256: planted-neighbour property at full size.
"""
import os
import socket

import numpy as np
import pytest
import torch

from tests.util import int_data

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _config4_rank(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from ragroute_amd import config as C
        from ragroute_amd.flat_index import FlatIndex
        from ragroute_amd.pipeline import RetrievalPipeline
        from ragroute_amd.router import Router
        from ragroute_amd.sharded import SHARD_SHIFT
        from tests.util import synth_router_case
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        sources = C.DATA_SOURCES["feb4rag"]                       # 13 sources (config.py:34), 8 encoders, 768/1024/4096 wide
        case = synth_router_case("feb4rag", 21, n_queries=24)
        model_of = {s: C.EMBEDDING_MODELS_PER_DATA_SOURCE["feb4rag"][s][0] for s in sources}
        width = {m: len(v) for m, v in case["queries"][0].items()}
        rng = np.random.default_rng(5)                            # same stream on every rank
        corpora = {s: int_data(rng, int(rng.integers(300, 9000)), width[model_of[s]]) for s in sources}
        nq, k = len(case["queries"]), C.K["feb4rag"]
        emb = {m: int_data(rng, nq, w) for m, w in width.items()}  # integer embeddings: exact scores, plenty of ties
        r = Router("feb4rag", sources, "ragroute")
        r.set_router(case["sd"], case["centroids"])
        mine = list(range(rank, len(sources), world))             # source s -> rank s mod G: 7 + 6 sources on two ranks
        shards = []
        for s in mine:
            idx = FlatIndex(corpora[sources[s]].shape[1], device=dev)
            idx.add(corpora[sources[s]])
            shards.append(idx)
        if r._folded is None:
            r._fold()
        pipe = RetrievalPipeline(shards, mine, router=r._folded)
        xq_models = r.pack_queries(emb).to(dev)
        D, I = pipe.search({s: torch.from_numpy(emb[model_of[sources[s]]]).to(dev) for s in mine}, k, xq_models=xq_models)
        assert pipe.slots == 7
        _, mask = r.route_batch(xq_models)
        mask = mask.cpu().numpy()
        assert mask.any() and not mask.all(), "the synthetic router should select some, not all, sources"
        D, I = D.cpu().numpy(), I.cpu().numpy()
        for q in range(nq):
            cand = []
            for s, name in enumerate(sources):                    # http_server.py:198-209: one message per selected source
                if mask[q, s]:
                    Ds, Is = O.flat_search_ip(corpora[name], emb[model_of[name]][q:q + 1], k)
                    cand += [(-float(d_), (s << SHARD_SHIFT) + int(i)) for d_, i in zip(Ds[0], Is[0]) if i >= 0]
            cand.sort()                                           # http_server.py:280-293 + rerank.py:3-9, ties by id
            want_I = [i for _, i in cand[:k]] + [-1] * (k - len(cand[:k]))
            want_D = [-d_ for d_, _ in cand[:k]] + [-np.inf] * (k - len(cand[:k]))
            assert I[q].tolist() == want_I, (q, I[q].tolist(), want_I)
            assert D[q].tolist() == want_D
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_config4_two_ranks_on_one_device(gpu, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_config4_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_config5_80m_bf16_k100(gpu):
    """80M x 768 bf16, k = 100, B = 256.  The oracle cannot scan 80M rows, so: for every query 100 rows at random
    positions are set to (1 + j/32) x query (gaps far above bf16 noise), which makes them the exact top-100 in a known
    order; everything else is N(0,1)/sqrt(d).  Also determinism and the sub-batch property."""
    from ragroute_amd.flat_index import FlatIndex
    free, _ = torch.cuda.mem_get_info(gpu)
    if free < 140e9:
        pytest.skip(f"needs 140 GB of free HBM, {free / 1e9:.0f} GB available")
    n, d, nq, k = 80_000_000, 768, 256, 100
    g = torch.Generator(device=gpu)
    g.manual_seed(99)
    xb = torch.empty((n, d), dtype=torch.bfloat16, device=gpu)
    for s in range(0, n, 1 << 21):
        e = min(n, s + (1 << 21))
        xb[s:e] = (torch.randn((e - s, d), generator=g, device=gpu) / d ** 0.5).to(torch.bfloat16)
    xq = torch.randn((nq, d), generator=g, device=gpu)
    xq = (xq / xq.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    pos = torch.randperm(n, generator=g, device=gpu)[: nq * k].reshape(nq, k)
    scale = 1 + torch.arange(k, device=gpu, dtype=torch.float32) / 32
    planted = (xq.float()[:, None, :] * scale[None, :, None]).to(torch.bfloat16)
    xb[pos.reshape(-1)] = planted.reshape(-1, d)
    idx = FlatIndex(d, dtype="bf16", device=gpu)
    idx.adopt(xb)
    D, I = idx.search_prepared(xq, k)
    want = (planted.float() * xq.float()[:, None, :]).sum(-1)
    order = torch.argsort(want, dim=1, descending=True, stable=True)
    assert torch.equal(I, torch.gather(pos, 1, order))
    assert float((D - torch.gather(want, 1, order)).abs().max()) < 2e-2      # bf16 products, |score| up to 4
    D2, I2 = idx.search_prepared(xq, k)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    D3, I3 = idx.search_prepared(xq[:5].contiguous(), k)
    assert torch.equal(I3, I[:5]) and torch.equal(D3, D[:5])
    del idx, xb
    torch.cuda.empty_cache()


def _bench_line(args, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=dict(os.environ, **(env or {})), capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, res.stdout[-1000:]
    return json.loads(lines[0])


def test_bench_workload_entry_point_is_placement_independent(gpu):
    """bench.py --workload feb4rag at 1/32 of the real row counts: one rank, two ranks on this device (gloo exchange) with the sliced
    and with the whole-source placement - three processes trees, the SAME result checksum, and every N > 1 line carries each rank's
    scan and local times."""
    common = ["--workload", "feb4rag", "--workload-scale", "32", "--steps", "3", "--warmup", "1", "--sustained-seconds", "0"]
    one = _bench_line(common)
    two = _bench_line(common + ["--gpus", "2"], {"RR_BENCH_BACKEND": "gloo", "RR_BENCH_ONE_DEVICE": "1"})
    whole = _bench_line(common + ["--gpus", "2", "--placement", "whole"], {"RR_BENCH_BACKEND": "gloo", "RR_BENCH_ONE_DEVICE": "1"})
    assert one["result_checksum"] == two["result_checksum"] == whole["result_checksum"]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["row_scale"] == "1/32"
    for line in (two, whole):
        assert len(line["per_rank_scan_ms"]["ranks"]) == 2 and len(line["per_rank_local_ms"]["ranks"]) == 2
        assert min(line["per_rank_local_ms"]["ranks"]) > 0 and line["rccl_ranks"] == 2
    assert sum(two["config"]["units_per_rank"]) >= 8 and two["roofline"]["frac"] > 0
