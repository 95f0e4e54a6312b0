"""GPU parity of the router MLP (K3), the merge (K4), the normalisation (K0) and the data-source glue,
all through the C ABI, against the oracle and the reference-generated golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import half_round, int_data, synth_medrag_corpus, synth_router_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOGIT_TOL = 2e-4   # f32 re-association of the folded fc1 (K up to 4096) vs the reference's f64 features + torch sgemm


def _router(dataset, seed):
    from ragroute_amd.router import Router
    case = synth_router_case(dataset, seed)
    r = Router(dataset, case["sources"], "ragroute")
    mean, scale = case["scaler"] if case["scaler"] is not None else (None, None)
    r.set_router(case["sd"], case["centroids"], mean, scale)
    return r, case


@pytest.mark.parametrize("dataset", ["medrag", "feb4rag", "wikipedia"])
def test_router_matches_reference_golden(gpu, dataset):
    """Logits within LOGIT_TOL of the reference's; selected sources identical wherever the reference logit is
    further than LOGIT_TOL from the decision boundary (router.py:276-282)."""
    g = json.load(open(os.path.join(GOLD, "router.json")))[dataset]
    r, case = _router(dataset, g["seed"])
    thr = 0.4924 if dataset == "medrag" else 0.5
    logit_thr = np.log(thr / (1 - thr))
    for q, want_logits, want_sel in zip(case["queries"], g["logits"], g["selected"]):
        logits, mask = r.route_batch(r.pack_queries(q))
        logits = logits[0].cpu().numpy()
        assert np.allclose(logits, want_logits, atol=LOGIT_TOL, rtol=0), np.abs(logits - want_logits).max()
        sel = r.select_relevant_sources(q)
        for c, wl in zip(case["sources"], want_logits):
            if abs(wl - logit_thr) > LOGIT_TOL:
                assert (c in sel) == (c in want_sel)
        assert sel == [c for c, m in zip(case["sources"], mask[0].cpu().numpy()) if m]


def test_router_batch_equals_single(gpu):
    r, case = _router("feb4rag", 12)
    names = r.model_names
    batch = {m: np.stack([np.pad(q[m], (0, 4096 - len(q[m]))) for q in case["queries"]]) for m in names}
    L, M = r.route_batch(r.pack_queries(batch))
    for i, q in enumerate(case["queries"]):
        l1, m1 = r.route_batch(r.pack_queries(q))
        assert torch.equal(l1[0], L[i]) and torch.equal(m1[0], M[i])


def test_router_large_batch_vs_oracle(gpu):
    """10^4 random rows through the unfolded forward (CorpusRoutingNN.forward) vs the numpy oracle: decision flips
    only where the logit is within LOGIT_TOL of the boundary."""
    from oracle import oracle as O
    from ragroute_amd.router import CorpusRoutingNN
    case = synth_router_case("medrag", 21)
    net = CorpusRoutingNN(1540)
    net.load_state_dict(case["sd"])
    x = np.random.default_rng(0).standard_normal((10_000, 1540)).astype(np.float32)
    got = net(x).cpu().numpy().reshape(-1)
    want = O.corpus_routing_nn(x, case["sd"]).reshape(-1)
    assert np.abs(got - want).max() < LOGIT_TOL
    flips = (got > 0) != (want > 0)
    assert np.all(np.abs(want[flips]) < LOGIT_TOL)


def test_merge_topk_vs_oracle(gpu):
    from oracle import oracle as O
    from ragroute_amd.rerank import merge_topk
    rng = np.random.default_rng(1)
    for nq, m, k in [(256, 256, 32), (256, 800, 100), (3, 1, 4), (5, 4096, 10), (2, 130, 10)]:
        D = np.round(rng.standard_normal((nq, m)) * 4).astype(np.float32) / 4   # many ties
        I = np.stack([rng.permutation(10 * m)[:m] for _ in range(nq)]).astype(np.int64) + (7 << 40)
        I[0, : m // 2] = -1
        D[-1, 0] = np.nan
        for desc in (True, False):
            Dg, Ig = merge_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda(), k, desc)
            Io = I.copy()
            Io[np.isnan(D)] = -1
            Dr, Ir = O.merge_topk(np.nan_to_num(D), Io, k, desc)
            assert np.array_equal(Ig.cpu().numpy(), Ir) and np.array_equal(Dg.cpu().numpy(), Dr)


def test_merge_compares_zeros_as_ieee_and_keeps_their_sign(gpu):
    """-0.0 == +0.0 for the merge (numpy's argsort of rerank.py:5,30 and the oracle compare floats, not bit patterns): ties between
    them go by ascending id, and each candidate keeps its own sign bit on output."""
    from oracle import oracle as O
    from ragroute_amd.rerank import merge_topk
    D = np.array([[0.0, -0.0, -0.0, 0.0, -1.0, 0.0, -0.0, 1.0]], np.float32)
    I = np.array([[50, 10, 40, 20, 5, 30, 60, 70]], np.int64)
    for desc in (True, False):
        Dg, Ig = merge_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda(), 8, desc)
        Dr, Ir = O.merge_topk(D, I, 8, desc)
        assert np.array_equal(Ig.cpu().numpy(), Ir)
        assert Dg.cpu().numpy().tobytes() == Dr.tobytes()


def test_merge_gathered_reads_the_exchange_buffer_in_place(gpu):
    """rr_merge_topk_gathered on a buffer laid out as the all-gather leaves it ([rank][D f32[slots][B][k] | pad | I i64[slots][B][k]])
    against the layout statement (sharded.unpack_gathered) + the oracle's merge; odd B*k*slots exercises the 8-byte pad."""
    from oracle import oracle as O
    from ragroute_amd import sharded as S
    rng = np.random.default_rng(5)
    for world, slots, B, k_in, k in [(1, 1, 256, 32, 32), (8, 1, 256, 32, 32), (3, 2, 7, 5, 9), (2, 3, 1, 1, 4), (8, 2, 256, 100, 100), (4, 1, 33, 64, 10)]:
        bufs = []
        for r in range(world):
            buf, D, I = S.alloc_packed(B, k_in, gpu, slots)
            D = D.view(slots, B, k_in)
            I = I.view(slots, B, k_in)
            D.copy_(torch.from_numpy(np.round(rng.standard_normal((slots, B, k_in)) * 3).astype(np.float32) / 3))   # ties across ranks
            I.copy_(torch.from_numpy(rng.permutation(slots * B * k_in).reshape(slots, B, k_in).astype(np.int64) + (r << 40)))
            if r == world - 1:
                I[slots - 1, :, k_in // 2:] = -1          # a short source: padding entries
                D[slots - 1, :, k_in // 2:] = float("-inf")
            bufs.append(buf)
        out = torch.stack(bufs)                          # what all_gather_into_tensor writes: [world, bytes per rank]
        assert out.shape[1] == S.packed_layout(B, k_in, slots)[2]
        for desc in (True, False):
            Dg, Ig = S.merge_gathered(out, B, k_in, slots, k, desc)
            Dl, Il = S.unpack_gathered(out, B, k_in, slots)
            Dr, Ir = O.merge_topk(Dl.cpu().numpy(), Il.cpu().numpy(), k, desc)
            assert np.array_equal(Ig.cpu().numpy(), Ir) and np.array_equal(Dg.cpu().numpy(), Dr), (world, slots, B, k_in, k, desc)


def test_sharded_search_results_do_not_alias_the_reused_exchange_buffer(gpu):
    """ShardedFlatSearch reuses its packed candidate buffer; what search N returned must still be search N's after search N+1
    (one rank, one shard, no mask: the case that used to return views of that buffer)."""
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.sharded import ShardedFlatSearch
    rng = np.random.default_rng(9)
    xb = int_data(rng, 20_000, 128)
    idx = FlatIndex(128, device=gpu)
    idx.add(xb)
    sh = ShardedFlatSearch([idx], [3])
    q1, q2 = int_data(rng, 16, 128), int_data(rng, 16, 128)
    D1, I1 = sh.search(idx.prepare_queries(q1), 10)
    keep_D, keep_I = D1.clone(), I1.clone()
    D2, I2 = sh.search(idx.prepare_queries(q2), 10)
    torch.cuda.synchronize()
    assert torch.equal(D1, keep_D) and torch.equal(I1, keep_I)
    assert D1.data_ptr() != D2.data_ptr()
    Dr, Ir = O.flat_search_ip(xb, q1, 10)
    assert np.array_equal(I1.cpu().numpy(), Ir + (3 << 40)) and np.array_equal(D1.cpu().numpy(), Dr)


def test_normalize_l2_and_ingest(gpu):
    from oracle import oracle as O
    from ragroute_amd._lib import check, lib
    from ragroute_amd.flat_index import normalize_L2
    rng = np.random.default_rng(2)
    for n, d in [(1, 768), (1000, 768), (37, 100), (5, 4096)]:
        x = (rng.standard_normal((n, d)) * 3).astype(np.float32)
        if n > 3:
            x[3] = 0
        want = x.copy()
        O.normalize_L2(want)
        got = x.copy()
        normalize_L2(got)
        assert np.allclose(got, want, atol=1e-6, rtol=1e-6)
        if n > 3:
            assert (got[3] == 0).all()
        xt = torch.from_numpy(x).cuda()
        for dtype, tdt in ((0, torch.float16), (1, torch.bfloat16)):
            dim = ((d + 127) // 128) * 128
            out = torch.full((n, dim), 7, dtype=tdt, device="cuda")
            check(lib().rr_rows_to_half(xt.data_ptr(), n, d, d, out.data_ptr(), dtype, dim, 1, None), "rows_to_half")
            torch.cuda.synchronize()
            ref = torch.from_numpy(want).to(tdt)
            assert torch.equal(out[:, d:].cpu(), torch.zeros((n, dim - d), dtype=tdt))
            diff = (out[:, :d].cpu().float() - ref.float()).abs().max().item()
            assert diff <= (2e-3 if dtype == 0 else 1.6e-2) * float(np.abs(want).max())


def test_data_source_glue_matches_reference_golden(gpu, tmp_path, monkeypatch):
    """retrieve_docs_medrag / retrieve_docs_wikipedia return exactly the tuples the reference's DataSource
    returned on the same corpus (fixture: tests/golden/data_source_glue.json)."""
    from oracle import oracle as O
    from ragroute_amd import config as C
    from ragroute_amd import data_source as DS
    g = json.load(open(os.path.join(GOLD, "data_source_glue.json")))
    xb, metadatas, chunks = synth_medrag_corpus(g["medrag"]["corpus_seed"])
    monkeypatch.setattr(C, "MEDRAG_DIR", str(tmp_path))
    monkeypatch.setattr(C, "WIKIPEDIA_DIR", str(tmp_path))
    ds = DS.DataSource(0, "medrag", "textbooks")
    os.makedirs(ds.index_dir)
    os.makedirs(tmp_path / "textbooks" / "chunk")
    DS.write_faiss_flat_index(ds.index_path, xb)
    open(ds.doc_ids_path, "w").write("\n".join(json.dumps(m) for m in metadatas))
    for b, lines in chunks.items():
        open(tmp_path / "textbooks" / "chunk" / f"{b}.jsonl", "w").write("\n".join(json.dumps(l) for l in lines))
    ds.load_faiss_index()
    queries = np.random.default_rng(g["medrag"]["query_seed"]).integers(-2, 3, size=(3, 768)).astype(np.float32)
    for q, want in zip(queries, g["medrag"]["results"]):
        indices, docs, scores = ds.retrieve_docs_medrag(q.reshape(1, -1), g["medrag"]["k"])
        assert indices == want["indices"] and docs == want["docs"] and scores == want["scores"]
        assert all(isinstance(s, float) for s in scores)
    # batched form = per-query tuples
    allres = ds.retrieve_docs_medrag(queries, g["medrag"]["k"])
    assert [r[0] for r in allres] == [w["indices"] for w in g["medrag"]["results"]]
    # feb4rag: docid strings, corpus.jsonl lookups (None when missing), no scores
    monkeypatch.setattr(C, "FEB4RAG_DIR", str(tmp_path))
    fs = DS.DataSource(1, "feb4rag", "scifact")
    os.makedirs(fs.index_dir)
    docids = [f"doc-{i * 7 % 1000}-{i}" for i in range(xb.shape[0])]
    DS.write_faiss_flat_index(fs.index_path, xb)
    json.dump(docids, open(fs.doc_ids_path, "w"))
    cdir = tmp_path / "dataset_creation/original_dataset" / "scifact" / "scifact"
    os.makedirs(cdir)
    with open(cdir / "corpus.jsonl", "w") as f:
        for i, did in enumerate(docids):
            if i % 5 != 0:
                f.write(json.dumps({"_id": did, "title": f"t{i}", "text": f"body {i}"}) + "\n")
    fs.load_faiss_index()
    for q, want in zip(queries, g["feb4rag"]["results"]):
        ids, docs, scores = fs.retrieve_docs_fed4rag(q.reshape(1, -1), g["feb4rag"]["k"])
        assert ids == want["ids"] and docs == want["docs"] and scores == [] == want["scores"]
    # wikipedia: normalised query, row ids, (title, text) docs
    n = xb.shape[0]
    ws = DS.DataSource(3, "wikipedia", "3")
    xbn = xb.copy()
    xbn[:, 0] += 3
    O.normalize_L2(xbn)
    ws.set_index(half_round(xbn), [], [f"title {i}" for i in range(n)], [f"text {i}" for i in range(n)])
    for q, want in zip(queries, g["wikipedia"]["results"]):
        ids, docs, scores = ws.retrieve_docs_wikipedia(q.copy().reshape(1, -1), g["wikipedia"]["k"])
        assert np.allclose(scores, want["scores"], atol=1e-3)
        if ids != want["ids"]:   # near-ties may swap within the f32 tolerance
            assert sorted(ids) == sorted(want["ids"]) or np.min(np.abs(np.diff(want["scores"]))) < 2e-3
        else:
            assert [list(d) for d in docs] == want["docs"]


def test_batched_serving_replies_match_the_oracle(gpu):
    """40 concurrent requests through DataSource.handle_query are coalesced into a few GPU batches and every reply holds what
    the reference's single-query path returns for that request (data_source.py:113-132, 165-194) — expected values from the
    oracle's search + the row -> metadata -> chunk lookups, not from the object's own single-query method."""
    import asyncio
    from oracle import oracle as O
    from ragroute_amd import data_source as DS
    xb, metadatas, chunks = synth_medrag_corpus(5)
    ds = DS.DataSource(0, "medrag", "textbooks")
    ds.set_index(xb, metadatas)
    ds.cache_jsonl = {b: [json.dumps(l) for l in lines] for b, lines in chunks.items()}
    queries = np.random.default_rng(1).integers(-2, 3, size=(40, 768)).astype(np.float32)

    async def go():
        return await asyncio.gather(*[ds.handle_query({"id": f"q{i}", "embedding": q.tolist()}) for i, q in enumerate(queries)])
    replies = asyncio.run(go())
    assert ds._batcher.items_run == 40 and ds._batcher.batches_run <= 4
    Dr, Ir = O.flat_search_ip(xb, queries, 32)
    for i, rep in enumerate(replies):
        want_idx = [metadatas[int(r)] for r in Ir[i]]                                   # data_source.py:190
        want_docs = [chunks[m["source"]][m["index"]] for m in want_idx]                # data_source.py:166-183
        assert rep["query_id"] == f"q{i}" and rep["client_id"] == 0 and rep["name"] == "textbooks"
        assert rep["indices"] == want_idx and rep["docs"] == want_docs
        assert rep["scores"] == [float(s) for s in Dr[i]]                              # integer data: exact
        json.dumps(rep)  # wire format must be JSON-serialisable


def test_centroid_kernel(gpu):
    from ragroute_amd.flat_index import FlatIndex
    rng = np.random.default_rng(3)
    for n, d, dtype in [(5000, 768, "fp16"), (777, 100, "bf16"), (3000, 1024, "fp16"), (1, 64, "fp16")]:
        x = half_round(rng.standard_normal((n, d)).astype(np.float32), dtype)
        idx = FlatIndex(d, dtype=dtype, device=gpu)
        idx.add(x)
        got = idx.centroid().cpu().numpy()
        assert got.shape == (d,)
        assert np.allclose(got, x.astype(np.float64).mean(0), atol=2e-5)


def test_pipeline_route_search_merge(gpu):
    """Whole hot path on device (router -> masked shard scans -> merge) == oracle over the shards each query was routed to."""
    from oracle import oracle as O
    from ragroute_amd import config as C
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.pipeline import RetrievalPipeline
    from ragroute_amd.sharded import SHARD_SHIFT
    r, case = _router("medrag", 11)
    rng = np.random.default_rng(4)
    parts = [int_data(rng, n, 768) for n in (9000, 20_000, 700, 12_345)]
    shards = []
    for p in parts:
        idx = FlatIndex(768, device=gpu)
        idx.add(p)
        shards.append(idx)
    xq = int_data(rng, 256, 768)   # BASELINE config 3: medrag, 4 corpora on one GPU + router forward, query batch 256
    r._fold()
    pipe = RetrievalPipeline(shards, [0, 1, 2, 3], router=r._folded)
    xt = torch.from_numpy(xq).to(gpu)
    D, I = pipe.search(xt, 32)
    _, mask = pipe.route(xt[:, None, :].contiguous())
    mask = mask.cpu().numpy()
    assert mask.any() and not mask.all()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    for q in range(0, 256, 5):
        sel = [s for s in range(4) if mask[q, s]]
        if not sel:
            assert (I[q] == -1).all()
            continue
        cat = np.concatenate([parts[s] for s in sel])
        gids = np.concatenate([np.arange(len(parts[s])) + (s << SHARD_SHIFT) for s in sel])
        Dr, Ir = O.flat_search_ip(cat, xq[q:q + 1], 32)
        assert np.array_equal(D[q], Dr[0]) and np.array_equal(I[q], gids[Ir[0]])


def test_pipeline_mixed_encoders_feb4rag_shapes(gpu):
    """FeB4RAG-like federation (BASELINE config 4 on one rank): sources of different embedding widths (768 / 1024 / 4096), each
    searched with its own encoder's query embedding, merged into one top-10 by score (k = 10, config.py:99)."""
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.pipeline import RetrievalPipeline
    from ragroute_amd.sharded import SHARD_SHIFT
    rng = np.random.default_rng(6)
    dims, sizes = [768, 1024, 4096, 1024], [12_000, 9_000, 3_000, 20_500]
    parts = [int_data(rng, n, d, -1, 2) for n, d in zip(sizes, dims)]
    queries = {s: int_data(rng, 33, d, -1, 2) for s, d in enumerate(dims)}
    shards = []
    for p_, d in zip(parts, dims):
        idx = FlatIndex(d, device=gpu)
        idx.add(p_)
        shards.append(idx)
    pipe = RetrievalPipeline(shards, [0, 1, 2, 3])
    D, I = pipe.search({s: torch.from_numpy(q).to(gpu) for s, q in queries.items()}, 10)
    cand_D, cand_I = [], []
    for s in range(4):
        Ds, Is = O.flat_search_ip(parts[s], queries[s], 10)
        cand_D.append(Ds)
        cand_I.append(Is + (s << SHARD_SHIFT))
    Dr, Ir = O.merge_topk(np.concatenate(cand_D, 1), np.concatenate(cand_I, 1), 10, True)
    assert np.array_equal(D.cpu().numpy(), Dr) and np.array_equal(I.cpu().numpy(), Ir)


def test_torch_custom_ops(gpu):
    """torch.ops.ragroute.* reach the same kernels: convert -> flat_topk -> merge_topk against the oracle."""
    from oracle import oracle as O
    import ragroute_amd.torch_ops  # noqa: F401  (registers the ops)
    rng = np.random.default_rng(9)
    xb, xq = int_data(rng, 25_000, 300), int_data(rng, 12, 300)
    dim = 384
    xbh = torch.ops.ragroute.rows_to_half(torch.from_numpy(xb).to(gpu), dim)
    xqh = torch.ops.ragroute.rows_to_half(torch.from_numpy(xq).to(gpu), dim)
    D, I = torch.ops.ragroute.flat_topk(xbh, xqh, 10)
    Dr, Ir = O.flat_search_ip(xb, xq, 10)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
    D2, I2 = torch.ops.ragroute.flat_topk(xbh, xqh, 10, 0, True)
    Dl, Il = O.flat_search_l2(xb, xq, 10)
    assert np.array_equal(I2.cpu().numpy(), Il) and np.array_equal(D2.cpu().numpy(), Dl)
    Dm, Im = torch.ops.ragroute.merge_topk(torch.cat([D, D], 1), torch.cat([I, I + (1 << 40)], 1), 10)
    assert torch.equal(Im[:, 0], I[:, 0]) and torch.equal(Dm[:, 0], D[:, 0])
    x = torch.from_numpy(rng.standard_normal((50, 77)).astype(np.float32)).to(gpu)
    want = x / x.norm(dim=1, keepdim=True)
    torch.ops.ragroute.l2_normalize_(x)
    assert torch.allclose(x, want, atol=1e-6)


def test_torch_router_op_equals_router_class(gpu):
    """torch.ops.ragroute.router_mlp on the folded weights == Router.route_batch (same kernel, same inputs)."""
    import ragroute_amd.torch_ops  # noqa: F401
    from ragroute_amd.router import Router
    case = synth_router_case("medrag", 3, n_queries=20)
    r = Router("medrag", case["sources"], "ragroute")
    r.set_router(case["sd"], case["centroids"], *case["scaler"])
    q = {m: np.stack([qq[m] for qq in case["queries"]]) for m in case["queries"][0]}
    xq = r.pack_queries(q).to(gpu)
    logits, mask = r.route_batch(xq)
    t = r._folded.t
    sd = case["sd"]
    l2, m2 = torch.ops.ragroute.router_mlp(xq, t["w1q"], t["c1"], t["ln1_g"], t["ln1_b"], t["w2"], t["b2"], t["ln2_g"], t["ln2_b"], t["w3"],
                                           t["model_of_source"], float(np.asarray(sd["fc3.bias"]).reshape(-1)[0]), r._folded.struct.prob_threshold)
    assert torch.equal(l2, logits) and torch.equal(m2, mask)


def test_data_source_serves_an_l2_index_wider_than_768(gpu, tmp_path, monkeypatch):
    """An IxF2 file decides the metric (faiss.read_index, data_source.py:71); L2 works at every width (here FeB4RAG's 1024)."""
    from oracle import oracle as O
    from ragroute_amd import config as C
    from ragroute_amd import data_source as DS
    monkeypatch.setattr(C, "FEB4RAG_DIR", str(tmp_path))
    rng = np.random.default_rng(4)
    xb = int_data(rng, 9000, 1024)
    ds = DS.DataSource(0, "feb4rag", "msmarco")     # e5-large: 1024 wide (config.py:45)
    os.makedirs(ds.index_dir)
    DS.write_faiss_flat_index(ds.index_path, xb, metric="l2")
    docids = [f"m{i}" for i in range(len(xb))]
    json.dump(docids, open(ds.doc_ids_path, "w"))
    cdir = tmp_path / "dataset_creation/original_dataset" / "msmarco" / "msmarco"
    os.makedirs(cdir)
    open(cdir / "corpus.jsonl", "w").write("\n".join(json.dumps({"_id": d, "text": d}) for d in docids))
    ds.load_faiss_index()
    assert ds.index_metric == "l2" and ds.faiss_indexes[0].metric == "l2" and ds.faiss_indexes[0].d == 1024
    xq = int_data(rng, 5, 1024)
    Dr, Ir = O.flat_search_l2(xb, xq, 10)
    D, I = ds.faiss_indexes[0].search(xq, 10)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    ids, docs, scores = ds.retrieve_docs_fed4rag(xq[:1], 10)
    assert ids == [docids[i] for i in Ir[0]] and scores == []


@pytest.mark.parametrize("dataset,nq", [("feb4rag", 256), ("feb4rag", 45), ("wikipedia", 200), ("medrag", 256)])
def test_router_batched_matrix_core_form_vs_oracle(gpu, dataset, nq):
    """Batches of >= 32 queries take the f32 matrix-core kernels (router_fc1_kernel + router_head_kernel; medrag's small router
    keeps the latency-oriented kernel): logits within LOGIT_TOL of the oracle's per-query forward (router.py:241-275), decisions
    identical off the boundary, bit-identical from run to run, ragged batch sizes."""
    from oracle import oracle as O
    from ragroute_amd import config as C
    r, case = _router(dataset, 33)
    rng = np.random.default_rng(nq)
    sources, d_max = case["sources"], case["d_max"]
    dims = {m: len(v) for m, v in case["queries"][0].items()}
    batch = {m: rng.standard_normal((nq, dm)).astype(np.float32) for m, dm in dims.items()}
    xq = r.pack_queries(batch)
    L, M = r.route_batch(xq)
    L2, M2 = r.route_batch(xq)
    assert torch.equal(L, L2) and torch.equal(M, M2)
    L, M = L.cpu().numpy(), M.cpu().numpy()
    model_of = {s: C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][s][0] for s in sources}
    ids = {"medrag": C.MEDRAG_SOURCE_TO_ID, "feb4rag": C.FEB4RAG_SOURCE_TO_ID, "wikipedia": None}[dataset]
    cents = {s: np.pad(case["centroids"][s], (0, d_max - len(case["centroids"][s]))) for s in sources}
    mean, scale = case["scaler"] if case["scaler"] is not None else (None, None)
    thr = 0.4924 if dataset == "medrag" else 0.5
    boundary = np.log(thr / (1 - thr))
    for i in range(0, nq, 7):
        want = O.router_logits(dataset, sources, model_of, ids, d_max, {m: batch[m][i] for m in dims}, cents, case["sd"], mean, scale)
        assert np.abs(L[i] - want).max() < LOGIT_TOL, (i, np.abs(L[i] - want).max())
        off = np.abs(want - boundary) > LOGIT_TOL
        assert np.array_equal(M[i][off], (want > boundary)[off])


@pytest.mark.parametrize("dataset", ["feb4rag", "wikipedia"])
def test_router_decisions_do_not_depend_on_how_many_requests_shared_the_window(gpu, dataset):
    """The router service coalesces concurrent requests (router.py: QueryBatcher, up to 256 per window); windows of fewer than 32
    queries take the latency-oriented kernel, larger ones the matrix-core form, which sums fc1 in another order.  The same query must
    get the same answer whatever the load: logits of the two forms within LOGIT_TOL of each other (both are within LOGIT_TOL of the
    reference's), and identical source masks wherever the logit is further than LOGIT_TOL from the decision boundary — the only
    place where the reference itself (one query per forward, router.py:207-219) and either form may differ."""
    r, case = _router(dataset, 35)
    rng = np.random.default_rng(5)
    dims = {m: len(v) for m, v in case["queries"][0].items()}
    nq = 256
    batch = {m: rng.standard_normal((nq, dm)).astype(np.float32) for m, dm in dims.items()}
    xq = r.pack_queries(batch)
    L_big, M_big = r.route_batch(xq)                       # matrix-core form
    thr = 0.5
    boundary = float(np.log(thr / (1 - thr)))
    for size in (1, 7, 31, 32, 100):                        # windows of other sizes, the same queries
        for start in (0, 64, nq - size):
            L, M = r.route_batch(xq[start:start + size].contiguous())
            dl = (L - L_big[start:start + size]).abs().max().item()
            assert dl < LOGIT_TOL, (size, start, dl)
            off = (L_big[start:start + size] - boundary).abs() > LOGIT_TOL
            assert torch.equal(M[off], M_big[start:start + size][off])
