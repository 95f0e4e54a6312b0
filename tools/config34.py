"""BASELINE configs 3 and 4 at their real corpus SHAPES on one MI355X (synthetic embeddings, resident in HBM).

    python tools/config34.py medrag|feb4rag [batches] [per-source|segments]
    python tools/config34.py medrag|feb4rag [batches] --plan G [--whole] [--with-one]

  --plan G: the G-GPU layout of ragroute_amd/placement.py (balanced row slices; --whole: source s -> GPU s mod G) rehearsed on ONE
  GPU: each rank's units are built, timed (router + local scans + merge over G x slots x k candidates) and freed in turn; prints
  every rank's unit list, predicted and measured ms, max / mean, and with --with-one the one-GPU step of the same federation
  (= the strong-scaling ratio the layout can reach: the exchange of G x slots x B x k x 12 bytes is not in it).

  segments (default): sources that share an encoder (config.py:37-71) live in one SegmentedIndex and are searched in ONE pass
  (rr_flat_search_segments); per-source: one search per source, as in round 2.

  medrag   (config 3: "medrag 4 corpora on 1 GPU + router MLP forward, query batch=256"): pubmed / statpearls / textbooks /
           wikipedia at MedRAG's snippet counts, 768 wide (config.py:28, 45), k = 32, router over 4 sources.
  feb4rag  (config 4's work, whole federation on one GPU instead of 2 sources per GPU on 8): the 13 BEIR corpora at their
           document counts, each as wide as its encoder (config.py:32-46: 768 / 1024 / 4096), k = 10, router over 13 sources
           and 8 encoders, every source searched with its own embedding (http_server.py:198-209), merge over 13 k.

One step = router MLP -> per source: query conversion + exact top-k with the route mask folded in -> merge; prints one JSON
line: ms per batch (one HIP-event pair per batch, median), queries/s, corpus bytes / time against the 8 TB/s HBM peak, and
the per-source search times (each source alone, same events)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ragroute_amd import config as C
from ragroute_amd.flat_index import FlatIndex, SegmentedIndex
from ragroute_amd.pipeline import RetrievalPipeline
from ragroute_amd.router import CorpusRoutingNN, FoldedRouter

ROWS = {   # MedRAG snippet counts (Xiong et al. 2024, table 1) and BEIR document counts (Thakur et al. 2021, table 1)
    "medrag": {"pubmed": 23_900_000, "statpearls": 301_200, "textbooks": 125_800, "wikipedia": 29_900_000},
    "feb4rag": {"msmarco": 8_841_823, "trec-covid": 171_332, "nfcorpus": 3_633, "scidocs": 25_657, "nq": 2_681_468,
                "hotpotqa": 5_233_329, "fiqa": 57_638, "arguana": 8_674, "webis-touche2020": 382_545, "dbpedia-entity": 4_635_922,
                "fever": 5_416_568, "climate-fever": 5_416_593, "scifact": 5_183},
}
WIDTH = {"e5-large": 1024, "SGPT-5.8B-weightedmean-msmarco-specb-bitfit": 4096, "UAE-Large-V1": 1024, "all-mpnet-base-v2": 768,
         "multilingual-e5-large": 1024, "ember-v1": 1024, "e5-base": 768, "gte-base": 768, "ncbi/MedCPT-Query-Encoder": 768}


def make(n, d, dim, seed, dev, out=None):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    xb = torch.zeros((n, dim), dtype=torch.float16, device=dev) if out is None else out
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        x = torch.randn((e - s, d), generator=g, device=dev)
        xb[s:e, :d] = (x / x.norm(dim=1, keepdim=True)).half()
    return xb


def timed_events(fn, n):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)


def plan_mode(dataset, batches, G, whole, with_one):
    from ragroute_amd import placement as P
    from ragroute_amd.sharded import alloc_packed, merge_gathered
    from tools import workloads as W
    dev = torch.device("cuda:0")
    fed = P.federation(dataset)
    k, B = C.K[dataset], 256
    cen = torch.zeros((len(fed), C.EMBEDDING_MAX_LENGTH[dataset]), device=dev)
    for i, src in enumerate(fed):        # centroid = mean of the source's first 100k rows, straight from the generator
        n = min(src.rows, 100_000)
        tmp = torch.zeros((n, src.dim), dtype=torch.float16, device=dev)
        W.fill_half(src, P.RowSlice(src.sid, 0, n), tmp)
        cen[i, : src.dim] = tmp.float().mean(0)
        del tmp
    router = W.router_for(dataset, fed, cen.cpu().numpy(), dev)
    emb = W.query_embeddings(fed, B, dev)
    xq, xq_models = W.queries_by_source(fed, emb), W.pack_router_input(dataset, fed, emb, dev)

    results = {}

    def run(plan):
        out = []
        results[len(plan.ranks)] = res_ranks = []
        for r in range(len(plan.ranks)):
            pipe = RetrievalPipeline.from_placement(plan, r, fill_half=W.fill_half, router=router, device=dev)
            step = lambda: pipe.search(xq, k, xq_models=xq_models)   # noqa: E731
            for _ in range(3):
                D, I = step()
            res_ranks.append((D.clone(), I.clone()))
            ms = timed_events(step, batches)
            out.append({"rank": r, "measured_ms": round(ms[len(ms) // 2], 4), "p90_ms": round(ms[(len(ms) * 9) // 10], 4),
                        "predicted_ms": round(plan.predicted_ms[r], 4),
                        "corpus_GB": round(sum(sl.n_rows * plan.sources[sl.sid].row_bytes for u in plan.ranks[r] for sl in u.slices) / 1e9, 3),
                        "units": plan.describe()[r]["units"]})
            del pipe, step
            torch.cuda.empty_cache()
        return out

    plan = P.whole_source_plan(fed, G) if whole else P.plan(fed, G)
    ranks = run(plan)
    meas = [r["measured_ms"] for r in ranks]
    # the G-rank merge (every rank runs it after the exchange): G x slots x k candidates per query, read in place
    buf, _, _ = alloc_packed(B, k, dev, plan.slots)
    gathered = buf[None].expand(G, -1).contiguous()
    mm = timed_events(lambda: merge_gathered(gathered, B, k, plan.slots, k, True), 20)
    res = {"config": f"{dataset}: {len(fed)} sources at their real shapes, {'source s -> GPU s mod G' if whole else 'balanced row slices (placement.plan)'} "
                     f"for G={G}, every rank's step (router + local scans + 1-rank merge) timed in turn on ONE MI355X, B={B}, k={k}",
           "G": G, "exchange_slots": plan.slots, "max_ms": max(meas), "mean_ms": round(sum(meas) / G, 4), "max_over_mean": round(max(meas) / (sum(meas) / G), 4),
           "predicted_max_ms": round(max(plan.predicted_ms), 4), "merge_of_G_ranks_ms": round(mm[len(mm) // 2], 4),
           "exchange_bytes_per_rank": plan.slots * B * k * 12, "ranks": ranks}
    if with_one:
        one = run(P.plan(fed, 1))[0]
        # the G ranks' local results merged (rerank.merge_topk = what the exchange + merge do) against the one-GPU search of the
        # same federation: the row slices must not change a single id or score
        from ragroute_amd.rerank import merge_topk
        Dm, Im = merge_topk(torch.cat([d for d, _ in results[G]], 1), torch.cat([i for _, i in results[G]], 1), k, True)
        D1, I1 = results[1][0]
        res["merged_G_ranks_equal_one_gpu"] = {"ids": bool(torch.equal(Im, I1)), "scores": bool(torch.equal(Dm, D1)),
                                               "queries": int(I1.shape[0]), "k": k, "ids_checksum": int(I1.sum())}
        res["one_gpu_ms"] = one["measured_ms"]
        res["one_gpu_units"] = len(one["units"])
        res["predicted_speedup_at_G"] = round(one["measured_ms"] / (max(meas) + res["merge_of_G_ranks_ms"]), 3)
    print(json.dumps(res))


def main():
    dataset = sys.argv[1] if len(sys.argv) > 1 else "medrag"
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mode = sys.argv[3] if len(sys.argv) > 3 else "segments"
    if "--plan" in sys.argv:
        return plan_mode(dataset, batches, int(sys.argv[sys.argv.index("--plan") + 1]), "--whole" in sys.argv, "--with-one" in sys.argv)
    dev = torch.device("cuda:0")
    sources = C.DATA_SOURCES[dataset]
    model_of = {s: C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][s][0] for s in sources}
    models = sorted(set(model_of.values()))
    d_max, k, B = C.EMBEDDING_MAX_LENGTH[dataset], C.K[dataset], 256
    shards, cents, total_bytes, units = [None] * len(sources), [None] * len(sources), 0, []
    by_model = {}
    for i, s in enumerate(sources):
        by_model.setdefault(model_of[s], []).append(i)
    for m, members in by_model.items():
        d = WIDTH[m]
        if mode == "segments" and len(members) > 1:   # one matrix for the sources of this encoder, filled in place
            seg = SegmentedIndex(d, [ROWS[dataset][sources[i]] for i in members], id_offsets=[i << 40 for i in members], mask_cols=members, device=dev)
            for j, i in enumerate(members):
                make(ROWS[dataset][sources[i]], d, seg.dim, 1234 + i, dev, out=seg.rows_of(j))
                shards[i] = seg.source(j)
            units.append((seg, members))
        else:
            for i in members:
                idx = FlatIndex(d, device=dev)
                idx.adopt(make(ROWS[dataset][sources[i]], d, idx.dim, 1234 + i, dev))
                shards[i] = idx
                units.append((idx, [i]))
    units.sort(key=lambda u: u[1][0])
    for i, s in enumerate(sources):
        idx, d = shards[i], WIDTH[model_of[s]]
        total_bytes += idx.ntotal * idx.dim * 2
        c = np.zeros(d_max, np.float32)
        c[:d] = idx.centroid().cpu().numpy()[:d]
        cents[i] = c
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    emb = {}
    for m in models:
        x = torch.randn((B, WIDTH[m]), generator=g, device=dev)
        emb[m] = x / x.norm(dim=1, keepdim=True)
    xq_models = torch.zeros((B, len(models), d_max), device=dev)
    for j, m in enumerate(models):
        xq_models[:, j, : WIDTH[m]] = emb[m]
    onehot = [C.FEB4RAG_SOURCE_TO_ID[s] for s in sources] if dataset == "feb4rag" else [C.MEDRAG_SOURCE_TO_ID[s] for s in sources]
    net = CorpusRoutingNN(C.ROUTER_INPUT_DIMENSION[dataset], seed=0)
    router = FoldedRouter.fold(net.state_dict(), np.stack(cents), onehot, len(sources), d_max, [models.index(model_of[s]) for s in sources],
                               C.ROUTER_THRESHOLD[dataset], device=dev)
    pipe = RetrievalPipeline(shards, list(range(len(sources))), router=router, units=units)
    xq = {i: emb[model_of[s]] for i, s in enumerate(sources)}

    def step():
        return pipe.search(xq, k, xq_models=xq_models)

    def timed(fn, n):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in evs)

    for _ in range(3):
        D, I = step()
    ms = timed(step, batches)
    med = ms[len(ms) // 2]
    _, mask = router.run(xq_models)
    per_source = {}
    for i, s in enumerate(sources):
        q = shards[i].prepare_queries(xq[i])
        t = timed(lambda: shards[i].search_prepared(q, k), 5)
        per_source[s] = {"rows": shards[i].ntotal, "dim": shards[i].dim, "ms": round(t[2], 4),
                         "frac_of_8TBps": round(shards[i].ntotal * shards[i].dim * 2 / (t[2] * 1e-3) / 8e12, 4)}
    print(json.dumps({
        "config": f"{dataset}: {len(sources)} sources at their real row counts and encoder widths on 1 GPU, B={B}, k={k}, router + "
                  + ("one segmented search per encoder group" if mode == "segments" else "per-source exact top-k") + " + merge",
        "search_units": [{"sources": [sources[i] for i in m], "kind": "segments" if isinstance(u, SegmentedIndex) else "source"} for u, m in units],
        "rows_total": int(sum(ROWS[dataset].values())), "corpus_GB": round(total_bytes / 1e9, 2), "median_ms_per_batch": round(med, 3),
        "p10_ms": round(ms[len(ms) // 10], 3), "p90_ms": round(ms[(len(ms) * 9) // 10], 3), "queries_per_s": round(B / med * 1e3, 1),
        "corpus_GBps": round(total_bytes / 1e9 / (med * 1e-3), 1), "frac_of_8TBps": round(total_bytes / (med * 1e-3) / 8e12, 4),
        "sources_selected_per_query": round(float(mask.float().sum(1).mean()), 2),
        "per_source_alone": per_source}))


if __name__ == "__main__":
    main()
