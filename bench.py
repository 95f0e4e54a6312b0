#!/usr/bin/env python3
"""bench.py — headline benchmark of the retrieval hot path on MI355X.

Metric (BASELINE.json): queries/sec for route + top-k (+ merge), d=768 fp16, 10M docs per shard, k=32,
query batch 256.  One "step" = one pass of the hot path over one batch of 256 synthetic queries:
router MLP forward (K3) -> query conversion (K0) -> fused similarity scan + top-k over this rank's
HBM-resident 10M x 768 shard (K1/K2) -> route mask -> [N>1: RCCL all_gather of the (score,id) candidates]
-> cross-shard merge (K4).  One process per GPU; weak scaling: every rank holds its own 10M-row shard and
every query is answered against all N shards, so the whole job performs N x 256 query-shard searches per
step and `value` = N * 256 * K / t  (at N=1 exactly queries/sec on one 10M shard).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16


def make_shard(n, d, dim, dtype, seed, dev):
    """i.i.d. N(0,1) rows, L2-normalised, cast to the storage dtype — generated on device in chunks."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    xb = torch.zeros((n, dim), dtype=dtype, device=dev)
    step = 1 << 20
    for s in range(0, n, step):
        e = min(n, s + step)
        x = torch.randn((e - s, d), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        xb[s:e, :d] = x.to(dtype)
    return xb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per shard (per GPU)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    backend = os.environ.get("RR_BENCH_BACKEND", "nccl")  # "gloo" + RR_BENCH_ONE_DEVICE=1 rehearses N>1 on a 1-GPU box
    if os.environ.get("RR_BENCH_ONE_DEVICE"):
        local = 0
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ragroute_amd._lib import check, lib
    from ragroute_amd.flat_index import FlatIndex
    from ragroute_amd.router import CorpusRoutingNN, FoldedRouter

    d, B, k, n = args.dim, args.batch, args.k, args.rows
    tdt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    idx = FlatIndex(d, metric="ip", dtype=args.dtype, device=dev)
    idx.adopt(make_shard(n, d, idx.dim, tdt, 1234 + rank, dev))

    g = torch.Generator(device=dev)
    g.manual_seed(4321)  # same queries on every rank
    xq = torch.randn((B, d), generator=g, device=dev)
    xq /= xq.norm(dim=1, keepdim=True)

    # router: CorpusRoutingNN over C = N sources (one per GPU), default init seed 0, centroid = mean of the
    # shard's first 100k rows, identity scaler; folded into the fused kernel's weights
    C = world
    cen = idx.xb[: min(n, 100_000), :d].float().mean(0)
    if world > 1:
        on_dev = backend == "nccl"
        cens = torch.empty(world * d, dtype=torch.float32, device=dev if on_dev else "cpu")
        dist.all_gather_into_tensor(cens, cen.contiguous().view(-1) if on_dev else cen.cpu().view(-1))
        cen_all = cens.view(world, d).cpu().numpy()
    else:
        cen_all = cen[None].cpu().numpy()
    net = CorpusRoutingNN(2 * d + C, seed=0)
    router = FoldedRouter.fold(net.state_dict(), cen_all, list(range(C)), C, d, [0] * C, 0.5, device=dev)
    xq_router = xq[:, None, :].contiguous()

    from ragroute_amd.pipeline import RetrievalPipeline
    pipe = RetrievalPipeline([idx], [rank], router=router)

    def step():  # K3 router -> K0 convert -> K1/K2 scan+top-k (route mask folded in) -> all_gather (N>1) -> K4 merge
        return pipe.search(xq, k, xq_models=xq_router)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    check(lib().rr_profile_begin(args.steps * 16), "rr_profile_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    scan_ms, n_launch, rows_scanned = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    check(lib().rr_profile_end(ctypes.byref(scan_ms), ctypes.byref(n_launch), ctypes.byref(rows_scanned)), "rr_profile_end")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        K = args.steps
        ms_per_step = elapsed / K * 1e3
        value = world * B * K / elapsed
        alg_bytes = n * d * 2 + B * d * 2 + B * k * 12      # per batch and shard
        flops = 2.0 * B * n * d
        launches_per_step = n_launch.value / K
        avg_launch_ms = scan_ms.value / max(1, n_launch.value)
        achieved = alg_bytes * K / (scan_ms.value * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("rows") == n and tj.get("dim") == d and tj.get("batch") == B:
                traffic = tj.get("hbm_bytes_per_launch")
        res = {
            "metric": "queries/sec route+top-k, d=768 fp16, 10M docs/shard, k=32; top-k recall vs CPU",
            "value": round(value, 1), "unit": "queries/sec", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{n} x {d} {args.dtype} rows per shard (one shard per GPU), query batch {B}, k={k}, "
                                   f"exact inner-product top-k + router MLP over {C} source(s) + cross-shard merge",
                       "rows_per_shard": n, "dim": d, "batch": B, "k": k, "parallelism": f"shard-per-gpu x{world}",
                       "unit_definition": "query x 10M-row shard searches per second, whole job (N=1: queries/sec on one shard)"},
            "roofline": {"bound": "hbm", "kernel": "flat_scan16_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(alg_bytes / launches_per_step),
                         "avg_launch_ms": round(avg_launch_ms, 4), "launches_per_step": round(launches_per_step, 2),
                         "mfma_tflops": round(flops * K / (scan_ms.value * 1e-3) / 1e12, 1),
                         "mfma_frac": round(flops * K / (scan_ms.value * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                         "scan_share_of_step": round(scan_ms.value / (elapsed * 1e3), 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(idx, xq, n, d, k)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(idx, xq, n, d, k):
    """The reference's CPU call pattern timed on this box's host cores: ONE f32 query per index.search call
    (data_source.py:113-114,186) over a flat f32 index — the oracle's C restatement ("port": faiss is not installed).
    Bounded sample: the shard's first 1M rows, as many single-query calls as fit in ~10 s; the rate is scaled
    linearly in rows to the full shard."""
    from oracle import oracle as O
    # the CPUs this process may really use: affinity mask, capped at the 1-GPU box's CPU share
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    O.set_threads(max(1, min(usable, int(os.environ.get("RR_CPU_BASELINE_THREADS", 16)))))
    sample = min(n, 1_000_000)
    xb = idx.xb[:sample, :d].float().cpu().numpy()
    q = xq.cpu().numpy().astype(np.float32)
    O.flat_search_ip_single(xb, q[0], k)  # warm-up (page-in, thread pool)
    t0, calls = time.perf_counter(), 0
    while calls < 256 and (calls < 8 or time.perf_counter() - t0 < 10.0):
        O.flat_search_ip_single(xb, q[calls % len(q)], k)
        calls += 1
    dt = time.perf_counter() - t0
    qps_sample = calls / dt
    # parity on the same sample: GPU top-k of the first `sample` rows vs the oracle (f64 scores of the same fp16 values)
    from ragroute_amd.flat_index import FlatIndex
    nchk = 32
    sub = FlatIndex(d, dtype=idx.dtype, device=idx.device)
    sub.adopt(idx.xb[:sample])
    Dg, Ig = sub.search_prepared(sub.prepare_queries(xq[:nchk]), k)
    Dg, Ig = Dg.cpu().numpy(), Ig.cpu().numpy()
    qh = sub.prepare_queries(xq[:nchk])[:, :d].float().cpu().numpy()
    Dr, Ir = O.flat_search_ip(xb, qh, k)
    recall = float(np.mean([len(set(Ig[i]) & set(Ir[i])) / k for i in range(nchk)]))
    parity = {"recall_at_k": round(recall, 6), "rank_order_identical_queries": int(sum(np.array_equal(Ig[i], Ir[i]) for i in range(nchk))),
              "queries_checked": nchk, "max_abs_score_diff": float(np.abs(Dg - Dr).max()),
              "sample": f"GPU top-{k} of the first {sample} rows vs oracle.flat_search_ip (f64 dot products of the same fp16-rounded values)"}
    extra = cpu_extra_rows(O, xb, q, k, sample, n)
    return {"value": round(qps_sample * sample / n, 3), "unit": "queries/sec", "cores": O.num_threads(), "kind": "port",
            "sample": f"{calls} single-query f32 searches (nq=1 per call, as the reference issues them) over the first {sample} rows "
                      f"in {dt:.2f} s = {qps_sample:.2f} q/s on the sample, scaled x{sample / n:.3g} to {n} rows",
            "parity_vs_cpu": parity, "other_rows": extra}


def cpu_extra_rows(O, xb, q, k, sample, n):
    """BASELINE.md rows C2-C4 on the same host cores (context only): batched sgemm + top-k, router forward, merge."""
    t0 = time.perf_counter()
    S = q @ xb.T                                             # C2: nq = 256 blocked sgemm (numpy BLAS) + per-query top-k
    idx = np.argpartition(-S, k, axis=1)[:, :k]
    np.take_along_axis(S, idx, axis=1).sort(axis=1)
    c2 = len(q) / (time.perf_counter() - t0) * sample / n
    rng = np.random.default_rng(0)                           # C3: reference-shaped router step per query (router.py:241-283)
    C, d = 4, q.shape[1]
    sd = {"fc1.weight": rng.standard_normal((256, 2 * d + C)).astype(np.float32) * 0.02, "fc1.bias": np.zeros(256, np.float32),
          "ln1.weight": np.ones(256, np.float32), "ln1.bias": np.zeros(256, np.float32),
          "fc2.weight": rng.standard_normal((128, 256)).astype(np.float32) * 0.05, "fc2.bias": np.zeros(128, np.float32),
          "ln2.weight": np.ones(128, np.float32), "ln2.bias": np.zeros(128, np.float32),
          "fc3.weight": rng.standard_normal((1, 128)).astype(np.float32) * 0.1, "fc3.bias": np.zeros(1, np.float32)}
    srcs = [str(i) for i in range(C)]
    cen = {s: rng.standard_normal(d).astype(np.float32) for s in srcs}
    mean, scale = np.zeros(2 * d + C), np.ones(2 * d + C)
    t0 = time.perf_counter()
    for i in range(64):
        lg = O.router_logits("medrag", srcs, {s: "m" for s in srcs}, {s: j for j, s in enumerate(srcs)}, d, {"m": q[i]}, cen, sd, mean, scale)
        O.router_select("medrag", srcs, lg)
    c3 = 64 / (time.perf_counter() - t0)
    cand = rng.standard_normal(4 * k).tolist()               # C4: np.argsort merge of S*k candidates per query (rerank.py:3-9)
    docs = list(range(4 * k))
    t0 = time.perf_counter()
    for _ in range(2000):
        O.rerank_medrag(docs, cand, k)
    c4 = 2000 / (time.perf_counter() - t0)
    return {"C2_batched_sgemm_topk_queries_per_s_scaled": round(c2, 2), "C3_router_numpy_queries_per_s": round(c3, 1),
            "C4_merge_numpy_queries_per_s": round(c4, 1)}


if __name__ == "__main__":
    main()
