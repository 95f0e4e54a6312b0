#!/usr/bin/env python3
"""bench.py — headline benchmark of the retrieval hot path on MI355X.

Metric (BASELINE.json): queries/sec for route + top-k (+ merge), d=768 fp16, 10M docs per shard, k=32,
query batch 256.  One "step" = one pass of the hot path over one batch of 256 synthetic queries:
router MLP forward (K3) -> query conversion (K0) -> fused similarity scan + top-k over this rank's
HBM-resident shard(s) (K1/K2, route mask folded in) -> [N>1: ONE RCCL all_gather of the packed (score,id)
candidates] -> cross-shard merge (K4).  One process per GPU.

  --scaling weak   (default) every rank holds its own 10M-row shard and every query is answered against all N
                   shards: the whole job performs N x 256 query-shard searches per step, `value` = N*256*K/t
                   = (query x shard) PAIRS per second (at N=1 exactly queries/sec on one 10M shard — the BASELINE
                   configuration).  By construction `value` grows ~N x while the exchange stays cheap; the number that
                   does not is `federation_queries_per_sec` = 256*K/t, queries answered per second against the whole
                   N-shard federation, printed beside it.
  --scaling strong SURVEY.md §8e (the >= 6x target): the federation is fixed at --total-shards (8) x 10M rows; rank r
                   holds shards r, r+N, ...; `value` = `federation_queries_per_sec` = 256*K/t.
  --workload feb4rag|medrag   BASELINE configs 4 / 3 at the federations' REAL shapes (13 BEIR corpora 768 / 1024 / 4096 wide, k = 10;
                   MedRAG's 4 corpora, k = 32; synthetic embeddings, tools/workloads.py) placed on the N GPUs by
                   ragroute_amd/placement.py: balanced row slices, the pieces of one encoder on a rank searched in one pass.
                   Fixed federation = strong scaling; `value` = queries/sec against the whole federation.
Every N > 1 line carries `per_rank_scan_ms` / `per_rank_local_ms` (each rank's scan-launch time and its whole local work per step,
one small all_gather after the timed region): a slow rank is visible.
Every line also carries `exchange_ms` / `merge_ms` (HIP events around the all_gather and the merge, mean per step on rank 0)
and `rccl_ranks` (dist.get_world_size() after init_process_group; 1 without a process group).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts the N ranks itself as child processes (the parent never touches the GPU).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16
METRIC = "queries/sec route+top-k, d=768 fp16, 10M docs/shard, k=32; top-k recall vs CPU"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--workload", default="headline", choices=["headline", "feb4rag", "medrag"],
                    help="headline: equal synthetic 10M x 768 shards (BASELINE metric); feb4rag / medrag: the federation at its real shapes, sliced over the GPUs")
    ap.add_argument("--placement", default="sliced", choices=["sliced", "whole"], help="--workload feb4rag|medrag: balanced row slices, or source s -> GPU s mod G")
    ap.add_argument("--workload-scale", type=int, default=1, help="--workload: divide every source's row count by this (tests / rehearsals; 1 = the real shapes)")
    ap.add_argument("--total-shards", type=int, default=8, help="--scaling strong: shards of the fixed federation")
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per shard")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=None, help="default: 32 (headline), the dataset's K (config.py:97-101) for --workload")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="extra steady-state block after the timed steps (>= this long and >= 500 batches); 0 disables it")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks as CHILD processes (this parent makes no GPU call, not even
    torch.cuda.is_available()), relay rank 0's JSON line, fail if it does not report n_gpus == N."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (exit code {proc.returncode}, "
                         f"{'no result line' if line is None else 'result line printed'})")
    got = json.loads(line).get("n_gpus")
    if got != args.gpus:
        print(line)
        raise SystemExit(f"bench.py: asked for {args.gpus} GPUs but the result line reports n_gpus={got}")
    print(line, flush=True)


def make_shard(torch, n, d, dim, dtype, seed, dev):
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


def percentile(sorted_vals, p):
    if not sorted_vals:
        return None
    i = p * (len(sorted_vals) - 1)
    lo = int(i)
    hi = min(lo + 1, len(sorted_vals) - 1)
    return sorted_vals[lo] + (sorted_vals[hi] - sorted_vals[lo]) * (i - lo)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
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
    from ragroute_amd.pipeline import RetrievalPipeline
    from ragroute_amd.router import CorpusRoutingNN, FoldedRouter

    fed_mode = args.workload != "headline"
    B = args.batch
    if fed_mode:
        # BASELINE configs 3 / 4 at the federation's real shapes, placed on the `world` GPUs by the planner
        from ragroute_amd import config as RC
        from ragroute_amd import placement as PL
        from tools import workloads as W
        k = args.k if args.k is not None else RC.K[args.workload]
        fed = PL.federation(args.workload, rows=None if args.workload_scale == 1 else
                            {name: max(64, rows // args.workload_scale) for name, rows in PL.ROWS[args.workload].items()})
        plan = PL.plan(fed, world) if args.placement == "sliced" else PL.whole_source_plan(fed, world)
        pipe = RetrievalPipeline.from_placement(plan, rank, fill_half=W.fill_half, device=dev)
        cen = W.local_centroids(args.workload, fed, pipe, dev)
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(cen)
            else:
                c = cen.cpu()
                dist.all_reduce(c)
                cen = c.to(dev)
        pipe.router = W.router_for(args.workload, fed, cen.cpu().numpy(), dev)
        emb = W.query_embeddings(fed, B, dev)
        xq, xq_router = W.queries_by_source(fed, emb), W.pack_router_input(args.workload, fed, emb, dev)
        my_slices = [sl for u in plan.ranks[rank] for sl in u.slices]
        n_units = max(1, len(pipe.units))
        # SURVEY §8(d) per unit: rows x padded width x 2 + the unit's query block and result
        alg_bytes = sum(sl.n_rows * plan.sources[sl.sid].row_bytes for sl in my_slices) + \
            sum(B * u[1].dim * 2 + B * k * 12 for u in pipe.units)
        flops = sum(2.0 * B * sl.n_rows * plan.sources[sl.sid].row_bytes / 2 for sl in my_slices)
        first_dim = pipe.units[0][1].dim if pipe.units else 0
        strong, C = True, len(fed)
    else:
        d, k, n = args.dim, args.k if args.k is not None else 32, args.rows
        tdt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
        strong = args.scaling == "strong"
        C = args.total_shards if strong else world                     # sources of the federation = router outputs
        my_shards = list(range(rank, C, world)) if strong else [rank]  # shard s -> GPU s mod G
        slots = -(-C // world)
        shards = []
        for sid in my_shards:
            idx = FlatIndex(d, metric="ip", dtype=args.dtype, device=dev)
            idx.adopt(make_shard(torch, n, d, idx.dim, tdt, 1234 + sid, dev))
            shards.append(idx)

        g = torch.Generator(device=dev)
        g.manual_seed(4321)  # same queries on every rank
        xq = torch.randn((B, d), generator=g, device=dev)
        xq /= xq.norm(dim=1, keepdim=True)

        # router: CorpusRoutingNN over the C sources, default init seed 0, centroid = mean of each shard's first 100k rows,
        # identity scaler; folded into the fused kernel's weights
        cen_mine = torch.zeros((slots, d), dtype=torch.float32, device=dev)
        for j, idx in enumerate(shards):
            cen_mine[j] = idx.xb[: min(n, 100_000), :d].float().mean(0)
        if world > 1:
            on_dev = backend == "nccl"
            cens = torch.empty(world * slots * d, dtype=torch.float32, device=dev if on_dev else "cpu")
            dist.all_gather_into_tensor(cens, cen_mine.view(-1) if on_dev else cen_mine.cpu().view(-1))
            cens = cens.view(world, slots, d).cpu().numpy()
        else:
            cens = cen_mine[None].cpu().numpy()
        cen_all = np.stack([cens[s % world, s // world] for s in range(C)]) if strong else np.stack([cens[r, 0] for r in range(world)])
        net = CorpusRoutingNN(2 * d + C, seed=0)
        router = FoldedRouter.fold(net.state_dict(), cen_all, list(range(C)), C, d, [0] * C, 0.5, device=dev)
        xq_router = xq[:, None, :].contiguous()
        pipe = RetrievalPipeline(shards, my_shards, router=router, slots=slots)
        n_units = len(shards)
        alg_bytes = len(shards) * (n * d * 2 + B * d * 2 + B * k * 12)  # SURVEY §8(d), per step on this rank
        flops = 2.0 * B * n * d * len(shards)
        first_dim = shards[0].dim

    def step():  # K3 router -> K0 convert -> K1/K2 scan+top-k (route mask folded in) -> all_gather (N>1) -> K4 merge
        return pipe.search(xq, k, xq_models=xq_router)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n_steps, with_events):
        """n_steps back-to-back steps: wall clock between two fences (max over ranks), live HIP-event time of every scan
        launch (rr_profile_*), and optionally one HIP-event pair per step on the launch stream."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)] if with_events else []
        fence()
        # scan launches per shard and step: bootstrap + up to 8 chunks (capi.hip chunk_schedule), per 256-query block;
        # rr_profile_end FAILS if a launch went unrecorded, so the roofline cannot be overstated by a short buffer
        check(lib().rr_profile_begin(n_steps * 12 * n_units * (-(-B // 256)) + 16), "rr_profile_begin")
        t0 = time.perf_counter()
        for i in range(n_steps):
            if with_events:
                evs[i][0].record()
            step()
            if with_events:
                evs[i][1].record()
        fence()
        elapsed = time.perf_counter() - t0
        scan_ms, n_launch, rows_scanned = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        check(lib().rr_profile_end(ctypes.byref(scan_ms), ctypes.byref(n_launch), ctypes.byref(rows_scanned)), "rr_profile_end")
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per_step = sorted(a.elapsed_time(b) for a, b in evs)
        return elapsed, scan_ms.value, n_launch.value, per_step

    for _ in range(args.warmup):
        step()
    elapsed, scan_ms, n_launch, per_step = timed(args.steps, True)
    # exchange / merge time per step: a few extra steps AFTER the timed region (the three event records per step stay out of `value`)
    pipe.time_stages(True)
    for _ in range(min(10, max(3, args.steps))):
        D_chk, I_chk = step()
    # checksums of the answer (the same on every rank), taken AFTER the timed region (their host reads idle the GPU for a moment,
    # and the steps right after such a pause run slower): for --workload the corpus is a pure function of (source, row), so the
    # line of ANY placement / GPU count must carry the same three numbers
    wts = 1 + torch.arange(I_chk.shape[1], device=dev, dtype=torch.int64)
    checksum = {"ids_sum": int(I_chk.sum()), "ids_rank_weighted": int((I_chk % 1000003 * wts).sum()),
                "score_sum": round(float(D_chk.double().masked_fill(~torch.isfinite(D_chk), 0.0).sum()), 6)}   # (padding scores are -inf)
    del D_chk, I_chk
    exchange_ms, merge_ms = pipe.stage_ms()
    local_ms = pipe.local_ms()
    pipe.time_stages(False)
    fence()
    # every rank's scan-launch time per step (timed region) and whole local work per step (the extra steps): a slow rank shows
    mine = torch.tensor([scan_ms / args.steps, local_ms, alg_bytes / 1e9], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        per_rank = torch.empty(world * 3, dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(per_rank, mine)
        per_rank = per_rank.view(world, 3).cpu().tolist()
    else:
        per_rank = [mine.cpu().tolist()]
    rccl_ranks = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1

    sustained = None
    if args.sustained_seconds > 0:
        est = elapsed / args.steps
        n_sus = max(100 if fed_mode else 500, int(args.sustained_seconds / est) + 1)
        s_elapsed, s_scan_ms, s_launch, _ = timed(n_sus, False)
        sustained = (n_sus, s_elapsed, s_scan_ms, s_launch)

    if rank == 0:
        K = args.steps
        units = B if strong else world * B                 # queries per step the whole job answers (weak: query x shard)
        launches_per_step = n_launch / K
        achieved = alg_bytes * K / (scan_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not fed_mode:
            tj = json.load(open(tpath))
            if (tj.get("rows") == n and tj.get("dim") == d and tj.get("batch") == B and tj.get("dtype", "fp16") == args.dtype
                    and tj.get("k", 32) == k and tj.get("lib_version", lib().rr_version()) == lib().rr_version()):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not measured in this run)"
        if fed_mode:
            total_rows = sum(s_.rows for s_ in fed)
            total_gb = sum(s_.rows * s_.row_bytes for s_ in fed) / 1e9
            config = {"workload": f"{args.workload}: {C} sources at {'their real row counts' if args.workload_scale == 1 else f'1/{args.workload_scale} of their real row counts'} and encoder widths ({total_rows} rows, {total_gb:.1f} GB fp16), "
                                  f"{'balanced row slices (placement.plan)' if args.placement == 'sliced' else 'source s -> GPU s mod G'} over {world} GPU(s), "
                                  f"query batch {B}, k={k}, router MLP + per-unit exact top-k + ONE all_gather + merge",
                      "dataset": args.workload, "row_scale": f"1/{args.workload_scale}", "placement": args.placement, "rows_total": total_rows, "corpus_GB": round(total_gb, 2), "batch": B, "k": k,
                      "sources": C, "units_per_rank": [len(u) for u in plan.ranks], "exchange_slots": pipe.slots,
                      "predicted_ms_per_rank": [round(t, 3) for t in plan.predicted_ms],
                      "parallelism": f"{C} sources in {sum(len(u.slices) for us in plan.ranks for u in us)} row slices over {world} gpu(s)",
                      "unit_definition": "queries per second against the whole fixed federation (= federation_queries_per_sec)",
                      "timing": "ms_per_step = wall clock of the K steps between fences / K; median/p10/p90 = one HIP-event pair per step"}
        else:
            S = len(shards)
            shape = f"{n} x {d} {args.dtype} rows per shard"
            layout = (f"{C} shards fixed, {S} per GPU (strong scaling)" if strong else "one shard per GPU (weak scaling)")
            config = {"workload": f"{shape}, {layout}, query batch {B}, k={k}, exact inner-product top-k + router MLP over "
                                  f"{C} source(s) + cross-shard merge",
                      "rows_per_shard": n, "dim": d, "batch": B, "k": k, "shards_total": C, "shards_per_gpu": S,
                      "parallelism": f"shard-per-gpu x{world}" if not strong else f"{C} shards over {world} gpu(s)",
                      "unit_definition": ("queries per second against the whole fixed federation (= federation_queries_per_sec)" if strong else
                                          f"(query x {n}-row shard) PAIRS searched per second, whole job = N x federation_queries_per_sec "
                                          "(N=1: queries/sec on one shard, the BASELINE configuration); federation_queries_per_sec = "
                                          "queries answered per second against all N shards"),
                      "timing": "ms_per_step = wall clock of the K steps between fences / K; median/p10/p90 = one HIP-event pair per step"}
        scans = [r[0] for r in per_rank]
        locals_ = [r[1] for r in per_rank]
        res = {
            "metric": METRIC, "value": round(units * K / elapsed, 1), "unit": "queries/sec", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if fed_mode else args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "median_ms": round(percentile(per_step, 0.5), 4), "p10_ms": round(percentile(per_step, 0.1), 4),
            "p90_ms": round(percentile(per_step, 0.9), 4),
            "federation_queries_per_sec": round(B * K / elapsed, 1), "exchange_ms": round(exchange_ms, 4), "merge_ms": round(merge_ms, 4),
            "per_rank_scan_ms": {"min": round(min(scans), 4), "max": round(max(scans), 4), "mean": round(sum(scans) / len(scans), 4),
                                 "ranks": [round(v, 4) for v in scans]},
            "per_rank_local_ms": {"min": round(min(locals_), 4), "max": round(max(locals_), 4), "mean": round(sum(locals_) / len(locals_), 4),
                                  "max_over_mean": round(max(locals_) / (sum(locals_) / len(locals_)), 4), "ranks": [round(v, 4) for v in locals_],
                                  "what": "router + query conversion + every local scan, HIP events up to the exchange, mean of the steps after the timed region"},
            "per_rank_corpus_GB": [round(r[2], 3) for r in per_rank], "result_checksum": checksum,
            "rccl_ranks": rccl_ranks, "exchange_backend": (backend if world > 1 else "none (one rank: the merge reads the local buffer)"),
            "config": config,
            "roofline": {"bound": "hbm", "kernel": lib().rr_flat_scan_kernel_name(first_dim, B).decode() if first_dim else None, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": round(alg_bytes / launches_per_step),
                         "avg_launch_ms": round(scan_ms / max(1, n_launch), 4), "launches_per_step": round(launches_per_step, 2),
                         "mfma_tflops": round(flops * K / (scan_ms * 1e-3) / 1e12, 1),
                         "mfma_frac": round(flops * K / (scan_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                         "scan_share_of_step": round(scan_ms / (elapsed * 1e3), 4),
                         "end_to_end_frac": round(alg_bytes * K / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if sustained is not None:
            n_sus, s_elapsed, s_scan_ms, s_launch = sustained
            res["roofline"]["sustained"] = {
                "batches": n_sus, "seconds": round(s_elapsed, 3), "ms_per_step": round(s_elapsed / n_sus * 1e3, 4),
                "value": round(units * n_sus / s_elapsed, 1), "avg_launch_ms": round(s_scan_ms / max(1, s_launch), 4),
                "achieved": round(alg_bytes * n_sus / (s_scan_ms * 1e-3) / 1e9, 1),
                "frac": round(alg_bytes * n_sus / (s_scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and not args.no_cpu_baseline and not fed_mode:
            res["cpu_baseline"] = cpu_baseline(shards[0], xq, n, d, k)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(idx, xq, n, d, k):
    """The reference's CPU call pattern timed on this box's host cores: ONE f32 query per index.search call
    (data_source.py:113-114,186) over a flat f32 index — the oracle's C restatement ("port"), on all usable cores and on
    one thread.  Bounded sample: the shard's first 1M rows, single-query calls for ~10 s (all cores) + ~5 s (one thread);
    the rate is scaled linearly in rows to the full shard.  If `import faiss` works on this box, IndexFlatIP.search is
    timed beside it and compared with the HIP path (the only route to pinning a2/a3)."""
    import numpy as np
    from oracle import oracle as O
    cores = max(1, int(os.environ.get("RR_CPU_BASELINE_THREADS", usable_cpus())))
    sample = min(n, 1_000_000)
    xb = idx.xb[:sample, :d].float().cpu().numpy()
    q = xq.cpu().numpy().astype(np.float32)

    def rate(threads, budget_s, max_calls):
        O.set_threads(threads)
        O.flat_search_ip_single(xb, q[0], k)  # warm-up (page-in, thread pool)
        t0, calls = time.perf_counter(), 0
        while calls < max_calls and (calls < 4 or time.perf_counter() - t0 < budget_s):
            O.flat_search_ip_single(xb, q[calls % len(q)], k)
            calls += 1
        dt = time.perf_counter() - t0
        return calls, dt, O.num_threads()

    calls, dt, used = rate(cores, 10.0, 256)
    calls1, dt1, _ = rate(1, 5.0, 64)
    O.set_threads(cores)
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
    out = {"value": round(qps_sample * sample / n, 3), "unit": "queries/sec", "cores": used, "kind": "port",
           "sample": f"{calls} single-query f32 searches (nq=1 per call, as the reference issues them) over the first {sample} rows "
                     f"in {dt:.2f} s = {qps_sample:.2f} q/s on the sample, scaled x{sample / n:.3g} to {n} rows; "
                     f"{used} OpenMP threads = every CPU this process may use",
           "one_thread": {"value": round(calls1 / dt1 * sample / n, 4), "cores": 1,
                          "sample": f"{calls1} searches in {dt1:.2f} s on the same sample, same scaling"},
           "parity_vs_cpu": parity, "other_rows": extra}
    try:
        out.update(faiss_probe(xb, q, qh, Dg, Ig, k, sample, n))
    except Exception as e:  # this leg has never met a real faiss (none in the image): it must not cost the bench line
        out.update({"faiss": "probe failed", "faiss_detail": f"{type(e).__name__}: {e}"[:200]})
    return out


def faiss_probe(xb, q, qh, Dg, Ig, k, sample, n):
    """BASELINE.md row C5 / SURVEY §8d: if faiss happens to be installed on this box, time IndexFlatIP.search at nq = 1 on the
    same sample and compare its results with the HIP path's (parity_vs_faiss); otherwise say so.  Nothing is installed."""
    import numpy as np
    try:
        import faiss  # noqa: F401
    except Exception as e:  # ImportError, or a broken wheel
        return {"faiss": "unavailable", "faiss_detail": f"import faiss: {type(e).__name__}"}
    index = faiss.IndexFlatIP(xb.shape[1])
    index.add(xb)
    index.search(q[:1], k)
    t0, calls = time.perf_counter(), 0
    while calls < 256 and (calls < 8 or time.perf_counter() - t0 < 5.0):
        index.search(q[calls % len(q)][None, :], k)
        calls += 1
    dt = time.perf_counter() - t0
    Df, If = index.search(np.ascontiguousarray(qh), k)
    gaps_ok = np.ones_like(If, dtype=bool)
    gaps_ok[:, 1:] &= (Df[:, :-1] - Df[:, 1:]) > 2e-3
    gaps_ok[:, :-1] &= (Df[:, :-1] - Df[:, 1:]) > 2e-3
    # the tie rule (include/ragroute_hip.h: equal scores by ascending row id) against FAISS on integer-valued rows, where every
    # score is exact and ties are plentiful: does FAISS return the same id SET, and the same ORDER inside runs of equal scores?
    from ragroute_amd.flat_index import FlatIndex
    rng = np.random.default_rng(11)
    xt = rng.integers(-1, 2, size=(30_000, xb.shape[1])).astype(np.float32)
    qt = rng.integers(-1, 2, size=(32, xb.shape[1])).astype(np.float32)
    it = faiss.IndexFlatIP(xb.shape[1])
    it.add(xt)
    Dt, It = it.search(qt, k)
    ours = FlatIndex(xb.shape[1])
    ours.add(xt)
    Do, Io = ours.search(qt, k)
    ties = {"scores_identical": bool(np.array_equal(Do, Dt)),
            "ties_set_identical": bool(all(set(a.tolist()) == set(b.tolist()) for a, b in zip(Io, It))),
            "ties_order_identical": bool(np.array_equal(Io, It)),
            "queries_with_a_tie_in_the_result": int(sum(len(set(r.tolist())) < len(r) for r in Dt)),
            "sample": "30000 x d rows and 32 queries with entries in {-1, 0, 1}, exact scores, k as the bench"}
    x = np.ascontiguousarray(q[:64].copy())
    y = x.copy()
    faiss.normalize_L2(x)
    from ragroute_amd.flat_index import normalize_L2
    normalize_L2(y)
    return {"faiss": getattr(faiss, "__version__", "present"),
            "C5_faiss_IndexFlatIP_nq1": {"value": round(calls / dt * sample / n, 3), "unit": "queries/sec",
                                         "sample": f"{calls} nq=1 searches over {sample} rows in {dt:.2f} s, scaled x{sample / n:.3g}",
                                         "threads": faiss.omp_get_max_threads()},
            "parity_vs_faiss": {"ids_identical_where_gap_gt_2e-3": bool(np.array_equal(Ig[gaps_ok], If[gaps_ok])),
                                "positions_compared": int(gaps_ok.sum()), "max_abs_score_diff": float(np.abs(Dg - Df).max()),
                                "score_tolerance": 1e-3, "normalize_L2_max_abs_diff": float(np.abs(x - y).max()),
                                "ties": ties}}


def cpu_extra_rows(O, xb, q, k, sample, n):
    """BASELINE.md rows C2-C4 on the same host cores (context only): batched sgemm + top-k, router forward, merge."""
    import numpy as np
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
