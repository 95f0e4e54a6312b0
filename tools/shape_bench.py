"""Bench-style JSON line for one flat-search shape (development / profiles; bench.py is the contract for the headline).

    python tools/shape_bench.py ROWS DIM [BATCH] [K] [DTYPE] [ITERS] [ip|l2]

Synthetic N(0,1)/sqrt(d) corpus generated on device, queries resident in HBM; times ITERS back-to-back searches with one
HIP-event pair per search (median / p10 / p90) and the scan launches with the library's own HIP events (rr_profile_*), then
the same ITERS searches between one event pair without any of those events (`back_to_back_ms`: the throughput figure);
`roofline.achieved` = algorithmic bytes (rows x padded dim x 2 + batch x dim x 2 + batch x k x 12) / scan time."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragroute_amd._lib import check, lib
from ragroute_amd.flat_index import FlatIndex


def main():
    n, d = int(sys.argv[1]), int(sys.argv[2])
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    dtype = sys.argv[5] if len(sys.argv) > 5 else "fp16"
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
    metric = sys.argv[7] if len(sys.argv) > 7 else "ip"
    dev = torch.device("cuda:0")
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    idx = FlatIndex(d, metric=metric, dtype=dtype, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    xb = torch.empty((n, idx.dim), dtype=tdt, device=dev)
    for s in range(0, n, 1 << 19):
        e = min(n, s + (1 << 19))
        xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).to(tdt)
    idx.adopt(xb)
    xq = torch.randn((nq, idx.dim), generator=g, device=dev).to(tdt)
    for _ in range(5):
        D, I = idx.search_prepared(xq, k)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    check(lib().rr_profile_begin(iters * 16), "rr_profile_begin")
    for a, b in evs:
        a.record()
        D, I = idx.search_prepared(xq, k)
        b.record()
    torch.cuda.synchronize()
    scan_ms, n_launch, rows = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    check(lib().rr_profile_end(ctypes.byref(scan_ms), ctypes.byref(n_launch), ctypes.byref(rows)), "rr_profile_end")
    per = sorted(a.elapsed_time(b) for a, b in evs)
    med = per[len(per) // 2]
    # throughput form (what bench.py's ms_per_step is): the same searches back to back between ONE event pair, without the
    # per-launch profiling events - the per-search event pairs and the 2 events around every scan launch above cost a
    # sub-millisecond search several gaps of ~5 us each
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        D, I = idx.search_prepared(xq, k)
    t1.record()
    torch.cuda.synchronize()
    b2b = t0.elapsed_time(t1) / iters
    alg = n * idx.dim * 2 + nq * idx.dim * 2 + nq * k * 12
    flops = 2.0 * nq * n * idx.dim
    ach = alg * iters / (scan_ms.value * 1e-3) / 1e9
    S = xq[: min(nq, 4)].float() @ xb[:100000].float().T          # sanity: the top-1 of the first 100k rows
    if metric == "l2":
        S = S - 0.5 * (xb[:100000].float() ** 2).sum(1)[None, :]
    sane = bool(((I[: min(nq, 4), 0] >= 100000) | (I[: min(nq, 4), 0] == S.argmax(1))).all())
    print(json.dumps({
        "workload": f"{n} x {d} {dtype} rows (padded dim {idx.dim}), query batch {nq}, k={k}, exact {'squared-L2' if metric == 'l2' else 'inner-product'} top-k, one shard, 1 GPU",
        "kernel": lib().rr_flat_scan_kernel_name(idx.dim, nq).decode(),
        "median_ms": round(med, 4), "p10_ms": round(per[len(per) // 10], 4), "p90_ms": round(per[(9 * len(per)) // 10], 4),
        "queries_per_s": round(nq / med * 1e3, 1), "end_to_end_GBps": round(alg / med / 1e6, 1),
        "end_to_end_frac_of_8TBps": round(alg / med / 1e6 / 8000, 4),
        "back_to_back_ms": round(b2b, 4), "back_to_back_frac_of_8TBps": round(alg / b2b / 1e6 / 8000, 4),
        "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000, 4),
                     "scan_launches_per_search": n_launch.value / iters, "avg_launch_ms": round(scan_ms.value / max(1, n_launch.value), 4),
                     "algorithmic_bytes_per_search": alg, "mfma_tflops": round(flops * iters / (scan_ms.value * 1e-3) / 1e12, 1)},
        "sanity_top1": sane, "lib_version": lib().rr_version()}), flush=True)


if __name__ == "__main__":
    main()
