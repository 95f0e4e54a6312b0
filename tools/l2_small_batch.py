"""Timing of small-batch searches by metric and width (which kernel serves them differs: the L2 metric at 768 < d <= 1536 takes the
wide-row kernel, the inner product the half-resident one)."""
import sys, torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for d, n in ((1024, 4_000_000), (1536, 2_000_000), (2048, 2_000_000), (768, 4_000_000)):
    for metric in ("ip", "l2"):
        idx = FlatIndex(d, metric, "fp16")
        xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
        for s in range(0, n, 1 << 20):
            e = min(n, s + (1 << 20))
            xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).half()
        idx.adopt(xb)
        for nq in (1, 64):
            xq = (torch.randn((nq, idx.dim), generator=g, device=dev) / d ** 0.5).half()
            idx.search_prepared(xq, 32); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): idx.search_prepared(xq, 32)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            print(f"d={d} n={n} {metric} nq={nq}: {ms:.3f} ms  {n*idx.dim*2/ms/1e9:.2f} TB/s", flush=True)
        del idx, xb
