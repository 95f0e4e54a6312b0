"""Timing of the squared-L2 search (d = 768 and a wide-row case)."""
import sys, torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for d, n in ((768, 10_000_000), (1024, 4_000_000)):
    for metric in ("ip", "l2"):
        idx = FlatIndex(d, metric, "fp16")
        xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
        for s in range(0, n, 1 << 20):
            e = min(n, s + (1 << 20))
            xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).half()
        idx.adopt(xb)
        xq = (torch.randn((256, idx.dim), generator=g, device=dev) / d ** 0.5).half()
        idx.search_prepared(xq, 32); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): idx.search_prepared(xq, 32)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"d={d} n={n} {metric}: {ms:.3f} ms  {n*idx.dim*2/ms/1e9:.2f} TB/s", flush=True)
        del idx, xb
