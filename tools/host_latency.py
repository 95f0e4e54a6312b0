"""Host-visible latency of the faiss-shaped call (numpy in, numpy out; the reference's call shape data_source.py:114, 158):
FlatIndex.search(xq float32 [nq, d], k) including host->device, query conversion, search, device->host and the sync."""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from ragroute_amd.flat_index import FlatIndex

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
out = {}
for n, d in ((100_000, 768), (1_000_000, 768), (10_000_000, 768), (171_332, 4096)):
    idx = FlatIndex(d, device=dev)
    xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).half()
    idx.adopt(xb)
    for nq in (1, 256):
        xq = np.random.default_rng(0).standard_normal((nq, d)).astype(np.float32)
        for _ in range(5):
            idx.search(xq, 32)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            D, I = idx.search(xq, 32)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        xh = idx.prepare_queries(xq)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            idx.search_prepared(xh, 32)
        b.record()
        torch.cuda.synchronize()
        out[f"{n}x{d} nq={nq}"] = {"host_call_ms_median": round(ts[25], 4), "host_call_ms_p10": round(ts[5], 4),
                                   "device_only_ms": round(a.elapsed_time(b) / 20, 4)}
        print(f"{n}x{d} nq={nq}", out[f"{n}x{d} nq={nq}"], flush=True)
    del idx, xb
print(json.dumps(out))
