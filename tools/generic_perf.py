"""Timing of the wide-row kernels (d > 768): usage generic_perf.py d [n_rows]"""
import sys, torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex
d = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
idx = FlatIndex(d, "ip", "fp16")
xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
for s in range(0, n, 1 << 18):
    e = min(n, s + (1 << 18))
    xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).half()
idx.adopt(xb)
xq = torch.randn((256, idx.dim), generator=g, device=dev).half()
for nq in (256, 1):
    q = xq[:nq].contiguous()
    idx.search_prepared(q, 32); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): D, I = idx.search_prepared(q, 32)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"d={d} n={n} nq={nq}: {ms:.3f} ms  {n*idx.dim*2/ms/1e9:.2f} TB/s", flush=True)
S = xq[:4].float() @ xb[:100000].float().T
D, I = idx.search_prepared(xq[:4].contiguous(), 32)
