"""Development check of the int8 screened search: agreement with the plain search, proof rate, timing."""
import sys, time
import numpy as np
import torch
from ragroute_amd.flat_index import FlatIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, k = 768, 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
idx = FlatIndex(d, "ip", "fp16")
xb = torch.empty((n, d), dtype=torch.float16, device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    t = torch.randn((e - s, d), generator=g, device=dev)
    xb[s:e] = (t / t.norm(dim=1, keepdim=True)).half()
idx.adopt(xb)
q = torch.randn((256, d), generator=g, device=dev)
q = (q / q.norm(dim=1, keepdim=True))
xq = idx.prepare_queries(q)
D0, I0 = idx.search_prepared(xq, k)
torch.cuda.synchronize()
t0 = time.time(); idx.build_screen(); torch.cuda.synchronize(); print("build_screen s", time.time() - t0, "stats", idx._x8_stats[:3].view(torch.float32).tolist())

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for L in (128, 256, 512, 1024):
    D, I, ex = idx.search_screened(xq, k, list_len=L)
    torch.cuda.synchronize()
    same = (I == I0).all(dim=1)
    print(f"L={L}: proven {int(ex.sum())}/256, ids equal {int(same.sum())}/256, proven&wrong {int((ex.bool() & ~same).sum())}, "
          f"max|dD| {float((D - D0).abs().max()):.2e}, ms {timed(lambda: idx.search_screened(xq, k, list_len=L)):.3f}")
x8 = idx._x8; idx._x8 = None
print("plain ms", timed(lambda: idx.search_prepared(xq, k)))
print("plain nq=1 ms", timed(lambda: idx.search_prepared(xq[:1], k)))
idx._x8 = x8
for L in (256, 512):
    print(f"screened nq=1 L={L} ms", timed(lambda: idx.search_screened(xq[:1], k, list_len=L)))
