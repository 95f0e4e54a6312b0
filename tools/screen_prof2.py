import sys, torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex
n = 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for d in (384, 768):
    idx = FlatIndex(d, "ip", "fp16")
    xb = torch.empty((n, d), dtype=torch.float16, device=dev)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        t = torch.randn((e - s, d), generator=g, device=dev)
        xb[s:e] = (t / t.norm(dim=1, keepdim=True)).half()
    idx.adopt(xb)
    q = torch.randn((256, d), generator=g, device=dev)
    xq = idx.prepare_queries(q / q.norm(dim=1, keepdim=True))
    ms = timed(lambda: idx.search_prepared(xq, 32))
    print(f"fp16 d={d}: {ms:.3f} ms  {n*d*2/ms/1e9:.2f} TB/s")
    if d == 768:
        idx.build_screen()
        for L in (32, 128, 256, 512):
            ms = timed(lambda: idx.search_screened(xq, 32, list_len=L))
            print(f"screened L={L}: {ms:.3f} ms")
    del idx, xb
