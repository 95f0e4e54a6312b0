"""Screened vs plain search on a clustered corpus (what embedding corpora look like) and on isotropic noise."""
import sys, torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex
n, d, k = 10_000_000, 768, 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n_clusters = 10_000
centers = torch.randn((n_clusters, d), generator=g, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
idx = FlatIndex(d, "ip", "fp16")
xb = torch.empty((n, d), dtype=torch.float16, device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    c = torch.randint(0, n_clusters, (e - s,), generator=g, device=dev)
    t = centers[c] + torch.randn((e - s, d), generator=g, device=dev) * (1.0 / d ** 0.5)   # noise norm ~1: cos(row, centre) ~ 0.7
    xb[s:e] = (t / t.norm(dim=1, keepdim=True)).half()
idx.adopt(xb)
c = torch.randint(0, n_clusters, (256,), generator=g, device=dev)
q = centers[c] + torch.randn((256, d), generator=g, device=dev) * (1.0 / d ** 0.5)
xq = idx.prepare_queries(q / q.norm(dim=1, keepdim=True))
D0, I0 = idx.search_prepared(xq, k)
print("plain nq=256 ms", timed(lambda: idx.search_prepared(xq, k)), " nq=1 ms", timed(lambda: idx.search_prepared(xq[:1], k)))
print("score of rank 1 / 32:", float(D0[:, 0].mean()), float(D0[:, -1].mean()))
idx.build_screen()
for L in (64, 128, 256, 512):
    D, I, ex = idx.search_screened(xq, k, list_len=L)
    print(f"L={L}: proven {int(ex.sum())}/256 ids equal {int((I == I0).all(dim=1).sum())}/256  nq=256 ms {timed(lambda: idx.search_screened(xq, k, list_len=L)):.3f}"
          f"  nq=1 ms {timed(lambda: idx.search_screened(xq[:1], k, list_len=L)):.3f}")
