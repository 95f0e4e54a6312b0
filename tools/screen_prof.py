"""rocprofv3 target: a few screened searches at full size.  usage: screen_prof.py n_rows list_len [nq]"""
import sys
import torch
sys.path.insert(0, ".")
from ragroute_amd.flat_index import FlatIndex

n, L = int(sys.argv[1]), int(sys.argv[2])
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
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
q = torch.randn((nq, d), generator=g, device=dev)
xq = idx.prepare_queries(q / q.norm(dim=1, keepdim=True))
idx.build_screen()
for _ in range(4):
    D, I, ex = idx.search_screened(xq, k, list_len=L)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    D, I, ex = idx.search_screened(xq, k, list_len=L)
b.record(); torch.cuda.synchronize()
print("proven", int(ex.sum()), "of", nq, "ms", a.elapsed_time(b) / 5)
