"""BASELINE.json config 5 on one GPU: synthetic 80M x 768 bf16 (HBM-resident, 122.9 GB), k=100, B=256.
Parity by planted neighbours (the oracle cannot scan 80M rows), then throughput with HIP events."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragroute_amd.flat_index import FlatIndex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
d, nq, k = 768, 256, 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(99)
t0 = time.time()
xb = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    xb[s:e] = (torch.randn((e - s, d), generator=g, device=dev) / d ** 0.5).to(torch.bfloat16)
xq = torch.randn((nq, d), generator=g, device=dev)
xq = (xq / xq.norm(dim=1, keepdim=True)).to(torch.bfloat16)
pos = torch.randperm(n, generator=g, device=dev)[: nq * k].reshape(nq, k)
scale = 1 + torch.arange(k, device=dev, dtype=torch.float32) / 32          # gaps of 1/32 >> bf16 noise
planted = (xq.float()[:, None, :] * scale[None, :, None]).to(torch.bfloat16)
xb[pos.reshape(-1)] = planted.reshape(-1, d)
torch.cuda.synchronize()
print(f"generated {n} x {d} bf16 ({n * d * 2 / 1e9:.1f} GB) in {time.time() - t0:.1f} s", flush=True)
idx = FlatIndex(d, dtype="bf16", device=dev)
idx.adopt(xb)
D, I = idx.search_prepared(xq, k)
want = (planted.float() * xq.float()[:, None, :]).sum(-1)
order = torch.argsort(want, dim=1, descending=True, stable=True)
ids_ok = bool(torch.equal(I, torch.gather(pos, 1, order)))
score_err = float((D - torch.gather(want, 1, order)).abs().max())
for _ in range(2):
    idx.search_prepared(xq, k)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 5
ev0.record()
for _ in range(iters):
    idx.search_prepared(xq, k)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / iters
res = {"config": f"{n} x {d} bf16, B={nq}, k={k}, 1 GPU", "planted_top_k_ids_identical": ids_ok, "max_abs_score_error": score_err,
       "ms_per_batch": round(ms, 3), "queries_per_s": round(nq / ms * 1e3, 1), "GB_per_s": round(n * d * 2 / ms / 1e6, 1),
       "TFLOP_per_s": round(2 * nq * n * d / ms / 1e9, 1), "hbm_frac_of_8TBs": round(n * d * 2 / ms / 1e6 / 8000, 4)}
print(json.dumps(res), flush=True)
assert ids_ok and score_err < 2e-2
