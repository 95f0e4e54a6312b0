"""Quick device timing of the flat search (development aid; bench.py is the contract)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from ragroute_amd.flat_index import FlatIndex

def main(n=1_000_000, d=768, nq=256, k=32, iters=10):
    dev = torch.device("cuda:0")
    idx = FlatIndex(d, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).to(torch.float16)
    idx.adopt(xb)
    xq = torch.randn((nq, idx.dim), generator=g, device=dev).to(torch.float16)
    for _ in range(3):
        D, I = idx.search_prepared(xq, k)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        D, I = idx.search_prepared(xq, k)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    gb = n * d * 2 / 1e9
    print(f"n={n} d={d} nq={nq} k={k}: {ms:.3f} ms/batch  {nq/ms*1e3:.0f} q/s  {gb/ms*1e3:.0f} GB/s  {2*nq*n*d/ms/1e9:.0f} TFLOP/s", flush=True)
    # spot-check against torch
    S = xq.float() @ xb[:200000].float().T
    print("sanity top1 (first 200k rows) matches where global top1 < 200k:",
          bool(((I[:, 0] >= 200000) | (I[:, 0] == S.argmax(1))).all()))

if __name__ == "__main__":
    main(n=int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, d=int(sys.argv[2]) if len(sys.argv) > 2 else 768, nq=int(sys.argv[3]) if len(sys.argv) > 3 else 256)
