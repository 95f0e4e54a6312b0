"""Streaming rate of the K0 kernels (ingest conversion, L2 normalise, half-norms, centroid) for DESIGN.md's kernel table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragroute_amd._lib import check, lib

dev = torch.device("cuda:0")
n, d = 2_000_000, 768
x = torch.randn((n, d), device=dev)
out = torch.empty((n, d), dtype=torch.float16, device=dev)
hn = torch.empty(n, dtype=torch.float32, device=dev)
cen = torch.empty(d, dtype=torch.float32, device=dev)


def timeit(fn, bytes_moved, name, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} {ms:7.3f} ms  {bytes_moved / ms / 1e6:7.0f} GB/s  ({bytes_moved / ms / 1e6 / 8000:.2f} of 8 TB/s)")


timeit(lambda: check(lib().rr_rows_to_half(x.data_ptr(), n, d, d, out.data_ptr(), 0, d, 0, None), "x"), n * d * 6, "rows_to_half (f32 -> f16)")
timeit(lambda: check(lib().rr_rows_to_half(x.data_ptr(), n, d, d, out.data_ptr(), 0, d, 1, None), "x"), n * d * 6, "rows_to_half + normalise")
timeit(lambda: check(lib().rr_l2_normalize_f32(x.data_ptr(), n, d, None), "x"), n * d * 8, "l2_normalize_f32 (in place)")
timeit(lambda: check(lib().rr_half_sqnorms(out.data_ptr(), 0, n, d, hn.data_ptr(), None), "x"), n * d * 2, "half_sqnorms (f16 rows)")
timeit(lambda: check(lib().rr_centroid(out.data_ptr(), 0, n, d, d, cen.data_ptr(), None), "x"), n * d * 2, "centroid (f16 rows)")
