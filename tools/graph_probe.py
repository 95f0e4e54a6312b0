"""GPU: does replaying a search from a captured HIP graph shorten it?  (launch gaps between its ~10 dependent kernels)

    python tools/graph_probe.py ROWS DIM [BATCH] [K]

Times ITERS back-to-back searches issued eagerly (ctypes -> rr_flat_search -> hipLaunchKernel x ~10) and the same search
replayed from a torch.cuda.CUDAGraph, one HIP-event pair around the whole block each."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragroute_amd.flat_index import FlatIndex


def main():
    n, d = int(sys.argv[1]), int(sys.argv[2])
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    iters = 50
    dev = torch.device("cuda:0")
    idx = FlatIndex(d, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    xb = torch.empty((n, idx.dim), dtype=torch.float16, device=dev)
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        xb[s:e] = (torch.randn((e - s, idx.dim), generator=g, device=dev) / d ** 0.5).half()
    idx.adopt(xb)
    xq = torch.randn((nq, idx.dim), generator=g, device=dev).half()
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for _ in range(5):
        idx.search_prepared(xq, k, out=(D, I))
    torch.cuda.synchronize()
    D0, I0 = D.clone(), I.clone()

    def block(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    eager = min(block(lambda: idx.search_prepared(xq, k, out=(D, I))) for _ in range(3))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        idx.search_prepared(xq, k, out=(D, I))
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        idx.search_prepared(xq, k, out=(D, I))
    D.zero_()
    graph.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(D, D0) and torch.equal(I, I0))
    replay = min(block(graph.replay) for _ in range(3))
    print(json.dumps({"rows": n, "dim": d, "batch": nq, "k": k, "eager_ms": round(eager, 4), "graph_ms": round(replay, 4),
                      "graph_over_eager": round(replay / eager, 4), "graph_result_identical": same}))


if __name__ == "__main__":
    main()
