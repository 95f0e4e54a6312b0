"""Latency of the list-shaped merge the front-end calls per query (`rerank_medrag(docs, scores, k)`, http_server.py:288-293):
round 2's device round trip (list -> CUDA tensor -> rr_merge_topk -> list, three host<->device hops) against round 3's host
sort (numpy, no HIP context), on the reference's sizes: S*k = 4*32 = 128 candidates (medrag), 10*10 (wikipedia).

    python tools/rerank_latency.py          (needs an MI355X for the "device" column only)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ragroute_amd.rerank import merge_topk, rerank_medrag


def device_list_rerank(docs, scores, k):
    """what ragroute_amd/rerank.py did in round 2"""
    s32 = np.asarray(scores, np.float64).astype(np.float32)
    D = torch.from_numpy(s32[None, :]).to("cuda")
    I = torch.arange(len(scores), dtype=torch.int64, device="cuda")[None, :]
    _, o = merge_topk(D, I, min(k, len(scores)), True)
    order = [i for i in o[0].cpu().tolist() if i >= 0]
    return [docs[i] for i in order], [scores[i] for i in order]


def timeit(fn, n):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t.sort()
    return {"p50_us": round(t[len(t) // 2] * 1e6, 1), "p99_us": round(t[int(len(t) * 0.99)] * 1e6, 1)}


def main():
    rng = np.random.default_rng(0)
    out = {}
    for name, m, k in (("medrag 4 x 32 candidates, k = 32", 128, 32), ("wikipedia 10 x 10 candidates, k = 10", 100, 10)):
        scores = rng.standard_normal(m).tolist()
        docs = [{"id": i} for i in range(m)]
        row = {"host_numpy_round3": timeit(lambda: rerank_medrag(docs, scores, k), 2000)}
        if torch.cuda.is_available():
            row["device_round_trip_round2"] = timeit(lambda: device_list_rerank(docs, scores, k), 500)
            assert device_list_rerank(docs, scores, k) == rerank_medrag(docs, scores, k)
        out[name] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
