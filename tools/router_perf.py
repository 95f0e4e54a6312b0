"""Timing of the fused router MLP for the three datasets' shapes (256 queries, synthetic weights)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.util import synth_router_case
from ragroute_amd.router import Router
for ds in ("medrag", "feb4rag", "wikipedia"):
    case = synth_router_case(ds, 1, n_queries=256)
    r = Router(ds, case["sources"], "ragroute")
    sc = case["scaler"]
    r.set_router(case["sd"], case["centroids"], *(sc if sc else (None, None)))
    q = {m: np.stack([qq[m] for qq in case["queries"]]) for m in case["queries"][0]}
    xq = r.pack_queries(q).cuda()
    for _ in range(3): r.route_batch(xq)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): r.route_batch(xq)
    b.record(); torch.cuda.synchronize()
    print(ds, "sources", len(case["sources"]), "xq", tuple(xq.shape), f"{a.elapsed_time(b) / 20 * 1000:.1f} us per 256 queries", flush=True)
