"""Seeded random (d, n, nq, k, dtype, metric) cases for the flat search, biased to the kernel-selection boundaries, and
(segment_cases / run_segment_case) random segment tables + route masks for the one-pass search over several sources.

Shared by tests/test_fuzz_gpu.py (a bounded set, every GPU run) and tools/fuzz_parity.py (as many as the budget allows)."""
import numpy as np

# every row width class: d <= 768 (resident queries), half-resident 896..1536, wide rows (multiples of 128 up to 8192),
# and widths that need zero padding
DIMS = [8, 64, 100, 128, 256, 384, 500, 512, 640, 768, 769, 896, 1000, 1024, 1152, 1280, 1408, 1536, 1664, 1792, 2048, 2432, 2560,
        2816, 3072, 3200, 4096, 5120, 8192]
QUERIES = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300]
ROWS = [1, 31, 32, 33, 255, 256, 257, 1000, 8191, 8192, 8193, 8224, 10_000, 16_384, 20_011, 33_000, 50_000]
KS = [1, 5, 10, 32, 33, 100, 128, 129, 300, 600, 1024]   # (129 ... 1024 on wide rows: the row-split kernel's any-k path, round 4)


ROWS_BIG = [70_000, 131_072, 300_000, 524_288, 1_000_000, 2_000_000]   # several chunks of the geometric schedule


def cases(seed, count, max_work=6e9, big=False):
    """-> list of dicts; work = n * d * nq (the oracle's f64 dot products) is bounded so a case checks in a second or two.
    big: corpora of 70 K ... 2 M rows (bootstrap + 2 ... 4 chunk launches + compactions) with correspondingly few queries."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        d = int(rng.choice(DIMS))
        n = int(rng.choice(ROWS_BIG if big else ROWS)) + int(rng.integers(0, 3))
        nq = int(rng.choice(QUERIES))
        k = int(rng.choice(KS))
        if n * d * nq > max_work or n * d > 1.5e9:   # (host-side generation of the corpus stays in seconds)
            continue
        out.append({"d": d, "n": n, "nq": nq, "k": k, "dtype": "fp16" if rng.integers(0, 2) else "bf16",
                    "metric": "l2" if rng.integers(0, 4) == 0 else "ip", "seed": int(rng.integers(0, 2 ** 31))})
    return out


def run_case(c, dev, guard=True):
    """One case on integer data (exact scores: ids and scores must match the oracle bit for bit).  guard: corpus and
    queries end on the last byte of their own device segment, so an out-of-range read faults."""
    import torch
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    from tests.test_guard_pages_gpu import _flush_to_end
    from tests.util import int_data
    rng = np.random.default_rng(c["seed"])
    d, n, nq, k = c["d"], c["n"], c["nq"], c["k"]
    xb, xq = int_data(rng, n, d), int_data(rng, nq, d)
    idx = FlatIndex(d, metric=c["metric"], dtype=c["dtype"], device=dev)
    tdt = torch.float16 if c["dtype"] == "fp16" else torch.bfloat16
    xb_h = torch.zeros((n, idx.dim), dtype=tdt)
    xb_h[:, :d] = torch.from_numpy(xb).to(tdt)
    xq_h = torch.zeros((nq, idx.dim), dtype=tdt)
    xq_h[:, :d] = torch.from_numpy(xq).to(tdt)
    if guard:
        keep_b, xb_dev = _flush_to_end(xb_h, dev)
        keep_q, xq_dev = _flush_to_end(xq_h, dev)
    else:
        keep_b = keep_q = None
        xb_dev, xq_dev = xb_h.to(dev), xq_h.to(dev)
    idx.adopt(xb_dev)
    D, I = idx.search_prepared(xq_dev, k)
    torch.cuda.synchronize()
    Dr, Ir = (O.flat_search_l2 if c["metric"] == "l2" else O.flat_search_ip)(xb, xq, k)
    ok = np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
    del keep_b, keep_q
    return ok


SEG_ROWS = [0, 1, 31, 33, 255, 256, 257, 1000, 3633, 8192, 8674, 20_011, 57_638, 126_000, 301_000]   # FeB4RAG / MedRAG small-source sizes among them


def segment_cases(seed, count, max_work=8e9):
    """Random segment tables for rr_flat_search_segments: 1 ... 7 sources of assorted sizes (empty and one-row sources, totals
    below and above the 8192-row dense path, several chunks), every width class, batch sizes around the query-block boundaries,
    mask densities from 'everything' to 'almost nothing' (a query routed to one tiny source is the hard case for the schedule)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        d = int(rng.choice([64, 384, 768, 769, 1024, 1536, 2048, 4096]))
        n_seg = int(rng.integers(1, 8))
        rows = [int(rng.choice(SEG_ROWS)) + int(rng.integers(0, 2)) for _ in range(n_seg)]
        nq = int(rng.choice(QUERIES))
        k = int(rng.choice([1, 10, 32, 100]))
        if sum(rows) * d * nq > max_work or sum(rows) * d > 1.2e9:
            continue
        out.append({"d": d, "rows": rows, "nq": nq, "k": k, "dtype": "fp16" if rng.integers(0, 2) else "bf16",
                    "density": float(rng.choice([1.0, 0.9, 0.5, 0.15])), "seed": int(rng.integers(0, 2 ** 31))})
    return out


def run_segment_case(c, dev):
    """One segmented case on integer data against the oracle chain (per-source oracle top-k -> concat -> oracle merge), the matrix
    flush against the end of its device segment, gaps poisoned."""
    import torch
    from ragroute_amd.flat_index import SegmentedIndex
    from tests.test_guard_pages_gpu import _flush_to_end
    from tests.test_segments_gpu import oracle_chain
    from tests.util import int_data
    rng = np.random.default_rng(c["seed"])
    d, rows, nq, k = c["d"], c["rows"], c["nq"], c["k"]
    parts = [int_data(rng, n, d) for n in rows]
    xq = int_data(rng, nq, d)
    mask = rng.random((nq, len(rows))) < c["density"]
    seg = SegmentedIndex(d, rows, dtype=c["dtype"], device=dev)
    tdt = torch.float16 if c["dtype"] == "fp16" else torch.bfloat16
    host = torch.full((max(1, seg.n_rows_total), seg.dim), float("nan"), dtype=tdt)
    host[::5] = 30000.0
    for s_, p in enumerate(parts):
        host[seg.begins[s_]: seg.begins[s_] + rows[s_]] = 0
        host[seg.begins[s_]: seg.begins[s_] + rows[s_], :d] = torch.from_numpy(p).to(tdt)
    keep_b, arena = _flush_to_end(host, dev)
    seg._xb = arena
    xq_h = torch.zeros((nq, seg.dim), dtype=tdt)
    xq_h[:, :d] = torch.from_numpy(xq).to(tdt)
    keep_q, xq_dev = _flush_to_end(xq_h, dev)
    D, I = seg.search_prepared(xq_dev, k, route_mask=torch.from_numpy(mask.astype(np.uint8)).to(dev))
    torch.cuda.synchronize()
    Dr, Ir = oracle_chain(parts, xq, k, mask)
    ok = np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
    del keep_b, keep_q
    return ok
