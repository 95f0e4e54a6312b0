"""GPU: no kernel may read past the end of the corpus or of the query block.

Both tensors are placed flush against the END of a mapping made with HIP's virtual-memory API whose address range is one granule
longer than the mapping (tests/guard_alloc.py), so a read beyond the last byte hits unmapped addresses and faults instead of
silently reading a neighbour.  (Rounds 1-2 relied on PyTorch's caching allocator handing out exact-size segments; asserting the
segment end in round 3 showed it often did not.)  Results are also checked against the oracle; the cases run in child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import int_data

pytestmark = pytest.mark.gpu


def _flush_to_end(t, dev):
    """Copy of tensor t whose last byte is the last MAPPED byte of its address range (tests/guard_alloc.py: HIP virtual-memory
    reservation one granule longer than the mapping), so a read past it faults whatever the caching allocator's state.  Returns
    (owner, view): keep `owner` alive as long as the view is used."""
    from tests.guard_alloc import GuardedBuffer
    nbytes = max(1, t.numel() * t.element_size())
    owner = GuardedBuffer(nbytes, dev.index or 0)
    view = owner.tensor(tuple(t.shape), t.dtype) if t.numel() else torch.empty(t.shape, dtype=t.dtype, device=dev)
    if t.numel():
        view.copy_(t)
    return owner, view


def _run_cases(dev, cases):
    from oracle import oracle as O
    from ragroute_amd.flat_index import FlatIndex
    rng = np.random.default_rng(17)
    for d, n, nqs in cases:
        xb = int_data(rng, n, d)
        idx = FlatIndex(d, device=dev)
        dim = idx.dim
        xb_h = torch.zeros((n, dim), dtype=torch.float16)
        xb_h[:, :d] = torch.from_numpy(xb).half()
        keep_b, xb_dev = _flush_to_end(xb_h, dev)
        idx.adopt(xb_dev)
        for nq in nqs:
            xq = int_data(rng, nq, d)
            xq_h = torch.zeros((nq, dim), dtype=torch.float16)
            xq_h[:, :d] = torch.from_numpy(xq).half()
            keep_q, xq_dev = _flush_to_end(xq_h, dev)
            D, I = idx.search_prepared(xq_dev, 10)
            torch.cuda.synchronize()
            Dr, Ir = O.flat_search_ip(xb, xq, 10)
            assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr), (d, n, nq)
            del keep_q
        del keep_b


def _run_segmented(dev):
    from ragroute_amd import _lib
    from ragroute_amd.flat_index import SegmentedIndex
    from tests.test_segments_gpu import oracle_chain
    rng = np.random.default_rng(23)
    for d, rows, nqs in [(768, [20_000, 333, 13_001], (1, 256)), (1024, [9_000, 17_777], (100, 256)), (4096, [9_001, 5_003], (3, 256))]:
        parts = [int_data(rng, n, d) for n in rows]
        seg = SegmentedIndex(d, rows, device=dev)
        host = torch.zeros((seg.n_rows_total, seg.dim), dtype=torch.float16)
        for s_, p in enumerate(parts):
            host[seg.begins[s_]: seg.begins[s_] + rows[s_], :d] = torch.from_numpy(p).half()
        keep_b, arena = _flush_to_end(host, dev)
        seg._xb = arena                                       # the matrix now ends with its device segment
        for nq in nqs:
            xq = int_data(rng, nq, d)
            xq_h = torch.zeros((nq, seg.dim), dtype=torch.float16)
            xq_h[:, :d] = torch.from_numpy(xq).half()
            keep_q, xq_dev = _flush_to_end(xq_h, dev)
            mask = torch.from_numpy(rng.integers(0, 2, size=(nq, len(rows))).astype(np.uint8)).to(dev)
            D, I = seg.search_prepared(xq_dev, 10, route_mask=mask)
            torch.cuda.synchronize()
            Dr, Ir = oracle_chain(parts, xq, 10, mask.cpu().numpy().astype(bool))
            assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr), (d, rows, nq)
            del keep_q
        del keep_b


CASES = [(768, 33_000, (1, 4, 100, 256)), (384, 20_011, (3, 256)), (1024, 40_000, (1, 4, 128, 129, 256)),
         (1536, 20_000, (7, 200)), (2048, 30_000, (1, 4, 17, 256)), (4096, 12_345, (1, 5, 256))]


def _in_child(code, extra_env=None):
    """An over-read is a GPU fault that takes the process down: run the guard layout in a child, so that it fails ONE test
    instead of aborting the suite (and so that RR_WIDE_WAVES, read once per process, can be forced)."""
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), **(extra_env or {}))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "ok" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-2000:])


def test_no_read_past_the_end_of_corpus_or_queries(gpu):
    _in_child("import torch, tests.test_guard_pages_gpu as t; t._run_cases(torch.device('cuda:0'), t.CASES); print('ok')")


def test_no_read_past_the_end_segmented_search(gpu):
    """The same layout for rr_flat_search_segments: the matrix of three sources ends flush against unmapped pages (the last
    run of a chunk launch is padded to whole 256-row groups in ordinal space: the pad must clamp, not read on)."""
    _in_child("import torch, tests.test_guard_pages_gpu as t; t._run_segmented(torch.device('cuda:0')); print('ok')")


def test_no_read_past_the_end_every_wide_row_kernel(gpu):
    """Query counts that reach every wide-row kernel under the product dispatch (a product library ignores RR_WIDE_WAVES, which forced
    the flavours in rounds 2-3): 9 / 11 / 12 query blocks and d > 2048 up to 192 queries -> 4 waves (flat_scan_wide_pd_kernel), other
    counts up to 208 at d <= 2048 -> 8 waves (flat_scan_wide8_kernel), 209 ... 256 (193 ... 256 at d > 2048) -> the row-split kernel;
    17 ... 128 at d <= 1536 the half-resident kernel."""
    from ragroute_amd import _lib
    L = _lib.lib()
    names = {L.rr_flat_scan_kernel_name(d, q).decode() for d, q in ((1024, 144), (1024, 200), (1024, 256), (1024, 100), (2048, 160), (2048, 176), (4096, 180), (4096, 200))}
    assert names == {"flat_scan_wide_pd_kernel", "flat_scan_wide8_kernel", "flat_scan_wide_rs_kernel", "flat_scan16h_kernel"}, names
    _in_child("import torch, tests.test_guard_pages_gpu as t; "
              "t._run_cases(torch.device('cuda:0'), [(1024, 40_000, (100, 144, 200, 209, 256)), (2048, 30_000, (1, 4, 17, 160, 176, 208, 256)), "
              "(4096, 12_345, (1, 5, 180, 193, 200, 256))]); print('ok')")
