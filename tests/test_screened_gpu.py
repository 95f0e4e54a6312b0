"""GPU parity of the int8-screened search (rr_screen_build / rr_flat_search_screened) against the CPU oracle and the
plain search: same call site as index.search (ragroute/data_source.py:158,186,203), same results, fewer bytes."""
import numpy as np
import pytest
import torch

from tests.util import assert_topk_close, half_round, int_data

pytestmark = pytest.mark.gpu


def _index(gpu, xb, d, **kw):
    from ragroute_amd.flat_index import FlatIndex
    idx = FlatIndex(d, device=gpu, **kw)
    idx.add(xb)
    return idx


def _clustered(rng, n, d, n_clusters, nq):
    centres = rng.standard_normal((n_clusters, d)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    xb = centres[rng.integers(0, n_clusters, n)] + rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)
    xq = centres[rng.integers(0, n_clusters, nq)] + rng.standard_normal((nq, d)).astype(np.float32) / np.sqrt(d)
    return xb, xq


@pytest.mark.parametrize("n,nq,k,d", [(100_000, 300, 32, 768), (70_001, 5, 10, 384), (20_000, 40, 100, 1024), (300, 7, 32, 768)])
def test_screened_integer_data_bit_exact(gpu, n, nq, k, d):
    """Integer-valued rows: whatever the screen proves or falls back on, ids and scores equal the oracle's bit for bit."""
    from oracle import oracle as O
    rng = np.random.default_rng(n + nq + k)
    xb, xq = int_data(rng, n, d), int_data(rng, nq, d)
    idx = _index(gpu, xb, d).build_screen()
    D, I = idx.search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_screened_clustered_is_proven_and_equal(gpu, dtype):
    """A corpus with a score gap (clustered rows, as embedding corpora are): every query is proven exact from the int8
    pass alone (no fallback), and agrees with the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n, nq, d, k = 150_000, 64, 768, 32
    xb, xq = _clustered(rng, n, d, 200, nq)
    idx = _index(gpu, xb, d, metric="cosine", dtype=dtype).build_screen(list_len=512)
    xq_h = idx.prepare_queries(xq)
    D, I, exact = idx.search_screened(xq_h, k)
    assert int(exact.sum()) == nq
    D2, I2 = idx.search(xq, k)
    assert idx.screen_fallbacks == 0
    assert np.array_equal(I2, I.cpu().numpy())
    xbn = half_round(xb / np.linalg.norm(xb, axis=1, keepdims=True), dtype)
    xqn = half_round(xq / np.linalg.norm(xq, axis=1, keepdims=True), dtype)
    Dref, Iref = O.flat_search_ip(xbn, xqn, k)
    assert_topk_close(D2, I2, Dref, Iref, None, tol=1e-3)


def test_screened_never_claims_a_wrong_result(gpu):
    """Isotropic noise is the hard case for the bound (no score gap): the proof may fail, it must never pass on a wrong list."""
    rng = np.random.default_rng(5)
    n, nq, d, k = 400_000, 128, 768, 32
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    idx = _index(gpu, xb, d, metric="cosine")
    xq_h = idx.prepare_queries(xq)
    D0, I0 = idx.search_prepared(xq_h, k)
    idx.build_screen()
    for L in (32, 64, 256, 1024):
        D, I, exact = idx.search_screened(xq_h, k, list_len=L)
        ok = exact.bool()
        assert torch.equal(I[ok], I0[ok])
        assert torch.allclose(D[ok], D0[ok], atol=1e-5)
    assert int(exact.sum()) > 0   # the longest list proves at least some queries


def test_screened_fallback_on_ties(gpu):
    """All rows identical: no list can be proven, the batch is repeated on the f16 rows and the tie rule (ascending id) holds."""
    d, n, k = 768, 50_000, 10
    row = np.full((1, d), 0.5, dtype=np.float32)
    idx = _index(gpu, np.repeat(row, n, axis=0), d).build_screen()
    D, I = idx.search(row, k)
    assert idx.screen_fallbacks == 1
    assert np.array_equal(I[0], np.arange(k)) and np.all(D == d * 0.25)


def test_screened_route_mask_offset_and_short_corpus(gpu):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    d, n, nq, k = 768, 200, 6, 32          # fewer rows than the list: everything is re-scored, always proven
    xb, xq = int_data(rng, n, d), int_data(rng, nq, d)
    idx = _index(gpu, xb, d).build_screen()
    mask = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.uint8, device=gpu)
    D, I, exact = idx.search_screened(idx.prepare_queries(xq), k, id_offset=1 << 40, route_mask=mask)
    assert int(exact.sum()) == nq
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    for q in range(nq):
        if mask[q]:
            assert np.array_equal(I[q], Iref[q] + (1 << 40)) and np.array_equal(D[q], Dref[q])
        else:
            assert np.all(I[q] == -1) and np.all(np.isneginf(D[q]))


def test_screen_is_dropped_when_the_corpus_changes(gpu):
    rng = np.random.default_rng(12)
    idx = _index(gpu, int_data(rng, 1000, 768), 768).build_screen()
    assert idx._x8 is not None
    idx.add(int_data(rng, 10, 768))
    assert idx._x8 is None
    from ragroute_amd._lib import RagrouteHipError
    with pytest.raises(RagrouteHipError):
        idx.search_screened(idx.prepare_queries(int_data(rng, 1, 768)), 5)
    with pytest.raises(RagrouteHipError):
        _index(gpu, int_data(rng, 100, 768), 768, metric="l2").build_screen()


def test_screened_nan_rows_and_large_k(gpu):
    """NaN rows are never returned (their int8 image is arbitrary, their re-scored value NaN), also at k = 100 with the
    longest list."""
    from oracle import oracle as O
    rng = np.random.default_rng(14)
    n, d, k = 60_000, 768, 100
    xb, xq = int_data(rng, n, d), int_data(rng, 9, d)
    bad = rng.choice(n, size=25, replace=False)
    xb[bad, 7] = np.nan
    idx = _index(gpu, xb, d).build_screen(list_len=1024)
    D, I = idx.search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert not np.isin(I, bad).any()
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)
