"""GPU parity of the flat search (K1/K2) through the C ABI against the CPU oracle.
Reference call site: ragroute/data_source.py:158,186,203 (index.search)."""
import numpy as np
import pytest

from tests.util import assert_topk_close, half_round, int_data

pytestmark = pytest.mark.gpu


def _index(gpu, xb, d, **kw):
    from ragroute_amd.flat_index import FlatIndex
    idx = FlatIndex(d, device=gpu, **kw)
    if xb.shape[0]:
        idx.add(xb)
    return idx


@pytest.mark.parametrize("n,nq,k", [(5000, 256, 32), (100_000, 300, 32), (70_001, 5, 10), (33, 3, 32), (8193, 17, 100)])
def test_integer_data_bit_exact(gpu, n, nq, k):
    """Integer-valued embeddings: scores are exact, so ids AND scores must equal the oracle's bit for
    bit, including the (score desc, id asc) tie rule, across the dense path, the bootstrap and the chunks."""
    from oracle import oracle as O
    rng = np.random.default_rng(n + nq + k)
    xb, xq = int_data(rng, n, 768), int_data(rng, nq, 768)
    D, I = _index(gpu, xb, 768).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert np.array_equal(I, Iref)
    assert np.array_equal(D, Dref)


def test_gaussian_fp16(gpu):
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    n, nq, d, k = 200_000, 64, 768, 32
    xb = half_round(rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d))
    xq = half_round(rng.standard_normal((nq, d)).astype(np.float32))
    D, I = _index(gpu, xb, d).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    S = (xq.astype(np.float64) @ xb.astype(np.float64).T).astype(np.float32)
    assert_topk_close(D, I, Dref, Iref, S, tol=1e-3)
    recall = np.mean([len(set(I[q]) & set(Iref[q])) / k for q in range(nq)])
    assert recall >= 0.999


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("d,k", [(100, 10), (384, 32), (768, 100), (800, 10), (1024, 10), (1536, 32), (4096, 10)])
def test_dims_and_dtypes(gpu, dtype, d, k):
    from oracle import oracle as O
    rng = np.random.default_rng(d + k)
    n, nq = 30_000, 9
    xb = half_round(rng.standard_normal((n, d)).astype(np.float32), dtype)
    xq = half_round(rng.standard_normal((nq, d)).astype(np.float32), dtype)
    D, I = _index(gpu, xb, d, dtype=dtype).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert_topk_close(D, I, Dref, Iref, None, tol=2e-3 * np.sqrt(d))


@pytest.mark.parametrize("n", [0, 1, 31, 32])
def test_tiny_and_empty(gpu, n):
    """k > ntotal pads with (-inf, -1) like FAISS; empty index returns only padding."""
    from oracle import oracle as O
    rng = np.random.default_rng(n)
    xb, xq = int_data(rng, n, 768), int_data(rng, 4, 768)
    D, I = _index(gpu, xb, 768).search(xq, 32)
    Dref, Iref = O.flat_search_ip(xb.reshape(n, 768), xq, 32)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


@pytest.mark.parametrize("d", [1024, 4096])
def test_generic_dimension_bit_exact(gpu, d):
    """FeB4RAG-sized embeddings (config.py:45-57) take the generic-dimension kernel: integer data, exact parity,
    across bootstrap + chunks (n > 8192) and with an odd number of tiles."""
    from oracle import oracle as O
    rng = np.random.default_rng(d)
    n = 20_033
    xb, xq = int_data(rng, n, d, -1, 2), int_data(rng, 20, d, -1, 2)
    D, I = _index(gpu, xb, d).search(xq, 10)
    Dref, Iref = O.flat_search_ip(xb, xq, 10)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)
    D2, I2 = _index(gpu, xb[:700], d).search(xq, 10)      # dense path
    Dr2, Ir2 = O.flat_search_ip(xb[:700], xq, 10)
    assert np.array_equal(I2, Ir2) and np.array_equal(D2, Dr2)


@pytest.mark.parametrize("nq", [1, 16, 17, 63, 65, 129, 200])
def test_partial_query_blocks(gpu, nq):
    """Waves / query blocks without real queries skip their MFMA work; every batch size must stay exact."""
    from oracle import oracle as O
    rng = np.random.default_rng(nq)
    xb, xq = int_data(rng, 40_000, 768), int_data(rng, nq, 768)
    D, I = _index(gpu, xb, 768).search(xq, 32)
    Dref, Iref = O.flat_search_ip(xb, xq, 32)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


def test_no_queries(gpu):
    idx = _index(gpu, int_data(np.random.default_rng(0), 100, 768), 768)
    D, I = idx.search(np.zeros((0, 768), np.float32), 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)


def test_sorted_corpus_forces_compaction(gpu):
    """Adversarial order: scores increase with the row id, so every tile beats every threshold and the
    per-lane candidate buffers overflow again and again; the exact in-kernel compaction must keep the
    result identical to the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n, d, k = 70_000, 768, 32
    xb = int_data(rng, n, d, -1, 2)
    xb[:, 0] = np.arange(n) // 40          # <= 1749, exact in fp16
    xq = np.zeros((3, d), np.float32)
    xq[0, 0] = 1.0                          # score = row // 40 : ascending with ties
    xq[1, 0] = -1.0                         # descending: nothing after the first tiles survives
    xq[2] = int_data(rng, 1, d)[0]
    D, I = _index(gpu, xb, d).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


def test_all_ties(gpu):
    """Every score equal: the answer is ids 0..k-1 (ascending-id tie rule), from every code path."""
    n, d, k = 20_000, 768, 32
    xb = np.ones((n, d), np.float32)
    xq = np.ones((2, d), np.float32)
    D, I = _index(gpu, xb, d).search(xq, k)
    assert np.array_equal(I, np.tile(np.arange(k), (2, 1)))
    assert np.array_equal(D, np.full((2, k), 768.0, np.float32))


def test_cosine_metric(gpu):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n, nq, d, k = 50_000, 8, 768, 10
    xb = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.1, 10, (n, 1)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32) * 3
    D, I = _index(gpu, xb, d, metric="cosine").search(xq, k)
    xbn, xqn = xb.copy(), xq.copy()
    O.normalize_L2(xbn)
    O.normalize_L2(xqn)
    Dref, Iref = O.flat_search_ip(half_round(xbn), half_round(xqn), k)
    assert_topk_close(D, I, Dref, Iref, None, tol=1e-3)
    assert np.all(D <= 1.0 + 1e-3)


def test_batched_equals_single_query(gpu):
    """The reference searches one query per call (data_source.py:114); a 256-query batch must give each
    query exactly the result of its own single-query call."""
    rng = np.random.default_rng(5)
    n, d, k = 60_000, 768, 32
    xb = half_round(rng.standard_normal((n, d)).astype(np.float32))
    xq = half_round(rng.standard_normal((40, d)).astype(np.float32))
    idx = _index(gpu, xb, d)
    D, I = idx.search(xq, k)
    for q in (0, 17, 39):
        D1, I1 = idx.search(xq[q : q + 1], k)
        assert np.array_equal(I1[0], I[q]) and np.array_equal(D1[0], D[q])


def test_bad_arguments(gpu):
    from ragroute_amd import RagrouteHipError
    idx = _index(gpu, int_data(np.random.default_rng(0), 64, 768), 768)
    with pytest.raises((ValueError, RagrouteHipError)):
        idx.search(np.zeros((1, 768), np.float32), 0)
    with pytest.raises((ValueError, RagrouteHipError)):
        idx.search(np.zeros((1, 768), np.float32), 5000)
    with pytest.raises(ValueError):
        idx.search(np.zeros((1, 100), np.float32), 5)


def test_randomised_shapes(gpu):
    """30 seeded random (n, d, nq, k, dtype) configurations on exact integer data: ids and scores bit for bit."""
    from oracle import oracle as O
    rng = np.random.default_rng(20261004)
    for it in range(30):
        n = int(rng.choice([0, 1, 33, 700, 8192, 8193, 9000, 20_000, 66_000]) + rng.integers(0, 50))
        d = int(rng.choice([8, 64, 100, 128, 200, 384, 600, 768, 769, 1024]))
        nq = int(rng.choice([1, 2, 15, 16, 17, 64, 100, 256, 257]))
        k = int(rng.choice([1, 5, 10, 32, 33, 100, 128, 200]))
        dtype = "fp16" if rng.integers(0, 2) else "bf16"
        xb, xq = int_data(rng, n, d, -2, 3), int_data(rng, nq, d, -2, 3)
        D, I = _index(gpu, xb, d, dtype=dtype).search(xq, k)
        Dref, Iref = O.flat_search_ip(xb.reshape(n, d), xq, k)
        assert np.array_equal(I, Iref), (it, n, d, nq, k, dtype)
        assert np.array_equal(D, Dref), (it, n, d, nq, k, dtype)


@pytest.mark.parametrize("n,nq,k,d", [(5000, 40, 10, 768), (90_000, 256, 32, 768), (30_000, 7, 100, 200), (33, 3, 50, 128),
                                      (40_000, 200, 10, 1024), (40_000, 5, 10, 1024), (20_000, 9, 10, 2048), (300, 3, 10, 4096),
                                      # 209 ... 256 queries on wide rows: the row-split kernel's L2 form (round 4), incl. a ragged last tile
                                      (40_000, 256, 10, 1024), (33_333, 209, 32, 2048), (20_011, 256, 100, 4096), (50_000, 300, 10, 1536)])
def test_l2_metric_bit_exact_on_integer_data(gpu, n, nq, k, d):
    """Squared-L2 flat search (faiss.IndexFlatL2 contract): nearest first, ties by ascending id, (+inf,-1) padding."""
    from oracle import oracle as O
    rng = np.random.default_rng(n + k)
    xb, xq = int_data(rng, n, d, -2, 3), int_data(rng, nq, d, -2, 3)
    D, I = _index(gpu, xb, d, metric="l2").search(xq, k)
    Dref, Iref = O.flat_search_l2(xb, xq, k)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


def test_l2_metric_gaussian(gpu):
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n, nq, d, k = 100_000, 16, 768, 10
    xb = half_round(rng.standard_normal((n, d)).astype(np.float32))
    xq = half_round(rng.standard_normal((nq, d)).astype(np.float32))
    D, I = _index(gpu, xb, d, metric="l2").search(xq, k)
    Dref, Iref = O.flat_search_l2(xb, xq, k)
    assert np.allclose(D, Dref, rtol=2e-5, atol=2e-2)
    assert np.mean([len(set(I[q]) & set(Iref[q])) / k for q in range(nq)]) >= 0.99


def test_nan_rows_are_never_selected(gpu):
    """A row whose score is NaN is absent from every result (oracle and GPU agree), through the dense path and the chunks."""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    for n in (3000, 40_000):
        xb, xq = int_data(rng, n, 768), int_data(rng, 5, 768)
        bad = rng.choice(n, size=20, replace=False)
        xb[bad, 3] = np.nan
        D, I = _index(gpu, xb, 768).search(xq, 32)
        Dref, Iref = O.flat_search_ip(xb, xq, 32)
        assert not np.isin(I, bad).any()
        assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


@pytest.mark.parametrize("k", [500, 1024])
def test_large_k(gpu, k):
    """k up to RR_MAX_K = 1024 (FAISS-GPU's own limit is 2048): bigger candidate buffers, same exact result."""
    from oracle import oracle as O
    rng = np.random.default_rng(k)
    xb, xq = int_data(rng, 30_000, 256), int_data(rng, 6, 256)
    D, I = _index(gpu, xb, 256).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


def test_repeated_searches_are_stable(gpu):
    """Back-to-back searches on one index reuse the workspace and the LDS of resident workgroups: every repetition must give
    the same exact result (ip, l2, and a partial query block)."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for metric, n, nq, k in (("ip", 100_000, 300, 32), ("l2", 90_000, 256, 32), ("ip", 30_000, 7, 100)):
        xb, xq = int_data(rng, n, 768), int_data(rng, nq, 768)
        idx = _index(gpu, xb, 768, metric=metric)
        Dr, Ir = (O.flat_search_l2 if metric == "l2" else O.flat_search_ip)(xb, xq, k)
        for _ in range(3):
            D, I = idx.search(xq, k)
            assert np.array_equal(I, Ir) and np.array_equal(D, Dr), metric


def test_l2_metric_bf16_and_single_query(gpu):
    """L2 on bf16 storage, and the small-batch path (1 query: three of the four waves only feed the DMA ring)."""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    xb, xq = int_data(rng, 70_000, 640), int_data(rng, 1, 640)
    D, I = _index(gpu, xb, 640, metric="l2", dtype="bf16").search(xq, 10)
    Dref, Iref = O.flat_search_l2(xb, xq, 10)
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("case", ["gauss", "int", "cosine"])
def test_committed_fixture(gpu, case, dtype):
    """The HIP path against tests/golden/flat_search.npz (f64 numpy statement, generated by tests/golden/make_golden.py)."""
    import os
    from tests.util import flat_golden_inputs
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "flat_search.npz"))
    xb, xq = flat_golden_inputs(case)
    if dtype == "bf16" and case != "int":
        pytest.skip("the fixture's non-integer inputs are fp16-rounded")
    D, I = _index(gpu, xb, 768, dtype=dtype).search(xq, 32)
    assert np.array_equal(I, G[case + "_I"])
    if case == "int":
        assert np.array_equal(D.astype(np.float64), G["int_D"])
    else:
        assert np.allclose(D, G[case + "_D"], atol=1e-3, rtol=0)


@pytest.mark.parametrize("d,nq", [(1024, 200), (1024, 3), (2048, 130), (4096, 17)])
def test_wide_rows_adversarial(gpu, d, nq):
    """The wide-row kernel (and the half-resident one for small batches at 1024) under the same adversarial inputs as the
    d = 768 kernel: ascending scores (every tile beats every threshold, buffers overflow), all-equal scores, NaN rows,
    a partial last query block, k = 100 - bit-exact against the oracle on integer data."""
    from oracle import oracle as O
    rng = np.random.default_rng(d + nq)
    n, k = 40_000, 100
    xb = int_data(rng, n, d, -1, 2)
    xb[:, 0] = np.arange(n) // 40                      # <= 999, exact in fp16
    bad = rng.choice(n, size=10, replace=False)
    xb[bad, 5] = np.nan
    xq = int_data(rng, nq, d, -1, 2)
    xq[0] = 0.0; xq[0, 0] = 1.0                        # ascending with ties
    xq[1] = 0.0; xq[1, 0] = -1.0                       # descending
    xq[2] = 0.0                                        # all scores equal (zero query): ids 0..k-1 minus the NaN rows
    D, I = _index(gpu, xb, d).search(xq, k)
    Dref, Iref = O.flat_search_ip(xb, xq, k)
    assert not np.isin(I, bad).any()
    assert np.array_equal(I, Iref) and np.array_equal(D, Dref)


@pytest.mark.parametrize("d,n,nq,k", [(1024, 60_000, 200, 500), (2048, 40_000, 256, 100), (4096, 30_000, 256, 300), (1536, 50_000, 129, 1024),
                                      # row-split kernel beyond k = 128 (round 4: 8 candidate buffers per workgroup at every k)
                                      (2048, 40_000, 256, 300), (1024, 70_000, 240, 1024), (4096, 30_000, 209, 129)])
def test_wide_rows_large_k(gpu, d, n, nq, k):
    """Large k through the wide-row kernels (two waves per SIMD up to d = 2048, one above): candidate buffers of 2k entries per
    lane, compactions inlined in the kernel, several chunks.  Integer data: ids and scores bit for bit."""
    from oracle import oracle as O
    rng = np.random.default_rng(d + k)
    xb, xq = int_data(rng, n, d), int_data(rng, nq, d)
    D, I = _index(gpu, xb, d).search(xq, k)
    Dr, Ir = O.flat_search_ip(xb, xq, k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


@pytest.mark.parametrize("d,nq", [(1024, 300), (1280, 260), (1536, 385), (1024, 128), (896, 129)])
def test_more_than_one_query_block_at_half_resident_widths(gpu, d, nq):
    """768 < d <= 1536: a call is served in blocks of 256 queries; a block of up to 128 takes the query-resident kernel (which
    reads the prep kernel's fragment-order copy), a fuller one the wide-row kernel (row-major queries) - e.g. 300 queries =
    256 + 44.  Each block must get the layout its own kernel reads."""
    from oracle import oracle as O
    rng = np.random.default_rng(d + nq)
    xb, xq = int_data(rng, 20_000, d), int_data(rng, nq, d)
    D, I = _index(gpu, xb, d).search(xq, 10)
    Dr, Ir = O.flat_search_ip(xb, xq, 10)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


@pytest.mark.parametrize("d", [2048, 4096, 1792, 6144, 8192])
def test_wide_row_scores_do_not_depend_on_batch_size_or_k(gpu, d):
    """Rows wider than 1536 walk their K loop in a rotated order that is a function of the global 256-row group alone
    (flat_scan_wide.hip): the same (query, row) must score bit-identically whatever kernel instance (1 query, 17, 256), chunk
    schedule (k = 10 vs 100 moves the chunk boundaries) or launch (sample vs chunk) computes it - on gaussian data, where a
    different summation order WOULD show in the last bits."""
    rng = np.random.default_rng(d)
    n = 70_001
    xb = half_round(rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d))
    xq = half_round(rng.standard_normal((256, d)).astype(np.float32))
    idx = _index(gpu, xb, d)
    D100, I100 = idx.search(xq, 100)            # k = 100: the sample launch + 2 chunk launches (capi.hip chunk_schedule)
    assert all(len(set(r.tolist())) == 100 for r in I100), "a row was returned twice"   # (a row scoring differently in two launches could be)
    D10, I10 = idx.search(xq, 10)
    assert np.array_equal(I10, I100[:, :10]) and np.array_equal(D10, D100[:, :10])
    for sub in (1, 17, 130):
        Ds, Is = idx.search(xq[:sub], 100)
        assert np.array_equal(Is, I100[:sub]) and np.array_equal(Ds, D100[:sub])
    from oracle import oracle as O
    Dref, Iref = O.flat_search_ip(xb, xq[:32], 100)
    assert_topk_close(D100[:32], I100[:32], Dref, Iref)
