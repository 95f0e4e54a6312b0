"""Shared helpers for the parity tests (oracle = checker, ragroute_amd = thing under test)."""
import numpy as np


def half_round(x, dtype="fp16"):
    """Round f32 values to the storage dtype and back, so oracle and GPU see identical inputs."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.to(torch.float16 if dtype == "fp16" else torch.bfloat16).to(torch.float32)
    return t.numpy()


def int_data(rng, n, d, lo=-2, hi=3):
    """Small-integer embeddings: every dot product is an exact f32 integer in any summation order,
    so GPU scores must equal the oracle's bit for bit and ties are plentiful."""
    return rng.integers(lo, hi, size=(n, d)).astype(np.float32)


def assert_topk_close(D, I, Dref, Iref, S=None, tol=1e-3):
    """Tolerance-aware parity: scores within tol position by position; ids identical wherever the
    oracle's ranking is unambiguous at tol (neighbouring gaps > 2*tol), and otherwise every returned
    id must be a legitimate member (its oracle score within tol of the position's score)."""
    assert D.shape == Dref.shape and I.shape == Iref.shape
    fin = np.isfinite(Dref)
    assert np.array_equal(np.isfinite(D), fin)
    assert np.allclose(D[fin], Dref[fin], atol=tol, rtol=0), float(np.abs(D[fin] - Dref[fin]).max())
    assert np.array_equal(I[~fin], Iref[~fin])
    nq, k = I.shape
    bad = 0
    for q in range(nq):
        if np.array_equal(I[q], Iref[q]):
            continue
        for j in range(k):
            if I[q, j] == Iref[q, j]:
                continue
            # ambiguous only if a neighbour of the oracle's position j is within 2*tol
            lo = Dref[q, j - 1] if j > 0 else np.inf
            hi = Dref[q, j + 1] if j + 1 < k else (-np.inf if S is None else np.partition(S[q], -k - 1)[-k - 1] if S.shape[1] > k else -np.inf)
            near = (lo - Dref[q, j] <= 2 * tol) or (Dref[q, j] - hi <= 2 * tol)
            assert near, f"query {q} pos {j}: id {I[q, j]} != {Iref[q, j]} with unambiguous oracle gap"
            if S is not None:
                assert abs(S[q, I[q, j]] - D[q, j]) <= tol
            bad += 1
    return bad
