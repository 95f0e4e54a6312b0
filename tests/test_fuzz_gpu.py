"""GPU: a bounded, seeded sample of tests/fuzz_cases.py - every row-width class, batch sizes around the 16 / 64 / 128 / 256
query-block boundaries, corpus sizes around the tile / sample / chunk boundaries, both dtypes, IP and L2 - on integer data,
corpus and queries flush against unmapped pages.  tools/fuzz_parity.py runs the same generator for as long as one likes."""
import pytest

from tests.fuzz_cases import cases, run_case

pytestmark = pytest.mark.gpu


def test_seeded_fuzz_sample(gpu):
    failed = [c for c in cases(20261004, 150) if not run_case(c, gpu)]
    assert not failed, failed


def test_seeded_fuzz_sample_large_corpora(gpu):
    """The same on 70 K ... 2 M-row corpora: bootstrap sample, 2 ... 4 chunk launches and their compactions at every width class."""
    failed = [c for c in cases(20261005, 16, max_work=3e10, big=True) if not run_case(c, gpu)]
    assert not failed, failed
