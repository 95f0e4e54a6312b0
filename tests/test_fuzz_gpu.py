"""GPU: a bounded, seeded sample of tests/fuzz_cases.py - every row-width class, batch sizes around the 16 / 64 / 128 / 256
query-block boundaries, corpus sizes around the tile / sample / chunk boundaries, both dtypes, IP and L2 - on integer data,
corpus and queries flush against unmapped pages.  tools/fuzz_parity.py runs the same generator for as long as one likes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _fuzz_in_child(args, gen="cases", run="run_case"):
    """The cases run on the guard-page layout, where an over-read is a GPU fault that kills the process: a child process turns
    that into ONE failed test (with the case it died on in the output) instead of an aborted suite."""
    code = (f"import torch\nfrom tests.fuzz_cases import {gen}, {run}\n"
            f"failed = []\nfor c in {gen}({args}):\n    print('case', c, flush=True)\n    if not {run}(c, torch.device('cuda:0')): failed.append(c)\n"
            "print('failed', failed)\nassert not failed\nprint('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0 and res.stdout.rstrip().endswith("ok"), (res.returncode, res.stdout[-1500:], res.stderr[-2000:])


def test_seeded_fuzz_sample(gpu):
    _fuzz_in_child("20261004, 150")


def test_seeded_fuzz_sample_large_corpora(gpu):
    """The same on 70 K ... 2 M-row corpora: bootstrap sample, 2 ... 4 chunk launches and their compactions at every width class."""
    _fuzz_in_child("20261005, 16, max_work=3e10, big=True")



def test_seeded_fuzz_segmented_search(gpu):
    """Random segment tables, widths, batch sizes and route masks for the one-pass search over several sources, bit-exact against
    the oracle chain on the guard-page layout."""
    _fuzz_in_child("20261006, 60", "segment_cases", "run_segment_case")
