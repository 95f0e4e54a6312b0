import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from ragroute_amd import _lib
    _lib.lib()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_gc_settings():
    """DataSource.start() tunes the garbage collector of its (service) process (DataSource.tune_runtime: gc.freeze + thresholds);
    tests that run a service loop in the pytest process must not leave that behind for the tests after them."""
    import gc
    before = gc.get_threshold()
    yield
    if gc.get_threshold() != before:
        gc.set_threshold(*before)
        gc.unfreeze()
