import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "gpu_slow: long GPU soaks kept OUT of `-m gpu` (the driver's suite has a time limit): run by hand with "
                                       "`pytest -m gpu_slow` on a GPU box (tests/test_slow_gpu.py)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords or "gpu_slow" in it.keywords:
            it.add_marker(skip)


def pytest_sessionstart(session):
    import torch
    if torch.cuda.is_available() and not os.environ.get("PYTEST_XDIST_WORKER"):
        from tests.util import PARITY_REPORT
        try:
            os.remove(PARITY_REPORT)
        except OSError:
            pass
