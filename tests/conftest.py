import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def built():
    """Make sure the host library and the oracle are built (CPU-only artefacts)."""
    import __graft_entry__ as ge
    ge.build(device=False)
    return True
