import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Every test session runs the committed Makefiles first (they are incremental: seconds when nothing changed).
    `*.so` is git-ignored but travels to the GPU box with the snapshot, so "the file exists" says nothing about whether
    it was built from the sources under test."""
    import __graft_entry__ as g
    g.build()
    yield
