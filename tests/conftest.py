import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The product libraries are built in-tree beforehand (__graft_entry__.build());
    only the oracle (test infrastructure) is built on demand."""
    from tests import oracle
    oracle.lib()
    yield
