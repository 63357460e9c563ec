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


@pytest.fixture(params=["reference", "stratified"])
def lens_mode(request):
    """One-to-one pixel comparisons run under both lens samplers: "reference" = the reference's two independent lens draws
    (implementation.rs:648-653; oracle ORC_RNG_COUNTER, product RT_OPT_LENS_SAMPLER 0) and "stratified" = the product's
    default sampler (oracle ORC_RNG_COUNTER_STRATIFIED, the one labelled product extension of the oracle)."""
    return request.param
