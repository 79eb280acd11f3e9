import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(autouse=True)
def _route_keys_at_defaults(request):
    """The route keys of hta_set_tuning are process-global: a test that leaks one silently moves every later test onto
    another kernel.  GPU tests start and end with every key at its default (hta_reset_tuning)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from hamiltorch_amd import _abi
    _abi.reset_tuning()
    yield
    _abi.reset_tuning()
    _abi.set_tuning("profile", 0)
