import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def solver():
    """One mpc_hip_handle on cuda:0 for the whole session.  No CPU fallback: fails if there is no device."""
    import mpc_amd
    s = mpc_amd.Solver(0)
    yield s
    s.close()
