import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "aaai2023-pvd_amd")
for p in (REPO, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib_built():
    """The HIP library must exist in-tree (it cross-compiles without a GPU)."""
    so = os.path.join(PKG, "libpvd_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()
    assert os.path.exists(so)
    return so
