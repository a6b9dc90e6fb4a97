import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_builds():
    """Make sure libkrotov_hip.so (hipcc, gfx950) and the C oracle exist and are current: `make` is
    incremental, so this is a no-op when the tree was built by __graft_entry__.build()."""
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="session")
def oracle_c():
    import oracle_c as oc
    oc.build()
    return oc
