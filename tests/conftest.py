import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libvdl.so (hipcc cross-compiles on CPU) and the oracle once per session."""
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(ROOT, "mplan2vdl_amd", "lib", "libvdl.so")) or \
            not os.path.exists(os.path.join(ROOT, "oracle", "libvdl_oracle.so")):
        g.build()
    return True


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return f.read()


@pytest.fixture(scope="session")
def q6_text():
    return golden("q6.vdl")


@pytest.fixture(scope="session")
def q1_text():
    return golden("q1.vdl")
