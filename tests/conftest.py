import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fs():
    """The product package (ctypes over libfluid_hip.so). Builds it if the .so is missing."""
    if not os.path.exists(os.path.join(entry.PKG_DIR, "libfluid_hip.so")):
        entry.build()
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement — the checker, never the product."""
    return entry.load_oracle()


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d
