import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so) — the checker, built on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gm():
    """The product: the C-ABI HIP library.  No fallback: a missing .so fails the test run."""
    import gnss_sdr_rs_amd
    gnss_sdr_rs_amd.lib()
    return gnss_sdr_rs_amd


@pytest.fixture(scope="session")
def gpu(gm):
    from gnss_sdr_rs_amd import _lib
    _lib.init(0)
    return gm
