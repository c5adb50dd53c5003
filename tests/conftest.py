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


class HipBuffers:
    """Device buffers for tests that drive the device-resident ABI entries without PyTorch (what a Rust host does):
    plain hipMalloc / hipMemcpy through the HIP runtime the product library already loaded."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so.7")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def alloc(self, nbytes, fill=0):
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), nbytes) == 0
        assert self.hip.hipMemset(p, fill, nbytes) == 0
        # hipMemset of device memory is asynchronous to the host and ordered on the NULL stream only: the library's streams are
        # non-blocking, so a kernel of theirs could otherwise write this buffer BEFORE the fill runs (seen once the suite's order
        # changed: a metrics block zeroed after the search that filled it)
        assert self.hip.hipDeviceSynchronize() == 0
        self.ptrs.append(p)
        return p.value

    def upload(self, arr):
        import numpy as np
        a = np.ascontiguousarray(arr)
        p = self.alloc(a.nbytes)
        assert self.hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    def download(self, ptr, nbytes, dtype):
        import numpy as np
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0
        return out

    def close(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []


@pytest.fixture
def hipbuf(gpu):
    b = HipBuffers()
    yield b
    b.close()
