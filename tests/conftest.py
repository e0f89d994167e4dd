import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import ctypes as C
        from hyslam_amd import _native as N
        n = C.c_int()
        return N.lib().hs_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must fail loudly (not skip) when selected with -m gpu on a box whose HIP path is broken."""
    from hyslam_amd import _native as N
    N.lib()   # raises if the extension is missing
    if not _gpu_available():
        pytest.fail("no usable HIP device: the -m gpu suite needs a real MI355X and the built libhyslam_amd.so")
    return True
