"""Device memory for the -m gpu tests without torch: ctypes on the HIP runtime libhyslam_amd.so itself is linked against
(the same libamdhip64 instance, already loaded by the library), so pointers can be handed to the `*_device` entry points."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from hyslam_amd import _native as N
        N.lib()                                         # loads libamdhip64 as a dependency
        L = C.CDLL("libamdhip64.so.7")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipDeviceSynchronize.argtypes = []
        L.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        L.hipStreamSynchronize.argtypes = [C.c_void_p]
        L.hipStreamDestroy.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with hipError %d" % (what, rc))


class DevBuf:
    """a hipMalloc'ed block; `.ptr` is the integer device address"""

    def __init__(self, nbytes, zero=True):
        self.nbytes = int(max(nbytes, 16))
        p = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(p), self.nbytes), "hipMalloc")
        self.ptr = p.value
        if zero:
            # hipMemset on device memory is ASYNCHRONOUS to the host and runs on the null stream; the library's streams are non-blocking, so a kernel launched
            # right after could finish BEFORE the memset and have its results zeroed (seen once in ~50 000 fuzz cases as a "mismatch" of all-zero outputs)
            _ok(hip().hipMemset(self.ptr, 0, self.nbytes), "hipMemset")
            _ok(hip().hipStreamSynchronize(None), "hipStreamSynchronize")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, zero=False)
        if a.nbytes:
            _ok(hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return b

    def to_numpy(self, dtype, count=None, offset=0):
        dt = np.dtype(dtype)
        n = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(n, dt)
        _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        if out.nbytes:
            _ok(hip().hipMemcpy(out.ctypes.data, self.ptr + offset, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def fill(self, byte):
        _ok(hip().hipMemset(self.ptr, byte, self.nbytes), "hipMemset")
        _ok(hip().hipStreamSynchronize(None), "hipStreamSynchronize")

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self):
        p = C.c_void_p()
        _ok(hip().hipStreamCreate(C.byref(p)), "hipStreamCreate")
        self.ptr = p.value

    def synchronize(self):
        _ok(hip().hipStreamSynchronize(self.ptr), "hipStreamSynchronize")

    def __del__(self):
        try:
            if self.ptr:
                hip().hipStreamDestroy(self.ptr)
                self.ptr = 0
        except Exception:
            pass


def sync():
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
