"""Device buffers owned by Python objects (the reference's ``_Data`` RAII holder,
raleigh/algebra/dense_cublas.py:801-811, with int64 sizes)."""

import ctypes

from ... import _lib


class DeviceBuffer:
    """nbytes of device memory released when the last Python reference dies."""

    def __init__(self, nbytes, zero=True):
        L = _lib.lib()
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _lib.check(L.rlh_malloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value or 0
        if zero and self.nbytes > 0:
            _lib.check(L.rlh_memset(self.ptr, 0, self.nbytes))

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', 0), 0
        if ptr:
            try:
                _lib.library().rlh_free(ptr)
            except Exception:       # interpreter shutdown
                pass


def upload(dptr, dpitch, array2d):
    """Copies a C-contiguous (rows, cols) host array to device rows of pitch
    dpitch bytes."""
    L = _lib.lib()
    rows, cols = array2d.shape
    width = cols * array2d.itemsize
    if rows == 0 or cols == 0:
        return
    if dpitch == width:
        _lib.check(L.rlh_h2d(dptr, _lib.host_ptr(array2d), rows * width))
    else:
        _lib.check(L.rlh_copy2d(dptr, dpitch, _lib.host_ptr(array2d), width, width, rows, 0))


def download(array2d, dptr, dpitch):
    L = _lib.lib()
    rows, cols = array2d.shape
    width = cols * array2d.itemsize
    if rows == 0 or cols == 0:
        return
    if dpitch == width:
        _lib.check(L.rlh_d2h(_lib.host_ptr(array2d), dptr, rows * width))
    else:
        _lib.check(L.rlh_copy2d(_lib.host_ptr(array2d), width, dptr, dpitch, width, rows, 1))
