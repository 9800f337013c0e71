"""MI355X implementation of RALEIGH's dense Matrix operator.

Mirrors raleigh/algebra/dense_cublas.py:635-776 (constructor from ndarray or
Vectors, shape/order/data_type/dots/new_vectors/apply) with the GEMM of
``apply`` running in librlhip.so (rlh_dense_apply, MFMA for fp32).
"""

import numpy as np

from ... import _lib
from .memory import DeviceBuffer, upload
from .vectors import Vectors, _padded


class Matrix:

    def __init__(self, arg):
        if isinstance(arg, Vectors):
            f, m = arg.selected()
            self._shape = (m, arg.dimension())
            self._dtype = arg.data_type()
            self._order = 'C_CONTIGUOUS'
            self._lda = arg.ld()
            self._buf = arg.vectors_data()
            self._off = arg.data_ptr() - (self._buf.ptr if self._buf is not None else 0)
        elif isinstance(arg, np.ndarray):
            if arg.ndim != 2:
                raise ValueError('Matrix data must be a 2D array')
            self._shape = arg.shape
            self._dtype = arg.dtype.type
            if arg.flags['C_CONTIGUOUS']:
                self._order = 'C_CONTIGUOUS'
                rows, cols, host = arg.shape[0], arg.shape[1], arg
            elif arg.flags['F_CONTIGUOUS']:
                self._order = 'F_CONTIGUOUS'
                rows, cols, host = arg.shape[1], arg.shape[0], arg.T
            else:
                raise ValueError('Matrix data must be either C- or F-contiguous')
            if self._dtype not in _lib.DTYPE_CODE:
                raise ValueError('data type %s not supported' % repr(self._dtype))
            self._lda = _padded(cols)
            es = arg.itemsize
            self._buf = DeviceBuffer(max(rows, 1) * self._lda * es)
            self._off = 0
            upload(self._buf.ptr, self._lda * es, host)
        else:
            raise ValueError('wrong argument %s in Matrix constructor' % repr(type(arg)))
        self._code = _lib.DTYPE_CODE[self._dtype]
        self._es = _lib.DTYPE_SIZE[self._dtype]

    def data_ptr(self):
        return (self._buf.ptr if self._buf is not None else 0) + self._off

    def matrix_data(self):
        return self._buf

    def lda(self):
        return self._lda

    def order(self):
        return self._order

    def shape(self):
        return self._shape

    def data_type(self):
        return self._dtype

    def data_size(self):
        return self._es

    def is_complex(self):
        return self._dtype in (np.complex64, np.complex128)

    def fill(self, data):
        host = data if self._order == 'C_CONTIGUOUS' else data.T
        upload(self.data_ptr(), self._lda * self._es, np.ascontiguousarray(host, dtype=self._dtype))

    def dots(self):
        v = Vectors(self, shallow=True)
        return v.dots(v)

    def absmax(self):
        """Largest modulus of the (real / imaginary parts of the) entries, computed on the device."""
        import ctypes
        rows = self._shape[0] if self._order == 'C_CONTIGUOUS' else self._shape[1]
        cols = self._shape[1] if self._order == 'C_CONTIGUOUS' else self._shape[0]
        out = ctypes.c_double()
        _lib.check(_lib.lib().rlh_absmax(self._code, cols, rows, self.data_ptr(), self._lda, ctypes.byref(out)))
        return float(out.value)

    def new_vectors(self, dim=None, nv=0):
        if dim is None:
            dim = self.shape()[1]
        return Vectors(dim, nv, self.data_type())

    def apply(self, x, y, transp=False):
        self.apply_r1(x, y, transp)

    def apply_r1(self, x, y, transp=False, u=None, c=None):
        """y = Op(A) x - u c^T with the rank-one term folded into the product's epilogue (rlh_dense_apply_r1):
        `c` a device pointer to x.nvec() coefficients (e.g. written by `coefficients_into`), `u` a Vectors
        window of ONE vector of y's dimension or None for a vector of ones.  c None: the plain product."""
        if x.data_type() != self._dtype or y.data_type() != self._dtype:
            raise ValueError('Matrix and vectors data types differ')
        m, n = self._shape
        if transp:
            if n != y.dimension() or m != x.dimension():
                raise ValueError('Matrix and vectors dimensions incompatible')
        else:
            if m != y.dimension() or n != x.dimension():
                raise ValueError('Matrix and vectors dimensions incompatible')
        k = x.nvec()
        if k != y.nvec():
            raise ValueError('Numbers of input and output vectors differ')
        if u is not None and (u.nvec() != 1 or u.dimension() != y.dimension() or c is None):
            raise ValueError('the rank-one term needs one vector of the output dimension and coefficients')
        _lib.check(_lib.lib().rlh_dense_apply_r1(
            self._code, m, n, self.data_ptr(), self._lda,
            0 if self._order == 'C_CONTIGUOUS' else 1, 1 if transp else 0,
            k, x.data_ptr(), x.ld(), y.data_ptr(), y.ld(),
            None if u is None else u.data_ptr(), c))


def coefficients_into(buf_ptr, x, w):
    """buf[j] = <x_j, w> = w^H x_j for the selected vectors of x and the ONE vector w, written to DEVICE
    memory at buf_ptr without a host synchronisation (rlh_gram with a device output): the coefficients
    of a rank-one epilogue."""
    if w.nvec() != 1 or w.dimension() != x.dimension():
        raise ValueError('one vector of the same dimension is needed')
    n = getattr(x, 'local_dimension', x.dimension)()
    _lib.check(_lib.lib().rlh_gram(x._code, n, x.nvec(), x.data_ptr(), x.ld(), 1, w.data_ptr(), w.ld(), buf_ptr, None))
