"""MI355X sparse symmetric/Hermitian operator for SciPy sparse matrices.

Counterpart of raleigh/algebra/sparse_mkl.py:16-48 (SparseSymmetricMatrix) and
:143-154 (Operator): same constructor argument (a SciPy sparse matrix whose
UPPER triangle defines the symmetric/Hermitian operator, as MKL's 'SUNF'/'HUNF'
descriptor does), same ``apply(x, y)``, but x and y are device Vectors and the
product runs in librlhip.so (rlh_spmm) on a sliced-ELL copy of the full matrix.
"""

import ctypes

import numpy as np
import scipy.sparse as scs

from ... import _lib


def full_from_upper(matrix):
    """A = U + U^H - diag(U) from U = triu(matrix): what the reference's MKL call
    computes with (raleigh/algebra/mkl_wrap.py:211-276)."""
    u = scs.triu(matrix, format='csr')
    s = scs.triu(matrix, k=1, format='csr')
    full = scs.csr_matrix(u + s.conj().T)
    full.sort_indices()
    return full


class CsrOperator:
    """Device CSR operator: y = A x for rows owned by this process.  upper=True: the Hermitian operator defined by
    the upper triangle of `csr` (entries below the diagonal ignored), mirrored inside the library
    (rlh_csr_create_upper) instead of by three SciPy passes on the host."""

    def __init__(self, csr, n_own=None, upper=False):
        csr = scs.csr_matrix(csr)
        if not csr.has_canonical_format:
            # an assembled matrix may carry duplicate or unsorted entries: a COPY is put into canonical form (equal
            # columns summed, indices sorted) -- the caller's own object is never modified
            csr = csr.copy()
            csr.sum_duplicates()
        dt = csr.data.dtype.type
        if dt not in _lib.DTYPE_CODE:
            raise ValueError('unsupported data type')
        self._dtype = dt
        self._shape = csr.shape
        self._nnz = int(csr.nnz)
        self._n_own = csr.shape[1] if n_own is None else int(n_own)
        indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
        values = np.ascontiguousarray(csr.data)
        h = ctypes.c_void_p()
        if upper:
            if csr.shape[0] != csr.shape[1] or n_own is not None:
                raise ValueError('an operator given by its upper triangle is square and unsharded')
            _lib.check(_lib.lib().rlh_csr_create_upper(ctypes.byref(h), _lib.DTYPE_CODE[dt], csr.shape[0], _lib.host_ptr(indptr),
                                                       _lib.host_ptr(indices), _lib.host_ptr(values)))
            nnz, dummy = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(_lib.lib().rlh_csr_info(h, ctypes.byref(dummy), ctypes.byref(dummy), ctypes.byref(nnz), ctypes.byref(dummy)))
            self._nnz = int(nnz.value)
        else:
            _lib.check(_lib.lib().rlh_csr_create(ctypes.byref(h), _lib.DTYPE_CODE[dt], csr.shape[0],
                                                 csr.shape[1], _lib.host_ptr(indptr),
                                                 _lib.host_ptr(indices), _lib.host_ptr(values)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.library().rlh_csr_destroy(h)
            except Exception:
                pass

    def shape(self):
        return self._shape

    def nnz(self):
        return self._nnz

    def data_type(self):
        return self._dtype

    def layout(self):
        """('sell' | 'well' | 'wide', stored entry slots, staged elements per slot): the device layout
        the library chose (rlh_csr_layout)."""
        lay, stored, ratio = ctypes.c_int(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(_lib.lib().rlh_csr_layout(self._h, ctypes.byref(lay), ctypes.byref(stored), ctypes.byref(ratio)))
        return ('sell', 'well', 'wide')[lay.value], int(stored.value), float(ratio.value)

    def stacks(self):
        """(stacks, staged elements per row and vector without / with them): the stacked row blocks of the windowed
        layout (rlh_csr_stacks); 0 stacks when the layout was not built."""
        n, a, b = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib().rlh_csr_stacks(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b)))
        return int(n.value), float(a.value), float(b.value)

    def bf16_ready(self, ldh=0):
        """Whether the bfloat16 Chebyshev step takes this operator (rlh_csr_bf16_ready); ldh: the leading dimension of
        the halo block of a row shard (ignored without halo columns)."""
        ok = ctypes.c_int()
        _lib.check(_lib.lib().rlh_csr_bf16_ready(self._h, self._n_own, int(ldh), ctypes.byref(ok)))
        return bool(ok.value)

    def apply_ptr(self, m, x_ptr, ldx, y_ptr, ldy, halo_ptr=None, ldh=0, part=0):
        """part 0: all rows; 1: the rows that need no halo column; 2: the others (rlh_spmm_part)."""
        _lib.check(_lib.lib().rlh_spmm_part(self._h, part, m, x_ptr, ldx, self._n_own, halo_ptr, ldh, y_ptr, ldy))

    def cheb_step_bf16(self, m, y, p, b, cy, cp, cb, halo_ptr=None, ldh=0, part=0):
        _lib.check(_lib.lib().rlh_spmm_cheb_bf16_part(self._h, part, m, y.ptr(), y.ld, self._n_own, halo_ptr, ldh,
                                                      p.ptr(), p.ld, b.ptr(), b.ld, float(cy), float(cp), float(cb)))

    def cheb_step_ptr(self, m, y, p, b, cy, cp, cb, halo_ptr=None, ldh=0, part=0):
        """p = cy y + cp p + cb (b - A y) in one pass (y, p, b: Vectors windows; p updated in place)."""
        _lib.check(_lib.lib().rlh_spmm_cheb_part(self._h, part, m, y.data_ptr(), y.ld(), self._n_own, halo_ptr, ldh,
                                                 p.data_ptr(), p.ld(), b.data_ptr(), b.ld(),
                                                 float(cy), float(cp), float(cb)))


class Bf16Block:
    """A block of vectors stored as bfloat16 (device memory, column-major, leading dimension a
    multiple of 8): work storage of the polynomial preconditioner, not a Vectors type."""

    def __init__(self, n, m):
        from .memory import DeviceBuffer
        self.n, self.m = int(n), int(m)
        self.ld = -(-self.n // 64) * 64
        self._buf = DeviceBuffer(max(self.m, 1) * self.ld * 2, zero=False)

    def ptr(self):
        return self._buf.ptr

    def zero(self, m):
        _lib.check(_lib.lib().rlh_memset(self._buf.ptr, 0, ((m - 1) * self.ld + self.n) * 2))

    def pack(self, x, scale=1.0):
        """self[:, :m] = bf16(scale * x) for a float32 / float64 Vectors window x."""
        _lib.check(_lib.lib().rlh_bf16_pack(x._code, self.n, x.nvec(), x.data_ptr(), x.ld(), float(scale),
                                            self._buf.ptr, self.ld))

    def unpack(self, y):
        _lib.check(_lib.lib().rlh_bf16_unpack(y._code, self.n, y.nvec(), self._buf.ptr, self.ld, y.data_ptr(), y.ld()))


class SparseSymmetricMatrix:

    def __init__(self, matrix):
        # the UPPER triangle of `matrix` defines the operator (sparse_mkl.py:16-40); the library mirrors it, so the
        # matrix goes in as it comes (both triangles or one) and triu() is only taken if somebody asks for csr()
        try:
            self.__given = matrix.csr()
        except Exception:
            self.__given = scs.csr_matrix(matrix)
        self.__upper = None
        self.__op = CsrOperator(self.__given, upper=True)

    def size(self):
        return self.__given.shape[0]

    def data_type(self):
        return self.__given.data.dtype

    def csr(self):
        if self.__upper is None:
            self.__upper = scs.triu(self.__given, format='csr')
            self.__upper.sort_indices()
        return self.__upper

    def nnz_full(self):
        return self.__op.nnz()

    def apply(self, x, y):
        if x.data_type() != self.__op.data_type() or y.data_type() != self.__op.data_type():
            raise ValueError('Matrix and vectors data types differ')
        if x.dimension() != self.size() or y.dimension() != self.size():
            raise ValueError('Matrix and vectors dimensions incompatible')
        if x.nvec() != y.nvec():
            raise ValueError('Numbers of input and output vectors differ')
        self.__op.apply_ptr(x.nvec(), x.data_ptr(), x.ld(), y.data_ptr(), y.ld())

    def cheb_step(self, y, p, b, cy, cp, cb):
        """Fused step of the three-term Chebyshev semi-iteration: p = cy y + cp p + cb (b - A y)."""
        self.__op.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb)

    def cheb_step_bf16(self, m, y, p, b, cy, cp, cb):
        """The same step on bfloat16 blocks (Bf16Block), float32 arithmetic; float32 operators in
        the windowed layout only (the library reports an error otherwise)."""
        self.__op.cheb_step_bf16(m, y, p, b, cy, cp, cb)

    def supports_bf16(self):
        return self.__op.bf16_ready()

    def layout(self):
        """Device layout of the operator and its stacked row blocks (diagnostics: CsrOperator.layout, .stacks)."""
        return self.__op.layout() + self.__op.stacks()


class Operator:
    """Wraps a user operator with ``apply(x, y)`` acting on device Vectors
    (raleigh/algebra/sparse_mkl.py:143-154 passes host arrays instead)."""

    def __init__(self, op):
        self.__op = op

    def apply(self, x, y):
        self.__op.apply(x, y)
