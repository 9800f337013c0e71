"""Symmetric indefinite factorisation P A P^T = L D L^H (rlh_ldlt_factor: host multifrontal factorisation with
1 x 1 / 2 x 2 threshold pivoting and delayed pivots) and its device solve.

Counterpart of the reference's PARDISO wrapper, raleigh/algebra/mkl_wrap.py:354-489 (mtype -2 / -4: real symmetric /
Hermitian indefinite; 2 / 4 with pos_def), which raleigh/algebra/sparse_mkl.py:51-119 drives: one triangle is
factorised, the inertia comes from D.
"""

import ctypes

import numpy as np

from ... import _lib

INFO = ('nnz_l', 'negative', 'positive', 'perturbed', 'two_by_two', 'delayed', 'max_front', 'supernodes', 'multiply_adds',
        'forced')


class SymmetricFactors:
    """L (strictly lower CSR, unit diagonal implied, pivot order), the block diagonal D (diag, subdiag, block) and the
    pivot order of one factorisation; `info` as rlh_ldlt_info reports it."""

    def __init__(self, lower, diag, subdiag, block, order, info, upper=None):
        self.lower, self.diag, self.subdiag, self.block, self.order, self.info = lower, diag, subdiag, block, order, info
        self.upper = upper                  # L^H as the library hands it out (None: transposed here when needed)

    def lower_transposed(self):
        """L^H (strictly upper CSR): the operator of the backward solve."""
        if self.upper is None:
            self.upper = self.lower.conj().T.tocsr()
        return self.upper

    def inertia(self):
        """(negative, positive) eigenvalue counts of A, from D (Sylvester's law)."""
        return int(self.info['negative']), int(self.info['positive'])

    def block_diagonal(self):
        """D as a SciPy matrix (tests, diagnostics)."""
        import scipy.sparse as scs
        first = np.flatnonzero(self.block == 1)
        e = self.subdiag[first]
        n = len(self.diag)
        return (scs.diags(self.diag) + scs.csr_matrix((e, (first + 1, first)), shape=(n, n))
                + scs.csr_matrix((np.conj(e), (first, first + 1)), shape=(n, n))).tocsr()

    def inverse_rows(self, dtype):
        """D^-1 as rlh_bdiag_solve takes it: (coef (n, 2), shift int32) with
        x'[i] = coef[i, 0] x[i] + coef[i, 1] x[i + shift[i]]."""
        n = len(self.diag)
        d = self.diag.astype(np.complex128 if np.iscomplexobj(self.diag) else np.float64)
        coef = np.zeros((n, 2), dtype=d.dtype)
        shift = np.zeros(n, dtype=np.int32)
        one = self.block == 0
        coef[one, 0] = 1.0 / d[one]
        first = np.flatnonzero(self.block == 1)
        if len(first):
            d11, d22, d21 = d[first].real, d[first + 1].real, self.subdiag[first]
            det = d11 * d22 - np.abs(d21) ** 2
            coef[first, 0] = d22 / det
            coef[first, 1] = -np.conj(d21) / det
            coef[first + 1, 0] = d11 / det
            coef[first + 1, 1] = -d21 / det
            shift[first] = 1
            shift[first + 1] = -1
        dt = np.dtype(dtype)
        if dt.kind != 'c' and np.iscomplexobj(coef):
            raise ValueError('complex factors cannot be applied to real vectors')
        return np.ascontiguousarray(coef, dtype=dt), shift


def ldlt(matrix, perm=None, pivot_threshold=0.01, perturb=1e-13):
    """Factorises a real symmetric / Hermitian SciPy sparse matrix (its upper triangle is read; a matrix given by its
    lower triangle alone is mirrored first).

    perm: optional fill-reducing ordering (perm[new] = old), default: the library's minimum degree.
    pivot_threshold: u of the threshold partial pivoting (0 = none: positive definite matrices).
    Returns SymmetricFactors."""
    import scipy.sparse as scs
    a = scs.csr_matrix(matrix)
    if a.shape[0] != a.shape[1]:
        raise ValueError('the matrix must be square')
    up = scs.triu(a, format='csr')
    if up.nnz <= a.shape[0] and a.nnz > up.nnz and scs.triu(a, 1).nnz == 0:
        up = scs.csr_matrix(scs.tril(a).conj().T)      # only the LOWER triangle was given: the same matrix
    a = up
    a.sum_duplicates()
    a.sort_indices()
    n = a.shape[0]
    cplx = np.iscomplexobj(a.data)
    dt = np.complex128 if cplx else np.float64
    indptr = np.ascontiguousarray(a.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(a.indices, dtype=np.int32)
    values = np.ascontiguousarray(a.data, dtype=dt)
    p = None
    if perm is not None:
        p = np.ascontiguousarray(perm, dtype=np.int64)
        if p.shape != (n,):
            raise ValueError('perm must have one entry per row')
    L = _lib.library()                         # host-only entry points: no device needed
    f = ctypes.c_void_p()
    _lib.check(L.rlh_ldlt_factor(ctypes.byref(f), _lib.DTYPE_CODE[dt], n, _lib.host_ptr(indptr), _lib.host_ptr(indices),
                                 _lib.host_ptr(values), _lib.host_ptr(p) if p is not None else None,
                                 float(pivot_threshold), float(perturb)))
    try:
        raw = np.zeros(len(INFO), dtype=np.int64)
        _lib.check(L.rlh_ldlt_info(f, _lib.host_ptr(raw)))
        info = dict(zip(INFO, (int(v) for v in raw)))
        nnz = info['nnz_l']
        ip = np.zeros(n + 1, dtype=np.int64)
        ix = np.empty(max(nnz, 1), dtype=np.int32)
        va = np.empty(max(nnz, 1), dtype=dt)
        up, ux, uv = np.zeros(n + 1, dtype=np.int64), np.empty(max(nnz, 1), dtype=np.int32), np.empty(max(nnz, 1), dtype=dt)
        _lib.check(L.rlh_ldlt_get_transposed(f, _lib.host_ptr(up), _lib.host_ptr(ux), _lib.host_ptr(uv)))
        d = np.zeros(max(n, 1), dtype=dt)
        e = np.zeros(max(n, 1), dtype=dt)
        blk = np.zeros(max(n, 1), dtype=np.int8)
        order = np.zeros(max(n, 1), dtype=np.int64)
        _lib.check(L.rlh_ldlt_get(f, _lib.host_ptr(ip), _lib.host_ptr(ix), _lib.host_ptr(va), _lib.host_ptr(d),
                                  _lib.host_ptr(e), _lib.host_ptr(blk), _lib.host_ptr(order)))
    finally:
        L.rlh_ldlt_destroy(f)
    lower = scs.csr_matrix((va[:nnz], ix[:nnz], ip), shape=(n, n))
    upper = scs.csr_matrix((uv[:nnz], ux[:nnz], up), shape=(n, n))
    lower.has_sorted_indices = upper.has_sorted_indices = True
    return SymmetricFactors(lower, d[:n], e[:n], blk[:n], order[:n], info, upper)


class SymmetricSolve:
    """x = A^-1 b on the device from SymmetricFactors: L^-1 (one persistent launch, rows gathered in pivot order),
    D^-1 (rlh_bdiag_solve), L^-H (one persistent launch, rows scattered back)."""

    def __init__(self, factors, dtype):
        from .memory import DeviceBuffer
        from .precond import TriangularChain
        self._dtype = np.dtype(dtype).type
        if np.dtype(dtype).kind != 'c' and np.iscomplexobj(factors.lower.data):
            raise ValueError('complex factors cannot be applied to real vectors')
        self._n = factors.lower.shape[0]
        def forward():
            return TriangularChain([(factors.lower, True, True)], self._dtype, perm_in=factors.order)

        def backward():
            return TriangularChain([(factors.lower_transposed(), False, True)], self._dtype, perm_out=factors.order)
        _lib.lib()
        if factors.lower.nnz > 1_000_000:
            # the host side of rlh_sptrsv_create is serial per operator and the two are independent (ctypes releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=2) as ex:
                made = [ex.submit(forward), ex.submit(backward)]
                self._forward, self._backward = made[0].result(), made[1].result()
        else:
            self._forward, self._backward = forward(), backward()
        coef, shift = factors.inverse_rows(self._dtype)
        L = _lib.lib()
        self._coef = DeviceBuffer(max(coef.nbytes, 1), zero=False)
        self._shift = DeviceBuffer(max(shift.nbytes, 1), zero=False)
        if self._n:
            _lib.check(L.rlh_h2d(self._coef.ptr, _lib.host_ptr(coef), coef.nbytes))
            _lib.check(L.rlh_h2d(self._shift.ptr, _lib.host_ptr(shift), shift.nbytes))
        self.nnz = self._forward.nnz + self._backward.nnz
        self.levels = self._forward.levels + self._backward.levels

    def size(self):
        return self._n

    def solve(self, b, x):
        self._forward.solve(b, x)
        _lib.check(_lib.lib().rlh_bdiag_solve(_lib.dtype_code(self._dtype), self._n, self._coef.ptr, self._shift.ptr,
                                              x.nvec(), x.data_ptr(), x.ld()))
        self._backward.solve(x, x)

    apply = solve
