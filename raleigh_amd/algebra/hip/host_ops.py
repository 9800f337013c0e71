"""Operators adjacent to the hot path whose FACTORISATION runs on the host.

The reference's shift-invert operator is PARDISO on the host (raleigh/algebra/sparse_mkl.py:51-119).
Here the library's own symmetric indefinite factorisation (`ldlt.py`: P A P^T = L D L^H, 1 x 1 / 2 x 2
pivots, inertia from D) runs once on the host; every solve then runs on the device (`SymmetricSolve`:
L, D^-1 and L^H on the whole n x m block in HBM), so a solver iteration moves no block across PCIe.
`method='superlu'` keeps the earlier rounds' unsymmetric SuperLU factors (`TriangularChain`).  `HostOperator` adapts a
caller's own host operator (two block transfers per application: the caller's choice, never a
stand-in for a HIP kernel -- every Vectors operation still runs in librlhip.so).
"""

import numpy as np
import scipy.sparse as scs
import scipy.sparse.linalg as sla


class HostOperator:
    """Adapts a host operator ``f(x_host (m, n)) -> y_host (m, n)`` (or an object with
    ``apply(x_host, y_host)``, the reference's preconditioner protocol,
    raleigh/interfaces/partial_hevp.py:41-49) to device Vectors."""

    def __init__(self, op):
        self._op = op

    def apply(self, x, y):
        xh = x.data()
        if hasattr(self._op, 'apply'):
            yh = np.zeros_like(xh)
            self._op.apply(xh, yh)
        else:
            yh = np.ascontiguousarray(self._op(xh), dtype=xh.dtype)
        y.fill(yh)


def _lib_error():
    from ... import _lib
    return _lib.RlhError


class SparseSymmetricSolver:
    """Factorisation of A - sigma B (counterpart of sparse_mkl.py:51-119 / mkl_wrap.py:354-489: PARDISO there).

    method='ldlt' (default): P (A - sigma B) P^T = L D L^H by the library's multifrontal factorisation on the host
    (one triangle, 1 x 1 and 2 x 2 pivots with delayed pivoting: saddle-point matrices and shifts that leave zeros
    on the diagonal factorise; `pos_def=True` switches the pivoting off like PARDISO's mtype 2 / 4; `pivot_threshold`
    defaults to 0.01, 0.1 for single precision blocks); the inertia the
    caller needs to map `which` (partial_hevp.py:172-194) is read off D.  A numerically singular matrix (a pivot
    had to be perturbed) raises at factorize() like the reference's "near singular matrix?" exit.
    method='superlu': SuperLU's unsymmetric LU in symmetric mode (both triangles; the inertia from diag(U) only
    when SuperLU kept the symmetric pivot order).

    The SOLVES run on the device (`device=True`, the default): the factors become sparse triangular operators on
    the whole n x m block in HBM (algebra/hip/ldlt.py SymmetricSolve, algebra/hip/precond.py TriangularChain), so a
    solver iteration moves no block across PCIe.  `device=False` solves on the host (two block transfers per
    application; SuperLU factors only)."""

    def __init__(self, dtype=np.float64, pos_def=False, device=True, method='ldlt', pivot_threshold=None):
        if method not in ('ldlt', 'superlu'):
            raise ValueError('method must be ldlt or superlu')
        if method == 'ldlt' and not device:
            raise ValueError('the L D L^H factors are applied on the device: device=False needs method=superlu')
        self._dtype = dtype
        self._lu = None
        self._factors = None
        self._method = method
        self._pos_def = bool(pos_def)
        if pivot_threshold is None:
            # threshold u of the pivoting: entries of L are bounded by 1 / u.  The factors are computed in double
            # precision whatever the blocks are; applied to single precision blocks, u = 0.01 costs two to three digits
            # of the seven on hard indefinite matrices (measured backward error 5e-4 against 3e-5 with u = 0.1)
            pivot_threshold = 0.1 if np.dtype(dtype).itemsize == (8 if np.dtype(dtype).kind == 'c' else 4) else 0.01
        self._threshold = 0.0 if pos_def else float(pivot_threshold)
        self._device = bool(device)
        self._chain = None
        self._work = None

    def analyse(self, a, sigma=0, b=None):
        a = scs.csc_matrix(a)
        if sigma != 0:
            if b is None:
                b = scs.identity(a.shape[0], dtype=a.dtype, format='csc')
            a = a - sigma * b
        self._a = scs.csc_matrix(a)
        self._n = a.shape[0]
        self._sigma = sigma

    def factorize(self):
        self._chain = None
        if self._method == 'ldlt':
            from .ldlt import ldlt
            try:
                f = ldlt(self._a, pivot_threshold=self._threshold)
            except _lib_error() as e:
                raise RuntimeError('factorization failed (%s)' % e)
            if f.info['perturbed'] or (self._pos_def and f.info['negative']):
                raise RuntimeError('factorization failed (near singular matrix?)')
            self._factors = f
            return
        try:
            self._lu = sla.splu(self._a, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0,
                                options=dict(SymmetricMode=True))
        except Exception:
            raise RuntimeError('factorization failed (near singular matrix?)')
        self._chain = None

    def _device_chain(self, dtype=None):
        """P_r A P_c = L U  =>  x = P_c U^-1 L^-1 P_r b: the scratch row r is row argsort(perm_r)[r] of b,
        and goes to row argsort(perm_c)[r] of x.  One chain per block data type (the factorisation may be wider
        than the blocks: analyse() promotes when sigma is a Python float on float32 data)."""
        dtype = np.dtype(self._dtype if dtype is None else dtype).type
        if self._chain is None:
            self._chain = {}
        if dtype not in self._chain and self._method == 'ldlt':
            from .ldlt import SymmetricSolve
            self._chain[dtype] = SymmetricSolve(self._factors, dtype)
        if dtype not in self._chain:
            from .precond import TriangularChain
            lu = self._lu
            lower = scs.tril(scs.csr_matrix(lu.L), -1, format='csr')        # SuperLU stores the unit diagonal
            upper = scs.csr_matrix(lu.U)
            if np.dtype(dtype).kind != 'c' and np.iscomplexobj(upper.data):
                raise ValueError('complex factors cannot be applied to real vectors')
            self._chain[dtype] = TriangularChain([(lower, True, True), (upper, False, False)], dtype,
                                                 np.argsort(lu.perm_r), np.argsort(lu.perm_c))
        return self._chain[dtype]

    def solve(self, b, x):
        if self._device:
            try:
                chain = self._device_chain(b.data_type())
            except _lib_error() as e:
                if 'memory' not in str(e).lower() or self._method == 'ldlt':
                    raise                       # (a missing library or GPU is an error, never a reason to solve on the host)
                # factors that do not fit next to the blocks in HBM: the host solve below, loudly
                import warnings
                warnings.warn('triangular factors could not be placed on the device (%s): solving on the host' % e)
                self._device = False
            else:
                if hasattr(b, 'comm'):
                    # row-sharded blocks (BASELINE config 5's layout): the factors are replicated, so every rank gathers the
                    # block on ITS GPU (one all_gather on the kernels' stream), runs the chain there and keeps its rows --
                    # no block leaves the devices (round 2 solved the gathered block on every rank's host cores)
                    from .vectors import Vectors
                    m = b.nvec()
                    w = self._work
                    if w is None or w[0].shape()[0] < m or w[0].data_type() != b.data_type():
                        w = self._work = [Vectors(self._n, m, data_type=b.data_type()) for _ in range(2)]
                    for v in w:
                        v.select(m)
                    b.gather_into(w[0])
                    chain.solve(w[0], w[1])
                    x.take_rows_of(w[1])
                else:
                    chain.solve(b, x)
                return
        bh = b.data()
        x.fill(np.ascontiguousarray(self._lu.solve(np.ascontiguousarray(bh.T)).T, dtype=bh.dtype))

    def apply(self, b, x):
        self.solve(b, x)

    def inertia(self):
        """(negative, positive) eigenvalue counts of A - sigma B: from D of the L D L^H factors (1 x 1 pivots by
        sign, 2 x 2 pivots by determinant and trace), or -- method='superlu' -- from the signs of diag(U), which is

        only valid when SuperLU kept the symmetric pivot order (perm_r == perm_c), so that
        U = D L^H: `diag_pivot_thresh=0` still interchanges rows off an exactly zero diagonal
        (saddle-point or unluckily shifted matrices), and then the sign count is not the inertia.
        That case, and a zero pivot, raise -- partial_hevp maps them to status -1 like the
        reference's "factorization too inaccurate" exit (partial_hevp.py:147-156)."""
        if self._method == 'ldlt':
            return self._factors.inertia()
        lu = self._lu
        if not np.array_equal(lu.perm_r, lu.perm_c):
            raise RuntimeError('unsymmetric pivoting in the factorization of A - sigma B: inertia unavailable, '
                               'consider moving the shift slightly')
        d = np.real(lu.U.diagonal())
        if np.any(d == 0) or not np.all(np.isfinite(d)):
            raise RuntimeError('zero pivot in the factorization of A - sigma B: consider moving the shift slightly')
        neg = int(np.sum(d < 0))
        return neg, int(self._n - neg)

    def size(self):
        return self._n

    def data_type(self):
        return self._dtype

    def sigma(self):
        return self._sigma

    def factors(self):
        """The SymmetricFactors of method='ldlt' (None before factorize())."""
        return self._factors
