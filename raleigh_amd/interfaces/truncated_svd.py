"""Truncated SVD of a dense matrix on MI355X: A V = U S with orthonormal U, V and the k largest
singular values.

Counterpart of raleigh/interfaces/truncated_svd.py:24-127 (truncated_svd), :130-203 (row-wise error
of the truncation) and :206-283 (stopping criteria), over this repository's PartialSVD: block JCG on
A^T A (or A A^T when there are fewer rows than columns) -- two dense products (rlh_dense_apply, MFMA
for fp32) per iteration -- then U (or V) from one more product, orthonormalised on the device
(Vectors.svd) with the other factor rotated to match.

Stopping without a given number of values (nsv < 0) needs tol != 0: 's' |sigma_k| <= tol sigma_0,
'f' Frobenius norm of the remainder, 'm' largest row norm of the remainder, relative for tol > 0,
absolute (-tol) for tol < 0.  Interactive stopping (tol = 0) is not offered.
"""

import math

import numpy

from ..algebra.dense_matrix import AMatrix
from ..core.solver import Options
from .pca import PartialSVD


class _VectorErrorCriteria:
    """'kinematic vector error' <= vtol (truncated_svd.py:385-392)."""

    def __init__(self, tol):
        self.tolerance = tol

    def satisfied(self, solver, i):
        err = solver.convergence_data('kinematic vector error', i)
        return err >= 0 and err <= self.tolerance


class _RowErrors:
    """Norms of the rows of A_ - U S V^H as singular triplets arrive (A_ = A, or A - e a with the mean shift):
    |row_i|^2 minus the squares of the row's coefficients (A_ v_j)_i (truncated_svd.py:130-203), accumulated
    on the device side by `dots(transp=True)`."""

    def __init__(self, matrix, psvd, shift=False):
        self.op = matrix.as_operator()
        self.m, self.n = matrix.shape()
        self.shifted = psvd.op_svd() if shift else None
        self.err2 = numpy.abs(matrix.dots()).astype(numpy.float64)
        if shift:               # |a_i - a|^2 = |a_i|^2 - 2 Re(a_i . conj(a)) + |a|^2
            if not hasattr(self.op, 'apply_r1'):
                raise ValueError("norm 'm' with the mean shift needs a Matrix with apply_r1")
            aves = self.shifted.aves
            b = self.op.new_vectors(self.m, 1)
            self.op.apply(self.shifted.aves_c, b)
            s = float(numpy.abs(aves.dots(aves))[0])
            self.err2 = numpy.maximum(self.err2 - 2 * numpy.real(b.data()[0]).astype(numpy.float64) + s, 0.0)
        self.initial = math.sqrt(float(numpy.amax(self.err2))) if self.m else 0.0
        self.ncon = 0

    def _forward(self, x, y):
        if self.shifted is not None:
            self.shifted.forward(x, y)
        else:
            self.op.apply(x, y)

    def _backward(self, x, z):
        if self.shifted is not None:
            self.shifted.backward(x, z)
        else:
            self.op.apply(x, z, transp=True)

    def update(self, x):
        ncon = x.nvec()
        new = ncon - self.ncon
        if new < 1:
            return
        sel = x.selected()
        x.select(new, sel[0] + self.ncon)
        if self.m < self.n:         # x: left vectors; (A_ A_^H x_j)_i conj(x_j)_i = sigma_j^2 |u_ij|^2
            z = self.op.new_vectors(self.n, new)
            self._backward(x, z)
            y = self.op.new_vectors(self.m, new)
            self._forward(z, y)
            q = numpy.real(x.dots(y, transp=True))
        else:                       # x: right vectors; |(A_ v_j)_i|^2
            y = self.op.new_vectors(self.m, new)
            self._forward(x, y)
            q = numpy.real(y.dots(y, transp=True))
        x.select(sel[1], sel[0])
        self.err2 = numpy.maximum(self.err2 - numpy.maximum(q, 0.0), 0.0)
        self.ncon = ncon

    def largest(self):
        return math.sqrt(float(numpy.amax(self.err2)))


class _TruncationStopping:
    """The three norms of truncated_svd.py:206-283 without the interactive branch."""

    def __init__(self, matrix, psvd, tol, norm, max_nsv, verb, shift=False, frob2=None):
        self.tol, self.norm, self.max_nsv, self.verb = tol, norm, max_nsv, verb
        self.ncon = 0
        self.sigma0 = None
        self.f2 = self.frob = None
        self.rows = None
        if norm == 'f':
            self.f2 = matrix.frobenius2() if frob2 is None else frob2
            self.frob = math.sqrt(self.f2)
        elif norm == 'm':
            self.rows = _RowErrors(matrix, psvd, shift)

    def satisfied(self, solver):
        if solver.rcon <= self.ncon:
            return False
        lmd = solver.eigenvalues[self.ncon:solver.rcon]
        sigma = -numpy.sort(-numpy.sqrt(numpy.abs(lmd)))
        if self.sigma0 is None:
            self.sigma0 = float(sigma[0])
        if self.norm == 'f':
            self.f2 -= float(numpy.sum(sigma * sigma))
            err_abs = math.sqrt(max(0.0, self.f2))
            err_rel = err_abs / self.frob if self.frob > 0 else 0.0
        elif self.norm == 'm':
            self.rows.update(solver.eigenvectors)
            err_abs = self.rows.largest()
            err_rel = err_abs / self.rows.initial if self.rows.initial > 0 else 0.0
        else:
            err_abs = float(sigma[-1])
            err_rel = err_abs / self.sigma0 if self.sigma0 > 0 else 0.0
        self.ncon = solver.rcon
        if self.verb > 0:
            print('sigma[%d] = %.2e*sigma[0], truncation error = %.2e' % (self.ncon - 1, sigma[-1] / self.sigma0, err_rel))
        done = err_rel <= self.tol if self.tol > 0 else err_abs <= -self.tol
        return done or (self.max_nsv > 0 and self.ncon >= self.max_nsv)


def truncated_svd(A, opt=None, nsv=-1, tol=0, norm='s', msv=-1, vtol=0, arch='hip', verb=0):
    '''Returns u (m, k), sigma (k,) in descending order and vt (k, n) with A vt^H = u diag(sigma), u and
    vt^H orthonormal (raleigh/interfaces/truncated_svd.py:24-127).

    nsv : number of singular values, or negative to stop by `tol`;
    tol, norm : with nsv < 0, stop when the `norm` ('s', 'f' or 'm') of A - u diag(sigma) vt is at most
        tol times that of A (tol > 0) or -tol (tol < 0);
    msv : cap on the number of values when tol is used;
    vtol : singular vector error tolerance (default sqrt(machine epsilon)).'''
    if norm not in ('s', 'f', 'm'):
        raise ValueError('norm %s is not supported' % repr(norm))
    if opt is None:
        opt = Options()
    if hasattr(A, 'as_operator'):
        matrix = A
    else:
        if not isinstance(A, numpy.ndarray) or A.ndim != 2:
            raise ValueError('a 2D array is needed')
        matrix = AMatrix(numpy.ascontiguousarray(A), arch=arch)
    psvd = PartialSVD(matrix)
    user_bs, user_cc, user_sc = opt.block_size, opt.convergence_criteria, opt.stopping_criteria
    if user_bs < 1 and (nsv < 0 or nsv > 100):
        opt.block_size = 128
    if user_cc is None:
        if vtol <= 0:
            vtol = math.sqrt(numpy.finfo(matrix.data_type()).eps)
        opt.convergence_criteria = _VectorErrorCriteria(vtol)
    if user_sc is None and nsv < 0:
        if tol == 0:
            raise ValueError('either nsv or tol must be given (interactive stopping is not available)')
        opt.stopping_criteria = _TruncationStopping(matrix, psvd, tol, norm, msv, verb)
    try:
        psvd.compute(opt, nsv, refine=True)
    finally:
        opt.block_size, opt.convergence_criteria, opt.stopping_criteria = user_bs, user_cc, user_sc
    if psvd.status < 0:
        raise RuntimeError('block JCG failed with status %d' % psvd.status)
    left, right, sigma = psvd.left_v(), psvd.right_v(), psvd.sigma
    k = left.nvec()             # every converged triplet, nsv or a few more, as the reference returns them
    if msv > 0:                 # (truncated_svd.py:112-116: only msv cuts)
        k = min(k, msv)
    left.select(k)
    right.select(k)
    truncated_svd.last = {'iterations': psvd.iterations, 'operator_time': psvd.op_svd().time}
    # rows of vt = the right vectors themselves, as the reference returns them (v.T, truncated_svd.py:127)
    return left.data().T, sigma[:k], right.data()
