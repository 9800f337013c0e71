"""Principal component analysis of a dense matrix on MI355X.

Counterpart of the reference's dense path pca -> LowerRankApproximation.compute ->
PartialSVD.compute -> block JCG on A_s^T A_s (raleigh/interfaces/pca.py:142-164,
lra.py:109-149, partial_svd.py:52-133, 238-301), restated for this repository's
Vectors / Matrix: the hot loop is the pair of dense products of ``_OperatorSVD.apply``
(rlh_dense_apply, MFMA for fp32) plus the usual block algebra.

Supported: a fixed number of components (``npc``) and the Frobenius-norm tolerance
(``tol`` with ``norm='f'``), with the mean shift; samples >= features or the transposed
case.  Out of scope: PCA update / incremental PCA, interactive stopping,
the 's' and 'm' norms.
"""

import math
import time

import numpy
import numpy.linalg as nla

from ..algebra.dense_matrix import AMatrix
from ..core.solver import Problem, Solver, Options


class _OperatorSVD:
    """x -> A_s^T A_s x (or A_s A_s^T x when there are fewer samples than features), A_s = A - e a
    the mean-shifted data (partial_svd.py:238-301)."""

    def __init__(self, matrix, v, transp=False, shift=False):
        self.op = matrix.as_operator()
        self.gpu = matrix.gpu()
        self.transp = transp
        self.shift = shift
        self.time = 0
        m, n = self.op.shape()
        # vectors are created by the OPERATOR so that a row-sharded matrix can hand out
        # sharded vectors for its row dimension and replicated ones for its column dimension
        self.w = self.op.new_vectors(n if transp else m, 0)
        if shift:
            dt = self.op.data_type()
            ones = numpy.ones((1, m), dtype=dt)
            self.ones = self.op.new_vectors(m, 1)
            self.ones.fill(ones)
            self.aves = self.op.new_vectors(n, 1)
            self.op.apply(self.ones, self.aves, transp=True)
            self.aves.scale(m * ones[0, :1])          # column means a

    def apply(self, x, y):
        m, n = self.op.shape()
        k = x.nvec()
        start = time.time()
        if self.transp:
            if self.w.nvec() < k:
                self.w = self.op.new_vectors(n, k)
            z = self.w
            z.select(k)
            self.op.apply(x, z, transp=True)
            if self.shift:
                z.add(self.aves, -1, x.dot(self.ones))
            self.op.apply(z, y)
            if self.shift:
                y.add(self.ones, -1, z.dot(self.aves))
        else:
            if self.w.nvec() < k:
                self.w = self.op.new_vectors(m, k)
            z = self.w
            z.select(k)
            self.op.apply(x, z)
            if self.shift:      # remove the mean along e, twice for accuracy
                z.add(self.ones, -1.0 / m, z.dot(self.ones))
                z.add(self.ones, -1.0 / m, z.dot(self.ones))
            self.op.apply(z, y, transp=True)
        if self.gpu is not None:
            self.gpu.synchronize()
        self.time += time.time() - start

    def mean_v(self):
        return self.aves if self.shift else None


class _SingularValueCriteria:
    """res^2 <= |lmd / lmd_max|^1.5 * svtol (lra.py:452-463)."""

    def __init__(self, tol):
        self.tolerance = tol

    def satisfied(self, solver, i):
        res = solver.convergence_data('residual', i)
        lmd = solver.convergence_data('eigenvalue', i)
        lmd_max = solver.convergence_data('max eigenvalue', i)
        return res >= 0 and res * res <= abs(lmd / lmd_max) ** 1.5 * self.tolerance


class _FrobeniusStopping:
    """Stop when ||A_s - L R||_F = sqrt(||A_s||_F^2 - sum sigma_i^2) <= eps
    (norm 'f' branch of truncated_svd.py:225-285)."""

    def __init__(self, frob2, tol, max_rank):
        self.f2 = frob2
        self.eps = tol * math.sqrt(frob2) if tol > 0 else -tol
        self.max_rank = max_rank
        self.ncon = 0

    def satisfied(self, solver):
        if solver.rcon <= self.ncon:
            return False
        lmd = solver.eigenvalues[self.ncon:solver.rcon]
        self.f2 -= float(numpy.sum(numpy.abs(lmd)))
        self.ncon = solver.rcon
        if self.max_rank > 0 and self.ncon >= self.max_rank:
            return True
        return math.sqrt(max(0.0, self.f2)) <= self.eps


class PartialSVD:
    """Leading singular triplets of A (optionally mean-shifted) via block JCG on the
    normal operator (partial_svd.py:19-160)."""

    def __init__(self, matrix, shift=False):
        self.__op = matrix.as_operator()
        m, n = matrix.shape()
        self.__transp = m < n
        self.__v = self.__op.new_vectors(m if self.__transp else n)
        self.__opsvd = _OperatorSVD(matrix, self.__v, self.__transp, shift)
        self.__shift = shift
        self.sigma = None
        self.iterations = -1
        self.u = self.v = None

    def op_svd(self):
        return self.__opsvd

    def vectors(self):
        return self.__v

    def compute(self, opt, nsv):
        op, v, transp, opSVD = self.__op, self.__v, self.__transp, self.__opsvd
        solver = Solver(Problem(v, opSVD))
        status = solver.solve(v, options=opt, which=(0, nsv))
        self.status = status
        if status < 0:
            return
        self.iterations = solver.iteration
        nv = v.nvec()
        M, N = op.shape()
        u = op.new_vectors(N if transp else M, nv)
        if nv < 1:
            self.sigma = numpy.zeros((0,), dtype=v.data_type())
            self.u, self.v = u, v
            return
        op.apply(v, u, transp)
        if self.__shift:            # u = A_s v (or A_s^T v)
            if not transp:
                u.add(opSVD.ones, -1, v.dot(opSVD.aves))
            else:
                u.add(opSVD.aves, -1, v.dot(opSVD.ones))
        sigma = numpy.sqrt(abs(u.dots(u)))
        u.scale(sigma)
        ind = numpy.argsort(-sigma)
        self.sigma = sigma[ind]
        for x in (u, v):
            w = x.new_vectors(nv)
            x.copy(w, ind)
            w.copy(x)
        self.u, self.v = u, v

    def left_v(self):
        return self.v if self.__transp else self.u

    def right_v(self):
        return self.u if self.__transp else self.v

    def mean_v(self):
        return self.__opsvd.mean_v()


def pca(A, npc=-1, tol=0, verb=0, arch='hip', norm='f', mpc=-1, svtol=1e-3, opt=None):
    '''PCA of the rows of A: returns (mean (1, n), trans (m, k), comps (k, n)) with
    trans @ comps ~ A - e mean, comps rows orthonormal, columns of trans in descending
    order of norm (raleigh/interfaces/pca.py:16-91).

    npc : number of components, or negative to use `tol`;
    tol : with npc < 0, stop when ||A_s - L R||_F <= tol ||A_s||_F (tol > 0) or <= -tol;
    mpc : cap on the number of components when tol is used;
    svtol : singular value tolerance relative to the largest one.'''
    if norm != 'f':
        raise ValueError("only the Frobenius norm ('f') stopping criterion is available")
    if opt is None:
        opt = Options()
    if hasattr(A, 'as_operator'):       # an AMatrix-like wrap, e.g. dist.ShardedAMatrix (rows sharded)
        matrix = A
    else:
        if not isinstance(A, numpy.ndarray) or not A.flags['C_CONTIGUOUS']:
            raise ValueError('matrix must be C_CONTIGUOUS')
        matrix = AMatrix(A, arch=arch)
    m, n = matrix.shape()
    psvd = PartialSVD(matrix, shift=True)
    user_bs, user_cc, user_sc = opt.block_size, opt.convergence_criteria, opt.stopping_criteria
    if user_bs < 1 and (npc < 0 or npc > 100):
        opt.block_size = 128
    if user_cc is None:
        opt.convergence_criteria = _SingularValueCriteria(svtol)
    if user_sc is None and npc < 0:
        if tol == 0:
            raise ValueError('either npc or tol must be given (interactive stopping is not available)')
        opSVD = psvd.op_svd()
        # ||A_s||_F^2 = sum_i ||a_i||^2 - m ||mean||^2
        frob2 = matrix.frobenius2() - m * float(numpy.abs(opSVD.aves.dots(opSVD.aves))[0])
        opt.stopping_criteria = _FrobeniusStopping(frob2, tol, mpc)
    try:
        psvd.compute(opt, npc)
    finally:
        opt.block_size, opt.convergence_criteria, opt.stopping_criteria = user_bs, user_cc, user_sc
    if psvd.status < 0:
        raise RuntimeError('block JCG failed with status %d' % psvd.status)
    left, right = psvd.left_v(), psvd.right_v()
    left.scale(psvd.sigma, multiply=True)
    k = left.nvec()
    if npc > 0:
        k = min(k, npc)
    elif mpc > 0:
        k = min(k, mpc)
    left.select(k)
    right.select(k)
    pca.last = {'iterations': psvd.iterations, 'operator_time': psvd.op_svd().time, 'sigma': psvd.sigma[:k]}
    return psvd.mean_v().data(), left.data().T, right.data()


def pca_error(data, mean, trans, comps):
    """Relative max-row and Frobenius errors of the PCA approximation (pca.py:167-175)."""
    ones = numpy.ones((data.shape[0], 1), dtype=data.dtype)
    data_s = data - numpy.dot(ones, numpy.reshape(mean, (1, comps.shape[1])))
    err = numpy.dot(trans, comps) - data_s
    rows = lambda a: numpy.sqrt(numpy.sum(numpy.abs(a) ** 2, axis=1))
    em = numpy.amax(rows(err)) / numpy.amax(rows(data_s))
    ef = nla.norm(err, ord='fro') / nla.norm(data_s, ord='fro')
    return em, ef
