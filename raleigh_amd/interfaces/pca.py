"""Principal component analysis of a dense matrix on MI355X.

Counterpart of the reference's dense path pca -> LowerRankApproximation.compute ->
PartialSVD.compute -> block JCG on A_s^T A_s (raleigh/interfaces/pca.py:142-164,
lra.py:109-149, partial_svd.py:52-133, 238-301), restated for this repository's
Vectors / Matrix: the hot loop is the pair of dense products of ``_OperatorSVD.apply``
(rlh_dense_apply, MFMA for fp32) plus the usual block algebra.

Supported: a fixed number of components (``npc``) or a tolerance (``tol``) on the Frobenius
norm ('f'), the largest singular value ('s') or the largest row ('m') of the remainder, with
the mean shift; samples >= features or the transposed case; the update of an existing
approximation with new samples (``have``) and incremental PCA (``batch_size``), see lra.py.
Out of scope: interactive stopping.
"""

import math
import time

import numpy
import numpy.linalg as nla

from ..algebra.dense_matrix import AMatrix
from ..core.solver import Problem, Solver, Options


def _project_out(x, basis):
    """x -= basis (basis^H x) for an orthonormal basis: one Gram and one block update."""
    x.add(basis, -1.0, x.dot(basis))


class _OperatorSVD:
    """The normal operator of the mean-shifted data A_s = A - e a^T (e: ones over the M samples, a: the
    N column means) for the block eigensolver: x -> A_s^T (A_s x) when M >= N, x -> A_s (A_s^T x)
    otherwise (the role of raleigh/interfaces/partial_svd.py:238-301).

    Written from the algebra.  With c = a^T x (k numbers per block):
        M >= N:   z = A x - e c^T        (centred: e^T z = 0 up to rounding, so A_s^T z = A^T z)
                  y = A^T z
        M <  N:   z = A^T x - a s^T,  s = e^T x
                  y = A z - e t^T,    t = a^T z
    On a Matrix with `apply_r1` every rank-one term rides in the epilogue of its GEMM
    (rlh_dense_apply_r1) and the coefficient vectors are formed on the device by a one-column Gram, so
    an application is two GEMMs and one or two small reductions, with no host synchronisation and no
    extra pass over the M x k or N x k block (the reference makes two dot + add passes per product).
    Other operators (row-sharded data) take the same steps with dot / add calls."""

    def __init__(self, matrix, v, transp=False, shift=False, mean=None, deflate=None):
        self.op = matrix.as_operator()
        self.gpu = matrix.gpu()
        self.transp = transp
        self.shift = shift
        self.time = 0
        # `mean`: a given row a (one vector of dimension N) instead of the column means of A -- then
        # e^T (A - e a) != 0 and both products carry their rank-one term.  `deflate` = (R, C): R an
        # orthonormal basis (vectors of dimension N) and C = A_s R^H (vectors of dimension M); the operator
        # becomes that of E = A_s (I - R^H R), the part of the data an existing set of components R does
        # not describe (lra.py: update).  Both need the fused products.
        self.own_mean = mean is None
        self.deflate = deflate
        m, n = self.op.shape()
        # vectors are created by the OPERATOR so that a row-sharded matrix can hand out
        # sharded vectors for its row dimension and replicated ones for its column dimension
        self.w = self.op.new_vectors(n if transp else m, 0)
        # (a row-sharded matrix folds the rank-one term into the non-transposed product only: dist.ShardedDenseMatrix)
        self._fused = hasattr(self.op, 'apply_r1') and (not transp or getattr(self.op, 'r1_transposed', True))
        self._coef = None
        if (mean is not None or deflate is not None) and not (self._fused and shift):
            raise ValueError('a given mean / a deflated operator need a Matrix with apply_r1 and shift')
        if mean is not None and not getattr(self.op, 'r1_transposed', True):
            # with a GIVEN mean e^T z != 0, so y = A_s^H z needs the rank-one term in the TRANSPOSED product, which a
            # row-sharded matrix does not fold in: said here, at construction, not in the middle of a solve with
            # collectives in flight
            raise ValueError('a given mean needs a Matrix that takes a rank-one term in the transposed product '
                             '(a row-sharded matrix does not)')
        if shift:
            dt = self.op.data_type()
            self.ones = self.op.new_vectors(m, 1)
            self.ones.fill(numpy.ones((1, m), dtype=dt))
            if mean is not None:
                self.aves = mean
            else:
                self.aves = self.op.new_vectors(n, 1)
                self.op.apply(self.ones, self.aves, transp=True)     # A^H e = M conj(a)
                self.aves.scale(numpy.full((1,), m, dtype=dt))
                if self.aves.is_complex():
                    self.aves.conjugate()                            # a itself
            # <x, conj(a)> = a^T x: the one-column Gram conjugates its second argument
            self.aves_c = self.aves
            if self.aves.is_complex():
                self.aves_c = self.aves.clone()
                self.aves_c.conjugate()

    def _coefficients(self, k, slot):
        from ..algebra.hip.memory import DeviceBuffer
        es = numpy.dtype(self.op.data_type()).itemsize
        if self._coef is None or self._coef.nbytes < 2 * k * es:
            self._coef = DeviceBuffer(2 * max(k, 16) * es, zero=False)
        return self._coef.ptr + slot * k * es

    def forward(self, x, z):
        """z = A_s x for vectors x of dimension N (fused products only)."""
        from ..algebra.hip.matrix import coefficients_into
        c0 = self._coefficients(x.nvec(), 0)
        coefficients_into(c0, x, self.aves_c)                             # c = a^T x
        self.op.apply_r1(x, z, False, None, c0)                           # z = A x - e c^T

    def backward(self, z, y):
        """y = A_s^T z for vectors z of dimension M (fused products only)."""
        from ..algebra.hip.matrix import coefficients_into
        c1 = self._coefficients(z.nvec(), 1)
        coefficients_into(c1, z, self.ones)                               # s = e^T z
        self.op.apply_r1(z, y, True, self.aves_c, c1)                     # y = A^H z - conj(a) s^T

    def apply(self, x, y):
        m, n = self.op.shape()
        k = x.nvec()
        start = time.time()
        if self.w.nvec() < k or self.w.shape()[0] < k:
            self.w = self.op.new_vectors(n if self.transp else m, k)
        z = self.w
        z.select(k)
        if not self.shift:
            self.op.apply(x, z, transp=self.transp)
            self.op.apply(z, y, transp=not self.transp)
        elif self._fused:
            from ..algebra.hip.matrix import coefficients_into
            c0, c1 = self._coefficients(k, 0), self._coefficients(k, 1)
            if self.transp:
                coefficients_into(c0, x, self.ones)                       # s = e^T x
                self.op.apply_r1(x, z, True, self.aves_c, c0)             # z = A^H x - conj(a) s^T
                if self.deflate is not None:
                    _project_out(z, self.deflate[0])                      # z = (I - R^H R) z
                coefficients_into(c1, z, self.aves_c)                     # t = a^T z
                self.op.apply_r1(z, y, False, None, c1)                   # y = A z - e t^T
            else:
                coefficients_into(c0, x, self.aves_c)                     # c = a^T x
                self.op.apply_r1(x, z, False, None, c0)                   # z = A x - e c^T
                if self.deflate is not None:                              # z = A_s (I - R^H R) x = A_s x - C (R x)
                    z.add(self.deflate[1], -1.0, x.dot(self.deflate[0]))
                if self.own_mean:
                    self.op.apply(z, y, transp=True)                      # y = A^T z (e^T z = 0)
                else:
                    self.backward(z, y)
                if self.deflate is not None:
                    _project_out(y, self.deflate[0])
        elif self.transp:
            self.op.apply(x, z, transp=True)
            z.add(self.aves_c, -1.0, x.dot(self.ones))                    # - conj(a) (e^T x)
            self.op.apply(z, y)
            y.add(self.ones, -1.0, z.dot(self.aves_c))                    # - e (a^T z)
        else:
            self.op.apply(x, z)
            for _ in range(2):          # the mean of every column of z along e, removed twice for accuracy
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

    def __init__(self, matrix, shift=False, mean=None, deflate=None):
        self.__op = matrix.as_operator()
        m, n = matrix.shape()
        self.__transp = m < n
        self.__v = self.__op.new_vectors(m if self.__transp else n)
        self.__opsvd = _OperatorSVD(matrix, self.__v, self.__transp, shift, mean, deflate)
        self.__shift = shift
        self.sigma = None
        self.iterations = -1
        self.u = self.v = None

    def op_svd(self):
        return self.__opsvd

    def vectors(self):
        return self.__v

    def compute(self, opt, nsv, refine=False):
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
        if opSVD.deflate is not None or not opSVD.own_mean:
            if not transp:          # v spans a subspace of range(I - R^H R) up to rounding: make it exact
                if opSVD.deflate is not None:
                    _project_out(v, opSVD.deflate[0])
                opSVD.forward(v, u)
            else:
                opSVD.backward(v, u)
                if opSVD.deflate is not None:
                    _project_out(u, opSVD.deflate[0])
        else:
            op.apply(v, u, transp)
            if self.__shift:            # u = A_s v (or A_s^T v)
                if not transp:
                    u.add(opSVD.ones, -1, v.dot(opSVD.aves_c))     # - e (a^T v)
                else:
                    u.add(opSVD.aves_c, -1, v.dot(opSVD.ones))     # - conj(a) (e^T v)
        if refine and nv > 1:
            # u = A_s^(T) v is orthogonal only as far as v has converged (svtol): make it orthonormal and rotate
            # v with it, A_s^(T) (v q) = u sigma.  (What pca.py:146-147 asks for when samples < features -- the
            # components are then u; partial_svd.py:99-110.  The reference's own route through _finalize_svd
            # stops at `nv = min(32, nsv/2)`, a float slice bound under Python 3, partial_svd.py:204-206.)
            sigma, q = u.svd()
            w = v.new_vectors(nv)
            v.multiply(numpy.ascontiguousarray(q), w)
            w.copy(v)
            self.sigma = sigma
            self.u, self.v = u, v
            return
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


def pca(A, npc=-1, tol=0, have=None, batch_size=None, verb=0, arch='hip', norm='f', mpc=-1, svtol=1e-3, opt=None):
    '''PCA of the rows of A: returns (mean (1, n), trans (m, k), comps (k, n)) with
    trans @ comps ~ A - e mean, comps rows orthonormal, columns of trans in descending
    order of norm (raleigh/interfaces/pca.py:16-164).

    npc : number of components, or negative to use `tol`;
    tol : with npc < 0, stop when the norm of A_s - L R is at most tol times that of A_s (tol > 0) or -tol;
    norm : 'f' Frobenius, 's' largest singular value, 'm' largest row norm;
    have : (mean0, trans0, comps0) of data A0 seen earlier -- the result then describes
        numpy.concatenate((A0, A)); with neither npc nor tol, as many components as comps0 has;
    batch_size : incremental PCA, `batch_size` rows of A in HBM at a time;
    mpc : cap on the number of components when tol is used;
    svtol : singular value tolerance relative to the largest one.'''
    from .lra import LowerRankApproximation, _as_matrix
    if norm not in ('f', 's', 'm'):
        raise ValueError('norm %s is not supported' % repr(norm))
    if opt is None:
        opt = Options()
    lra = LowerRankApproximation(have)
    if batch_size is None:
        matrix = _as_matrix(A, arch)        # an ndarray, or an AMatrix-like wrap, e.g. dist.ShardedAMatrix
        if have is None:
            m, n = matrix.shape()
            lra.compute(matrix, opt=opt, rank=npc, tol=tol, norm=norm, max_rank=mpc, svtol=svtol, shift=True, verb=verb,
                        refine=m < n)
        else:
            lra.update(matrix, opt=opt, rank=npc, tol=tol, norm=norm, max_rank=mpc, svtol=svtol, verb=verb)
    else:
        if not isinstance(A, numpy.ndarray) or not A.flags['C_CONTIGUOUS']:
            raise ValueError('matrix must be C_CONTIGUOUS')
        lra.icompute(A, batch_size, opt=opt, rank=npc, tol=tol, norm=norm, max_rank=mpc, svtol=svtol, shift=True,
                     verb=verb, arch=arch)
    pca.last = {'iterations': lra.iterations, 'operator_time': lra.operator_time, 'sigma': lra.sigma}
    return lra.mean(), lra.left(), lra.right()


def pca_error(data, mean, trans, comps):
    """Relative max-row and Frobenius errors of the PCA approximation (pca.py:167-175)."""
    ones = numpy.ones((data.shape[0], 1), dtype=data.dtype)
    data_s = data - numpy.dot(ones, numpy.reshape(mean, (1, comps.shape[1])))
    err = numpy.dot(trans, comps) - data_s
    rows = lambda a: numpy.sqrt(numpy.sum(numpy.abs(a) ** 2, axis=1))
    em = numpy.amax(rows(err)) / numpy.amax(rows(data_s))
    ef = nla.norm(err, ord='fro') / nla.norm(data_s, ord='fro')
    return em, ef
