"""Inexact shift-invert: (A - sigma B)^-1 applied by a preconditioned block MINRES on the device blocks.

The reference's shift-invert operator is a direct PARDISO factorisation on the host
(raleigh/algebra/sparse_mkl.py:51-119, raleigh/algebra/mkl_wrap.py:350-559).  A direct factor of BASELINE
config 5 (complex128, n = 126^3, 3-D) has ~10^9 entries and cannot be replicated on eight GPUs; here the
same `analyse / factorize / solve(b, x)` surface is served by an iterative solve that touches the blocks
only through the Vectors operations and `op.apply` -- so it runs unchanged on row-sharded blocks
(algebra/hip/dist.py: every reduction is one all-reduce of an m x m matrix, the operator exchanges its
halo) and moves no block across PCIe.

`block_minres` is the block form of Paige & Saunders' MINRES with a Hermitian positive definite
preconditioner M: a block Lanczos process in the M inner product
    K U_j = V_{j-1} beta_j^H + V_j alpha_j + V_{j+1} beta_{j+1},    U_j = M V_j,   U_i^H V_j = delta_ij I,
and a QR factorisation of the block tridiagonal matrix, updated with one 2-block unitary per step; the
iterate minimises the M-norm of the residual over the block Krylov space.  All right-hand sides share one
space, which is what makes it the right tool inside a block eigensolver: the block of 64 right-hand sides
resolves the ~64 eigenvalues of K nearest zero in a few steps (they are what makes K ill-conditioned next
to an interior shift), where 64 independent MINRES runs would each fight all of them.  Directions whose
M-norm falls below `drop` times the largest of their block are removed (the block narrows), so converged
or dependent right-hand sides cost nothing and never divide by a vanishing norm.  Only m x m matrices
reach the host (two reductions per step), as in the eigensolver itself.
"""

import numpy as np
import scipy.linalg as sla

from ...core.solver import _single_threaded_blas


class ShiftedOperator:
    """y = (A - sigma B) x on device blocks (B None: identity)."""

    def __init__(self, a, sigma=0.0, b=None):
        self._a, self._b, self._sigma = a, b, sigma
        self._t = None

    def apply(self, x, y):
        self._a.apply(x, y)
        if self._sigma == 0:
            return
        if self._b is None:
            y.add(x, -self._sigma)
            return
        m = x.nvec()
        t = self._t
        if t is None or t.shape()[0] < m:
            t = self._t = x.new_vectors(m)
        t.select(m)
        self._b.apply(x, t)
        y.add(t, -self._sigma)


def _split(g, drop):
    """G = beta^H beta for a Hermitian positive semi-definite G (k x k) with the directions below drop^2 of
    the largest eigenvalue removed: returns beta (r x k) and its right inverse on the kept space (k x r)."""
    g = 0.5 * (g + g.conj().T)
    lam, q = sla.eigh(g)
    top = lam[-1] if lam.size else 0.0
    if not np.isfinite(top):
        raise FloatingPointError('block MINRES: non-finite Gram matrix (is the preconditioner positive definite?)')
    # (the eigenvalues of a computed Gram matrix are only known to a few eps of the largest one: what lies below is
    # rounding noise, and a noise direction scaled to unit norm is not orthogonal to anything)
    keep = lam > max(top, 0.0) * max(drop * drop, 64 * np.finfo(g.dtype).eps)
    if top <= 0:
        keep[:] = False
    lam, q = lam[keep][::-1], q[:, keep][:, ::-1]
    root = np.sqrt(lam)
    return (q * root).conj().T, q / root


class MinresInfo:
    """What the last solve did (iterations = operator applications, each on the current block width)."""

    def __init__(self):
        self.iterations = 0
        self.residuals = None
        self.columns_applied = 0
        self.converged = False
        self.negative = None

    def _count_negative(self, alphas, betas):
        """Negative eigenvalues of the block tridiagonal Lanczos matrix T_k = U^H K U: by Cauchy interlacing a
        LOWER bound on the number of negative eigenvalues of K, reached as soon as the block Krylov space holds
        the negative invariant subspace (extreme, well separated eigenvalues of the preconditioned operator: a
        few steps of a block of that many vectors).  The three-term recurrence loses orthogonality once a Ritz
        pair has converged, and T_k then carries further copies of it.  The copies are told from genuine values
        by the FIRST-block rows of the Ritz vectors (the start block's coordinates of each Ritz vector, the
        quantity Cullum & Willoughby's test looks at): copies of one eigenvector have parallel rows -- or rows
        of weight ~1e-18 -- while distinct eigenvectors, multiple eigenvalues included, have independent ones
        for a random start block.  The count is therefore the numerical RANK of those rows over all negative
        Ritz values, which cannot exceed the width of the start block: a count that reaches the width says the
        probe was too narrow (IterativeSymmetricSolver.inertia widens it)."""
        sizes = [a.shape[0] for a in alphas]
        off = np.concatenate(([0], np.cumsum(sizes)))
        t = np.zeros((off[-1], off[-1]), dtype=alphas[0].dtype)
        for j, a in enumerate(alphas):
            t[off[j]:off[j + 1], off[j]:off[j + 1]] = a
            if j + 1 < len(alphas):
                bt = betas[j]                            # (r_{j+1} x r_j)
                t[off[j + 1]:off[j + 2], off[j]:off[j + 1]] = bt
                t[off[j]:off[j + 1], off[j + 1]:off[j + 2]] = bt.conj().T
        lam, vec = sla.eigh(t)
        first = vec[:sizes[0], lam < 0]
        self.negative_ritz = int(first.shape[1])
        self.ritz_values = int(lam.size)
        sv = np.linalg.svd(first, compute_uv=False) if first.size else np.zeros((0,))
        self.negative = int(np.sum(sv > 1e-3 * sv[0])) if sv.size else 0
        self.probe_width = int(sizes[0])


def block_minres(op, b, x, precond=None, tol=1e-8, max_iter=500, work=None, drop=1e-7, count_negative=False):
    """Solves op x = b for all selected vectors of b at once; x (same selection width) is overwritten, the
    start is zero.  op: Hermitian operator with apply(u, w); precond: Hermitian positive definite operator
    with apply(u, w) or None; tol: per-column bound on ||r||_M / ||b||_M (a scalar or one value per column).
    work: a dict that keeps the work blocks between calls; count_negative: also count the negative eigenvalues
    of the projected operator (MinresInfo.negative).  Returns a MinresInfo."""
    # (the m x m LAPACK / BLAS work is latency-bound: on a many-core host the thread pool costs 10x its serial time)
    with _single_threaded_blas():
        return _block_minres(op, b, x, precond, tol, max_iter, work, drop, count_negative)


def _block_minres(op, b, x, precond, tol, max_iter, work, drop, count_negative):
    m = b.nvec()
    info = MinresInfo()
    if m < 1:
        info.converged = True
        return info
    dt = b.data_type()
    if work is None:
        work = {}
    nblocks = 9 if precond is not None else 6
    blocks = work.get('blocks')
    if blocks is None or len(blocks) < nblocks or blocks[0].shape()[0] < m or blocks[0].dimension() != b.dimension() \
            or blocks[0].data_type() != dt:
        blocks = work['blocks'] = [b.new_vectors(m) for _ in range(nblocks)]
    for v in blocks:
        v.select(m)
    if precond is not None:
        v_prev, v_cur, w, u_cur, z, d_pp, d_prev, d_cur, _ = blocks[:9]
    else:
        v_prev, v_cur, w, d_pp, d_prev, d_cur = blocks[:6]
        u_cur, z = v_cur, w
    tolv = np.broadcast_to(np.asarray(tol, dtype=np.float64), (m,))

    x.zero()
    # ---- first Lanczos block from the right-hand sides
    b.copy(w)
    if precond is not None:
        precond.apply(w, z)
    g = w.dot(z)                                            # Z^H R
    bnorm = np.sqrt(np.abs(np.real(g.diagonal())))
    beta, pinv = _split(g, drop)
    r_cur = beta.shape[0]
    if r_cur == 0:                                          # b = 0
        info.converged = True
        info.residuals = np.zeros((m,))
        return info
    v_cur.select(r_cur)
    w.multiply(pinv.astype(dt), v_cur)
    if precond is not None:
        u_cur.select(r_cur)
        z.multiply(pinv.astype(dt), u_cur)
    phibar = beta.astype(np.complex128 if np.iscomplexobj(beta) else np.float64)
    safe = np.where(bnorm > 0, bnorm, 1.0)
    q_prev = q_pp = None
    beta_cur = None
    r_prev = r_pp = 0
    it = 0
    alphas, betas = [], []
    while True:
        it += 1
        # ---- block Lanczos step
        w.select(r_cur)
        op.apply(u_cur, w)
        info.columns_applied += r_cur
        alpha = w.dot(u_cur)                                # U^H K U
        alpha = 0.5 * (alpha + alpha.conj().T)
        w.add(v_cur, -1.0, alpha.astype(dt))
        if it > 1:
            v_prev.select(r_prev)
            w.add(v_prev, -1.0, np.ascontiguousarray(beta_cur.conj().T).astype(dt))
        if precond is not None:
            z.select(r_cur)
            precond.apply(w, z)
        g = w.dot(z)
        beta_next, pinv = _split(g, drop)
        r_next = beta_next.shape[0]
        if count_negative:
            alphas.append(alpha)
            betas.append(beta_next)
        # ---- QR of the new block column of the tridiagonal matrix: rows j-2 | j-1 | j | j+1
        rho3 = None
        if it > 2:
            tmp = q_pp.conj().T @ np.vstack((np.zeros((r_pp, r_cur), dtype=alpha.dtype), beta_cur.conj().T))
            rho3, t = tmp[:r_pp], tmp[r_pp:]
        elif it == 2:
            t = beta_cur.conj().T
        rho2 = None
        if it > 1:
            tmp = q_prev.conj().T @ np.vstack((t, alpha))
            rho2, t2 = tmp[:r_prev], tmp[r_prev:]
        else:
            t2 = alpha
        qj, rj = sla.qr(np.vstack((t2, beta_next)), mode='full')
        rho1 = rj[:r_cur]
        tmp = qj.conj().T @ np.vstack((phibar, np.zeros((r_next, m), dtype=phibar.dtype)))
        phi, phibar = tmp[:r_cur], tmp[r_cur:]
        # ---- direction block and solution update
        dg = np.abs(rho1.diagonal())
        if dg.size and dg.min() <= 1e-14 * dg.max():
            rho1_inv = np.linalg.pinv(rho1)                 # K singular on the Krylov space
        else:
            rho1_inv = sla.solve_triangular(rho1, np.eye(r_cur, dtype=rho1.dtype))
        d_cur.select(r_cur)
        u_cur.multiply(np.ascontiguousarray(rho1_inv).astype(dt), d_cur)
        if rho2 is not None:
            d_prev.select(r_prev)
            d_cur.add(d_prev, -1.0, np.ascontiguousarray(rho2 @ rho1_inv).astype(dt))
        if rho3 is not None:
            d_pp.select(r_pp)
            d_cur.add(d_pp, -1.0, np.ascontiguousarray(rho3 @ rho1_inv).astype(dt))
        x.add(d_cur, 1.0, np.ascontiguousarray(phi).astype(dt))
        res = np.sqrt(np.sum(np.abs(phibar) ** 2, axis=0)) / safe if r_next > 0 else np.zeros((m,))
        info.iterations = it
        info.residuals = res
        if np.all(res <= tolv) or r_next == 0:
            info.converged = True
            break
        if it >= max_iter:
            break
        # ---- next Lanczos block
        v_prev, v_cur = v_cur, v_prev                       # old V_{j-1} storage receives V_{j+1}
        v_cur.select(r_next)
        w.multiply(pinv.astype(dt), v_cur)
        if precond is not None:
            u_cur.select(r_next)
            z.multiply(pinv.astype(dt), u_cur)
        else:
            u_cur = v_cur
        d_pp, d_prev, d_cur = d_prev, d_cur, d_pp
        q_pp, q_prev = q_prev, qj
        beta_cur = beta_next
        r_pp, r_prev, r_cur = r_prev, r_cur, r_next
    for v in blocks:
        v.select(m)
    if count_negative:
        info._count_negative(alphas, betas)
    return info


def spectrum_upper_bound(op, make_vectors, n, dtype, steps=12):
    """An upper bound of the spectrum of a Hermitian operator from `steps` Lanczos steps on one random vector:
    theta_max + |beta_k s_k| (largest Ritz value plus the residual norm of its Ritz pair), the bound Chebyshev-filtered
    eigensolvers use for the same purpose (Zhou & Saad); 5 % are added on top.  For operators that come without a matrix
    (row-sharded ones), where no Gershgorin bound can be taken."""
    v_prev, v, w = make_vectors(n, 1, dtype), make_vectors(n, 1, dtype), make_vectors(n, 1, dtype)
    v.fill_random()
    nrm = np.sqrt(abs(v.dots(v)[0]))
    v.scale(np.array([nrm]))
    alphas, betas = [], []
    beta = 0.0
    for j in range(steps):
        op.apply(v, w)
        if j > 0:
            w.add(v_prev, -beta)
        a = float(np.real(w.dots(v)[0]))
        w.add(v, -a)
        b = float(np.sqrt(abs(w.dots(w)[0])))
        alphas.append(a)
        betas.append(b)
        if b <= 1e-14 * max(abs(a), 1.0):
            break
        v_prev, v, w = v, w, v_prev
        v.scale(np.array([b]))
        beta = b
    k = len(alphas)
    t = np.diag(alphas) + np.diag(betas[:k - 1], 1) + np.diag(betas[:k - 1], -1)
    lam, vec = sla.eigh(t)
    return 1.05 * (float(lam[-1]) + abs(betas[k - 1] * vec[-1, -1]))


class IterativeSymmetricSolver:
    """(A - sigma B)^-1 by preconditioned block MINRES: the surface of the reference's SparseSymmetricSolver
    (raleigh/algebra/sparse_mkl.py:51-119: analyse / factorize / solve / inertia / size / data_type / sigma)
    without a factorisation.

    a (analyse): a SciPy sparse matrix (its upper triangle defines the operator, as everywhere) or a ready
    device operator with apply(x, y), size() and data_type() -- e.g. a ShardedSparseMatrix, with row-sharded
    blocks.  preconditioner: None, an operator with apply(x, y) (Hermitian positive definite, FIXED and
    linear), or 'chebyshev' (needs pos_def=True: a polynomial p(A) ~ A^-1 of the given degree on
    [hi / ratio, hi], hi the Gershgorin bound of the matrix, a Lanczos bound for a ready operator, or `hi=`): the right choice for a shift in the lower part of the
    spectrum of a positive definite A, where A - sigma I has few negative eigenvalues and p(A)(A - sigma I)
    is a cluster at 1 plus the few hundred eigenvalues below hi / ratio, which the block Krylov space absorbs.
    tol: bound on the relative residual (in the preconditioner's norm) of every column; None: 1e-10, or, inside
    partial_hevp, a hundredth of the eigenvector tolerance asked for there (within [1e-12, 1e-6]) -- the images
    A X of the iterates are carried by recurrence, so what the solves leave behind is never corrected later and
    bounds the accuracy of the eigenvectors: 1e-4 stalls a 1e-6 eigensolve, 1e-6 costs it nothing."""

    def __init__(self, dtype=np.float64, pos_def=False, tol=None, max_iter=1000, preconditioner='auto', degree=16,
                 ratio=250.0, hi=None):
        self._dtype = np.dtype(dtype).type
        self._pos_def = bool(pos_def)
        self.tol = None if tol is None else float(tol)
        self.max_iter = int(max_iter)
        self._pre_spec = ('chebyshev' if pos_def else None) if isinstance(preconditioner, str) and preconditioner == 'auto' \
            else preconditioner
        self._degree, self._ratio, self._hi = int(degree), float(ratio), hi
        self._op = self._opa = self._opb = self._pre = None
        self._vectors = None
        self._work = {}
        self._neg = None
        self.solves = 0
        self.iterations = 0             # block MINRES steps over all solves
        self.columns_applied = 0        # operator applications, in vectors
        self.last = None

    def analyse(self, a, sigma=0, b=None, vectors=None):
        """vectors: factory ``f(n, nvec, data_type=)`` of the blocks the solver creates itself (the spectrum bound of a ready
        operator, the inertia probe) -- row-sharded ones for a row-sharded operator."""
        from .sparse import SparseSymmetricMatrix
        self._vectors = vectors
        self._matrix = None
        if hasattr(a, 'apply'):
            self._opa = a
        else:
            self._matrix = a
            self._opa = SparseSymmetricMatrix(a)
        if np.dtype(self._opa.data_type()).type != self._dtype:
            raise ValueError('the solver and the matrix data types differ')
        if b is not None and not hasattr(b, 'apply'):
            b = SparseSymmetricMatrix(b)
        self._opb = b
        self._n = self._opa.size()
        self._sigma = sigma
        self._op = ShiftedOperator(self._opa, sigma, b)

    def factorize(self):
        """Sets up the preconditioner (the counterpart of the reference's numerical factorisation)."""
        spec = self._pre_spec
        if spec is None or hasattr(spec, 'apply'):
            self._pre = spec
            return
        if spec != 'chebyshev':
            raise ValueError('unknown preconditioner %s' % repr(spec))
        if not self._pos_def:
            raise ValueError("the 'chebyshev' preconditioner needs a positive definite matrix (pos_def=True)")
        import scipy.sparse as scs
        from .precond import ChebyshevPreconditioner, gershgorin_upper_bound
        hi = self._hi
        if hi is None:
            if self._matrix is None:              # a ready operator (e.g. row-sharded): a Lanczos bound from its own products
                from .vectors import Vectors
                mk = self._vectors if self._vectors is not None else (lambda n, nv, data_type: Vectors(n, nv, data_type=data_type))
                hi = spectrum_upper_bound(self._opa, mk, self._n, self._dtype)
            else:
                u = scs.triu(scs.csr_matrix(self._matrix), format='csr')
                hi = gershgorin_upper_bound(u + scs.triu(u, 1).conj().T)
        self.hi = hi
        self._pre = ChebyshevPreconditioner(self._opa, hi, ratio=self._ratio, degree=self._degree)

    def _tol(self):
        return 1e-10 if self.tol is None else self.tol

    def solve(self, b, x, tol=None):
        info = block_minres(self._op, b, x, precond=self._pre, tol=self._tol() if tol is None else tol,
                            max_iter=self.max_iter, work=self._work)
        self.solves += 1
        self.iterations += info.iterations
        self.columns_applied += info.columns_applied
        self.last = info
        if not info.converged:
            raise RuntimeError('block MINRES did not reach %.1e in %d steps (worst column %.1e): move the shift or '
                               'strengthen the preconditioner' % (self._tol(), info.iterations, float(np.max(info.residuals))))

    def apply(self, b, x):
        self.solve(b, x)

    def signs(self, vectors=None):
        """(has a negative eigenvalue, has a positive eigenvalue) of A - sigma B from a SHORT Lanczos run (a block of eight
        random vectors, residual 1e-2): by interlacing a negative (positive) Ritz value proves a negative (positive)
        eigenvalue, and the extreme ones show within a few steps -- all partial_hevp needs to know for an integer `which`
        (one-sided or two-sided search); a tenth of the cost of the full count."""
        if self._neg is not None:
            return self._neg > 0, self._neg < self._n
        from .vectors import Vectors
        make = vectors or self._vectors or (lambda n, nv, data_type: Vectors(n, nv, data_type=data_type))
        k = min(8, self._n)
        b, x = make(self._n, k, data_type=self._dtype), make(self._n, k, data_type=self._dtype)
        b.fill_random()
        info = block_minres(self._op, b, x, precond=self._pre, tol=1e-2, max_iter=self.max_iter, count_negative=True)
        return info.negative > 0, info.negative < info.ritz_values

    def inertia(self, probe=None, vectors=None):
        """(negative, positive) eigenvalue counts of A - sigma B from a Lanczos count: a probe solve with a block
        of random right-hand sides, the negative eigenvalues of its projected operator counted
        (MinresInfo._count_negative: a lower bound that is sharp once the Krylov space holds the negative
        invariant subspace -- the probe runs to the solver's tolerance, far beyond that point for a shift in the
        lower part of the spectrum).  The reference reads the inertia off PARDISO's factors."""
        if self._neg is None:
            from .vectors import Vectors
            k = 32 if probe is None else int(probe)
            make = vectors or self._vectors or (lambda n, nv, data_type: Vectors(n, nv, data_type=data_type))
            while True:
                k = min(k, self._n)
                b, x = make(self._n, k, data_type=self._dtype), make(self._n, k, data_type=self._dtype)
                b.fill_random()
                info = block_minres(self._op, b, x, precond=self._pre, tol=self._tol(), max_iter=self.max_iter,
                                    count_negative=True)
                if not info.converged:
                    raise RuntimeError('block MINRES probe did not converge: inertia unavailable')
                if info.negative < info.probe_width or k >= self._n:
                    break
                k *= 2                      # as many negative eigenvalues as start vectors: the count is saturated
            self._neg = info.negative
        return self._neg, int(self._n - self._neg)

    def size(self):
        return self._n

    def data_type(self):
        return self._dtype

    def sigma(self):
        return self._sigma

    def operator(self):
        """The unshifted device operator A (for the Rayleigh-Ritz refinement of the converged pairs)."""
        return self._opa
