"""Block Jacobi-conjugated-gradients eigensolver driving the abstract Vectors.

This repository's own driver, written from the algorithm (not the text) of
raleigh/core/solver.py so that the hot path can be run end to end on machines
where the reference is not installed.  It keeps the reference's public surface
-- ``Options``, ``Problem``, ``Solver.solve / convergence_data /
set_preconditioner``, the result attributes and the status codes
(raleigh/core/solver.py:141-197, 224-301, 333-428) -- and its numerical method:

* Rayleigh-Ritz in span{X, Y}, X the current block of Ritz vectors, Y the
  preconditioned residuals conjugated against the previous search directions Z
  (solver.py:1315-1351) and B-orthogonalised against X and the locked vectors;
* linearly dependent directions removed by a pivoted Cholesky factorisation of
  the Gram matrix of (X, Y) (solver.py:1401-1435);
* eigenvector error estimates from the history of Ritz-value decrements
  ("kinematic") and from residuals + spectral gaps (solver.py:976-1048);
* converged pairs are locked (deflation) from the margins inwards and the block
  is refilled from the inner Ritz vectors (solver.py:1127-1313, 1491-1535).

It touches vectors ONLY through the Vectors methods and ``op.apply(x, y)``; all
m x m work is NumPy/SciPy on the host.  Differences from the reference, none of
which change converged results: the block is kept compacted (active vectors
first), the history bookkeeping is array-based, the pivoted Cholesky is
vectorised, and predicted Ritz-value decrements use |q|^2 for complex data.
"""

import math

import numpy as np
import scipy.linalg as sla

RECORDS = 100


class DefaultConvergenceCriteria:
    '''Convergence criteria used when Options.convergence_criteria is None.'''

    def __init__(self):
        self.tolerance = 1e-3
        self.error = 'kinematic eigenvector error'

    def set_error_tolerance(self, error, tolerance):
        self.error = error
        self.tolerance = tolerance

    def satisfied(self, solver, i):
        err = solver.convergence_data(self.error, i)
        return err >= 0 and err <= self.tolerance


class Options:
    '''Solver options (same attributes as raleigh/core/solver.py:141-197).'''

    def __init__(self):
        self.verbosity = 0
        self.max_iter = -1
        self.min_iter = 0
        self.block_size = -1
        self.threads = -1
        self.sigma = None
        self.convergence_criteria = None
        self.stopping_criteria = None
        self.detect_stagnation = True
        self.max_quota = 0.75


class EstimatedErrors:
    '''Kinematic / residual-based error estimates of the locked eigenpairs.'''

    def __init__(self):
        self.kinematic = np.zeros((0,), dtype=np.float32)
        self.residual = np.zeros((0,), dtype=np.float32)

    def __getitem__(self, item):
        return self.kinematic[item], self.residual[item]

    def append(self, est):
        self.kinematic = np.concatenate((self.kinematic, est[0, :]))
        self.residual = np.concatenate((self.residual, est[1, :]))

    def reorder(self, ind):
        self.kinematic = self.kinematic[ind]
        self.residual = self.residual[ind]


class Problem:
    '''Eigenvalue problem: 'std' A x = lmd x; 'gen' A x = lmd B x; 'pro' A B x = lmd x.'''

    def __init__(self, v, A, B=None, prod=None):
        self.__vector = v
        self.__A = A
        self.__B = B
        self.__type = 'std' if B is None else ('gen' if prod is None else 'pro')

    def A(self):
        return self.__A

    def B(self):
        return self.__B

    def type(self):
        return self.__type[0]

    def vector(self):
        return self.__vector


class _History:
    """Per-iterate convergence history, indexed by position in the active block."""

    def __init__(self, m):
        self.iterations = np.zeros((m,), dtype=np.int32)
        self.dlmd = np.zeros((m, RECORDS), dtype=np.float32)
        self.dX = np.ones((m,), dtype=np.float32)
        self.acf = np.ones((2, m), dtype=np.float32)

    def remap(self, src, solver):
        """src[i] = old position of new iterate i, or -1 for a fresh iterate."""
        src = np.asarray(src, dtype=np.int64)
        keep = src >= 0
        old = src[keep]

        def take(a, fresh, axis=0):
            shape = list(a.shape)
            shape[axis] = len(src)
            b = np.full(shape, fresh, dtype=a.dtype)
            if axis == 0:
                b[keep] = a[old]
            else:
                b[:, keep] = a[:, old]
            return b
        self.iterations = take(self.iterations, 0)
        self.dlmd = take(self.dlmd, 0)
        self.dX = take(self.dX, 1)
        self.acf = take(self.acf, 1, axis=1)
        solver.cnv = take(solver.cnv, 0)
        solver.lmd = take(solver.lmd, 0)
        solver.res = take(solver.res, -1)
        solver.err_lmd = take(solver.err_lmd, -1, axis=1)
        solver.err_X = take(solver.err_X, -1, axis=1)


class Solver:
    '''Block JCG solver; attributes as documented in raleigh/core/solver.py:254-301.'''

    def __init__(self, problem):
        self.__problem = problem
        self.__P = None
        self.iteration = 0
        self.lcon = 0
        self.rcon = 0
        self.eigenvalues = np.zeros((0,), dtype=np.float64)
        self.eigenvalue_errors = EstimatedErrors()
        self.eigenvector_errors = EstimatedErrors()
        self.residual_norms = np.zeros((0,), dtype=np.float32)
        self.convergence_status = np.zeros((0,), dtype=np.int32)
        self.eigenvectors = None
        self.eigenvectors_im = None
        self.block_size = None
        self.cnv = None
        self.lmd = None
        self.res = None
        self.err_lmd = None
        self.err_X = None

    def set_preconditioner(self, P):
        self.__P = P

    def problem(self):
        return self.__problem

    def preconditioner(self):
        return self.__P

    def convergence_data(self, what='residual', which=0):
        '''Current convergence data; `what` may be abbreviated (solver.py:333-387).'''
        if 'block' in what:
            return self.block_size
        if 'res' in what and 'vec' not in what:
            return self.res[which] / self._max_abs_eigenvalue()
        if 'val' in what:
            if 'max' in what:
                return self._max_abs_eigenvalue()
            if 'err' in what:
                return self.err_lmd[0, which] if 'k' in what else self.err_lmd[1, which]
            return self.lmd[which]
        if 'vec' in what:
            return self.err_X[0, which] if 'k' in what else self.err_X[1, which]
        raise ValueError('convergence data %s not found' % what)

    def _max_abs_eigenvalue(self):
        mx = np.amax(np.abs(self.lmd)) if len(self.lmd) else 0.0
        if self.lcon + self.rcon > 0:
            mx = max(mx, np.amax(np.abs(self.eigenvalues)))
        return mx

    # ------------------------------------------------------------------ solve
    def solve(self, eigenvectors, options=Options(), which=(-1, -1), extra=(-1, -1), init=(None, None)):
        '''Computes eigenpairs; arguments and return status as raleigh/core/solver.py:389-428.'''
        verb = options.verbosity
        largest = not _is_pair(which)
        if largest:
            left = which // 2 if which >= 0 else -1
            right = which - left if which >= 0 else -1
        else:
            left, right = int(which[0]), int(which[1])
        if left == 0 and right == 0:
            if verb > -1:
                print('No eigenpairs requested, quit')
            return 0
        m = int(options.block_size)
        if m < 0:
            m = _default_block_size(left, right, extra, init, options.threads)
        else:
            least = 3 if ((left == 0 or right == 0) and not largest) else 4
            if m < least:
                if verb > -1:
                    print('Block size %d too small, will use %d instead' % (m, least))
                m = least
        self.block_size = m
        n = eigenvectors.dimension()
        self.iteration = 0
        self.lcon = 0
        self.rcon = 0
        self.eigenvalues = np.zeros((0,), dtype=np.float64)
        self.eigenvalue_errors = EstimatedErrors()
        self.eigenvector_errors = EstimatedErrors()
        self.residual_norms = np.zeros((0,), dtype=np.float32)
        self.convergence_status = np.zeros((0,), dtype=np.int32)

        if m < n // 2:
            try:
                # the m x m LAPACK/BLAS work is latency-bound: one thread beats the pool by 10-40x
                with _single_threaded_blas():
                    status = self._iterate(eigenvectors, options, which, extra, init)
            except _Error as err:
                if verb > -1:
                    print('%s' % err.value)
                return -1
            if status > 1:
                if verb > -1:
                    print('core solver return status %d' % status)
                return status - 1
        else:
            status = 1
        if status == 0:
            return 0
        self._complement_rayleigh_ritz(eigenvectors, verb)
        return 0

    def _complement_rayleigh_ritz(self, eigenvectors, verb):
        """Dense Rayleigh-Ritz on a random basis of the complement of the locked vectors
        (small problems / leftovers; solver.py:502-585)."""
        problem = self.__problem
        std, pro = problem.type() == 's', problem.type() == 'p'
        n = eigenvectors.dimension()
        Xc = eigenvectors
        nc = Xc.nvec()
        m = n - nc
        if m < 1:
            return
        if verb > -1:
            print('%d eigenpairs not computed by CG, applying Rayleigh-Ritz procedure' % m)
            print('in the complement subspace...')
        A, B = problem.A(), problem.B()
        X = eigenvectors.new_vectors(m)
        X.fill_random()
        Y = X.new_vectors(m)
        Z = X.new_vectors(m)
        dt = eigenvectors.data_type()
        if nc > 0:
            BXc = Xc
            if not std:
                BXc = eigenvectors.clone()
                B.apply(Xc, BXc)
            Gci = 2 * np.identity(nc, dtype=dt) - BXc.dot(Xc)
            for _ in range(2):
                X.add(Xc, -1.0, np.dot(Gci, X.dot(BXc)))

        def gram_b():
            if std:
                return X.dot(X)
            B.apply(X, Y)
            return Y.dot(X)
        # Orthonormalise the random basis before the Rayleigh-Ritz step: X <- X Q diag(lmd)^-1/2 from the
        # eigenpairs of its Gram matrix, twice (the second pass removes what the conditioning of the first left
        # behind).  Single-precision blocks of a standard problem are taken to double precision on the device
        # for this: the Gram matrix of n - nc random vectors in the (n - nc)-dimensional complement has a
        # condition of 1e5 .. 1e7, beyond 1 / (100 eps) in fp32, and the reference (solver.py:541-566), which
        # then drops the "dependent" directions and hands eigh(XAX, XBX) what is left, loses 1.6 % on
        # sigma_max of a 200 x 400 fp32 PCA batch that way: the directions it drops are random ones, not
        # null vectors of the operator.
        single = dt in (np.float32, np.complex64)
        wide = {np.float32: np.float64, np.complex64: np.complex128}.get(dt, dt)
        W = X
        if single and std:
            W = X.new_vectors(m, data_type=wide)
            X.convert_to(W)
        Zw = W.new_vectors(m) if W is not X else Z
        for sweep in range(2):
            if W is X:
                XBX = gram_b()
            else:
                XBX = W.dot(W)
            lmd, Q = sla.eigh(-XBX)
            lmd = -lmd
            k = int(np.sum(lmd <= 100 * np.finfo(wide if W is not X else dt).eps * lmd[0])) if sweep == 0 else 0
            if k > 0 and verb > -1:
                print('dropping %d linear dependent vectors from the Rayleigh-Ritz procedure...' % k)
            m -= k
            T = (Q[:, :m] / np.sqrt(np.abs(lmd[:m]))[None, :]).astype(XBX.dtype)
            Zw.select(m)
            W.multiply(np.ascontiguousarray(T), Zw)
            for V in (X, Y, Z, W, Zw):
                V.select(m)
            Zw.copy(W)
        if W is not X:
            W.convert_to(X)
        XBX = gram_b()
        if pro:
            A.apply(Y, Z)
            XAX = Z.dot(Y)
        else:
            A.apply(X, Z)
            XAX = Z.dot(X)
        lmdx, Q = sla.eigh(XAX, XBX)
        X.multiply(Q, Z)
        Z.copy(X)
        eigenvectors.append(X)
        self.eigenvalues = np.concatenate((self.eigenvalues, lmdx))

    # ------------------------------------------------------------------ the CG iteration
    def _iterate(self, eigenvectors, options, which, extra, init):
        verb = options.verbosity
        sigma = options.sigma
        largest = not _is_pair(which)
        left, right = (which, which) if largest else (int(which[0]), int(which[1]))

        m = self.block_size
        if left == 0 and not largest:
            left_ratio, lbs = 0.0, 1
        elif right == 0:
            left_ratio, lbs = 1.0, m - 1
        elif left > 0 and right > 0:
            left_ratio = left / (left + 1.0 * right)
            lbs = min(max(int(round(left_ratio * m)), 2), m - 2)
        else:
            left_ratio, lbs = 0.5, m // 2
        extra_left, extra_right = int(extra[0]), int(extra[1])
        left_total = right_total = None
        if left >= 0:
            left_total = left + extra_left if extra_left > 0 else max(left + 1, lbs)
        if right >= 0:
            right_total = right + extra_right if extra_right > 0 else max(right + 1, m - lbs)
        if verb > 0:
            print('left block size %d, right block size %d' % (lbs, m - lbs))

        problem = self.__problem
        vector = problem.vector()
        ptype = problem.type()
        std, gen, pro = ptype == 's', ptype == 'g', ptype == 'p'
        dt = vector.data_type()
        eps = np.finfo(dt).eps
        single = dt in (np.float32, np.complex64)

        self.cnv = np.zeros((m,), dtype=np.int32)
        self.lmd = np.zeros((m,), dtype=np.float64)
        self.res = -np.ones((m,), dtype=np.float32)
        self.err_lmd = -np.ones((2, m), dtype=np.float32)
        self.err_X = -np.ones((2, m), dtype=np.float32)
        criteria = options.convergence_criteria or DefaultConvergenceCriteria()
        hist = _History(m)

        opA, opB, opP = problem.A(), problem.B(), self.__P
        # one-pass forms of multiply+add / copy+add when the Vectors type offers them
        fused = hasattr(vector, 'combine') and hasattr(vector, 'lincomb')
        # several Gram / dots reductions with ONE host synchronisation (and one all-reduce when the
        # rows are sharded): the reference issues them as back-to-back blocking calls on shared
        # operands (solver.py:854-861, 1321-1339, 1376-1381, 1444-1447)
        batched = fused and hasattr(vector, 'reduction_batch') and not pro

        # ---- work blocks: the active vectors always occupy the first nx slots
        X = vector.new_vectors(m)
        X.fill_random()
        Y, Z, W = vector.new_vectors(m), vector.new_vectors(m), vector.new_vectors(m)
        AX, AY, AZ = vector.new_vectors(m), vector.new_vectors(m), vector.new_vectors(m)
        if std:
            BX, BY, BZ = X, Y, Z
        else:
            BX, BY, BZ = vector.new_vectors(m), vector.new_vectors(m), vector.new_vectors(m)

        pos = 0
        for guess, cap in ((init[0], lbs), (init[1], m - lbs)):
            if guess is not None:
                k = min(cap, guess.nvec())
                X.select(k, pos)
                guess.select(k)
                guess.copy(X)
                pos += k
        X.select(m)
        s = X.dots(X)
        for i in np.nonzero(s == 0)[0]:
            if verb > -1:
                print('Zero initial guess, replacing with random')
            X.select(1, int(i))
            X.fill_random()
        X.select(m)
        X.scale(np.sqrt(np.abs(X.dots(X))))

        # ---- locked vectors (constraints) and the approximate inverse of their Gram matrix
        self.eigenvectors = eigenvectors
        Xc = eigenvectors
        if std:
            BXc = Xc
        else:
            BXc = eigenvectors.clone()
            if Xc.nvec() > 0:
                opB.apply(Xc, BXc)
            self.eigenvectors_im = BXc
        Gc = Gci = None
        if Xc.nvec() > 0:
            Gc = BXc.dot(Xc)
            Gci = 2 * np.identity(Xc.nvec(), dtype=dt) - Gc

        def project_out_locked(V, against, minus):
            """V -= minus * Gci * <V, against>  (coefficients (nc x nv))."""
            V.add(minus, -1.0, np.dot(Gci, V.dot(against)))

        if Xc.nvec() > 0:
            project_out_locked(X, BXc, Xc)
        if not std:
            opB.apply(X, BX)
        XBX = BX.dot(X)

        # drop linearly dependent start vectors, refill with random ones
        ind, dropped = _pivoted_cholesky(XBX.copy(), 0, 1e-2)[1:]
        if dropped > 0:
            if verb > 0:
                print('dropped %d initial vectors out of %d' % (dropped, m))
            keep = m - dropped
            if keep > 0:
                W.select(keep)
                X.copy(W, ind[:keep])
                X.select(keep)
                W.copy(X)
            X.select(dropped, keep)
            X.fill_random()
            if Xc.nvec() > 0:
                project_out_locked(X, BXc, Xc)
            X.select(m)
            W.select(m)
            if not std:
                opB.apply(X, BX)
            XBX = BX.dot(X)

        # ---- Rayleigh-Ritz in the initial subspace
        if pro:
            opA.apply(BX, AX)
            XAX = AX.dot(BX)
        else:
            opA.apply(X, AX)
            XAX = AX.dot(X)
        lmdx, Q = sla.eigh(XAX, XBX)
        for V in ((X, AX) if std else (X, AX, BX)):
            V.multiply(Q, W)
            W.copy(V)

        nx, leftX = m, lbs
        rightX = nx - leftX
        cap_left = lbs                 # capacity of the left part of the block
        nz, lmdz, rec = 0, None, 0
        dlmd_min_left = dlmd_min_right = 0.0
        max_iter = options.max_iter if options.max_iter >= 0 else 100
        min_iter = options.min_iter
        detect_stagn = options.detect_stagnation
        self.iteration = 0
        lmd, res, err_lmd, err_X = self.lmd, self.res, self.err_lmd, self.err_X

        def select_all(k, *blocks):
            for V in blocks:
                V.select(k)

        while True:
            lmd, res, err_lmd, err_X = self.lmd, self.res, self.err_lmd, self.err_X
            it_left = np.amax(hist.iterations[:leftX]) if (left != 0 and leftX > 0) else 0
            it_right = np.amax(hist.iterations[nx - rightX:nx]) if (right != 0 and rightX > 0) else 0
            if max(it_left, it_right) >= max_iter:
                if verb > -1:
                    print('iterations limit of %d exceeded, terminating' % max_iter)
                break
            if verb > 0:
                print('------------- iteration %d' % self.iteration)

            select_all(nx, X, AX, BX)
            speculative = None
            if batched and Xc.nvec() == 0:
                # One round trip for the Ritz-pair check AND the residual norms: the residuals are
                # formed with the Ritz values of the last Rayleigh-Ritz step (which the recomputed
                # Rayleigh quotients reproduce to rounding unless orthogonality was lost -- then the
                # restart below recomputes everything) while the Gram pair is in flight.
                W.select(nx)
                W.lincomb(1.0, AX, -np.asarray(lmdx[:nx]), BX if gen else X)
                rb = X.reduction_batch()
                rb.gram([X], [AX, BX] if not std else [AX, X])
                rb.dots(W, W)
                G2, speculative = rb.run()
                XAX, XBX = G2[:nx], G2[nx:]
            else:
                XAX = AX.dot(BX) if pro else AX.dot(X)
                XBX = BX.dot(X)
            new_lmd = np.real(XAX.diagonal() / XBX.diagonal())

            # ---- sanity of the Ritz pairs; restart through an SVD-orthonormalisation if lost
            rv_err = np.amax(np.abs(new_lmd - lmdx)) / np.amax(np.abs(lmdx))
            rv_no = np.amax(np.abs(XBX - np.eye(nx)))
            if max(rv_err, rv_no) > math.sqrt(eps):
                if verb > 0:
                    print('Ritz values error: %.1e' % rv_err)
                    print('Ritz vectors non-orthonormality: %.1e' % rv_no)
                    print('restarting...')
                rec, nz = 0, 0
                speculative = None
                X.svd()
                if not std:
                    opB.apply(X, BX)
                XBX = BX.dot(X)
                if pro:
                    opA.apply(BX, AX)
                    XAX = AX.dot(BX)
                else:
                    opA.apply(X, AX)
                    XAX = AX.dot(X)
                lmdx, Q = sla.eigh(XAX, XBX)
                W.select(nx)
                for V in ((X, AX) if std else (X, AX, BX)):
                    V.multiply(Q, W)
                    W.copy(V)
                XAX = AX.dot(BX) if pro else AX.dot(X)
                XBX = BX.dot(X)
                new_lmd = np.real(XAX.diagonal() / XBX.diagonal())

            hist.iterations[:nx] += 1
            if rec > 0:
                delta = lmd[:nx] - new_lmd
                thresh = math.sqrt(eps) * np.maximum(np.abs(lmd[:nx]), np.abs(new_lmd))
                big = np.abs(delta) > thresh
                hist.dlmd[:nx, rec - 1][big] = delta[big]
            lmd[:nx] = new_lmd

            # ---- residuals W = A X - (B) X lmd, orthogonalised against the locked vectors
            W.select(nx)
            Y.select(nx)
            if speculative is not None and np.all(np.abs(new_lmd - lmdx[:nx]) <= np.maximum(
                    1e-3 * np.sqrt(np.abs(speculative)), 1e-14 * np.amax(np.abs(lmdx[:nx])))):
                # W and its norms are already there: the Ritz values it was formed with differ from the
                # recomputed ones by less than 0.1 % of the residual norms
                s = speculative
            else:
                if fused:           # one pass: W = AX - (B)X diag(lmd)
                    W.lincomb(1.0, AX, -lmd[:nx], BX if gen else X)
                else:
                    AX.copy(W)
                    W.add(BX if gen else X, -lmd[:nx])
                if Xc.nvec() > 0:
                    project_out_locked(W, BXc if pro else Xc, BXc if gen else Xc)
                if pro:
                    W.copy(Y)
                    opB.apply(Y, W)
                    s = W.dots(Y)
                else:
                    s = W.dots(W)
            res[:nx] = np.sqrt(np.abs(s))

            self._kinematic_estimates(hist, rec, nx)
            if not gen:
                self._residual_estimates(hist, nx, leftX, rightX)
            if verb > 1:
                self._print_table(hist, nx)

            # ---- stagnation thresholds and clusters of close Ritz values
            eps67 = eps ** 0.67
            dlmd_min_lft = dlmd_min_rgt = 0.0
            if leftX > 0:
                dlmd_min_lft = eps67 * np.amax(np.abs(hist.dlmd[:leftX, rec - 1]))
            if rightX > 0:
                dlmd_min_rgt = eps67 * np.amax(np.abs(hist.dlmd[nx - rightX:nx, rec - 1]))
            if self.iteration == 2:
                dlmd_min_left, dlmd_min_right = dlmd_min_lft, dlmd_min_rgt
            cluster_len = np.zeros((nx,), dtype=np.int32)
            if self.iteration >= 2:
                for i in range(leftX - 1):
                    if abs(lmd[i + 1] - lmd[i]) <= dlmd_min_lft:
                        cluster_len[i] = max(cluster_len[i], 1)
                        cluster_len[i + 1] = cluster_len[i] + 1
                for i in range(nx - 1, nx - rightX, -1):
                    if abs(lmd[i - 1] - lmd[i]) <= dlmd_min_rgt:
                        cluster_len[i] = max(cluster_len[i], 1)
                        cluster_len[i - 1] = cluster_len[i] + 1

            # ---- convergence tests from the margins inwards
            def sweep(positions, want, sign_ok, dmin, inward):
                count = 0
                for k in positions:
                    if want == 0:
                        break
                    if sigma is not None and not sign_ok(lmd[k]):
                        break
                    it = hist.iterations[k]
                    if it < min_iter:
                        break
                    d1 = abs(hist.dlmd[k, max(0, rec - 1)])
                    d2 = abs(hist.dlmd[k, max(0, rec - 3)])
                    if criteria.satisfied(self, k):
                        if verb > 0:
                            print('eigenpair at %e converged after %d iterations, error %.1e / %.1e'
                                  % (lmd[k], it, err_X[0, k], err_X[1, k]))
                        count += 1
                        self.cnv[k] = self.iteration + 1
                    elif detect_stagn and it > 2 and d1 <= dmin and (d1 > d2 or d1 == 0.0):
                        if verb > 0:
                            print('eigenpair at %e stagnated, error %.1e / %.1e'
                                  % (lmd[k], err_X[0, k], err_X[1, k]))
                        count += 1
                        self.cnv[k] = -self.iteration - 1
                    else:
                        # a member of a cluster that keeps converging cancels the stagnation
                        # verdicts of the cluster members already passed
                        for j in range(1, cluster_len[k]):
                            l = k - inward * j
                            if 0 <= l < nx and self.cnv[l] == -self.iteration - 1:
                                self.cnv[l] = 0
                                count -= 1
                                if verb > 0:
                                    print('stagnation of %e cancelled' % lmd[l])
                        break
                return count

            lcon = sweep(range(leftX - leftX // 4), left, lambda v: v <= 0, dlmd_min_left, +1)
            rcon = sweep(range(nx - 1, nx - 1 - (rightX - rightX // 4), -1), right, lambda v: v >= 0,
                         dlmd_min_right, -1)
            if largest:         # lock the largest in modulus first
                while lcon > 0 and abs(lmd[lcon - 1]) < abs(lmd[nx - rcon - 1]):
                    self.cnv[lcon - 1] = 0
                    lcon -= 1
                while rcon > 0 and abs(lmd[lcon]) > abs(lmd[nx - rcon]):
                    self.cnv[nx - rcon] = 0
                    rcon -= 1

            # ---- lock converged pairs: move them to Xc and extend its Gram matrix
            for first, count in ((0, lcon), (nx - rcon, rcon)):
                if count < 1:
                    continue
                sl = slice(first, first + count)
                self.eigenvalues = np.concatenate((self.eigenvalues, lmd[sl]))
                self.eigenvalue_errors.append(err_lmd[:, sl])
                self.eigenvector_errors.append(err_X[:, sl])
                self.residual_norms = np.concatenate((self.residual_norms, res[sl]))
                self.convergence_status = np.concatenate((self.convergence_status, self.cnv[sl]))
                ncon = Xc.nvec()
                X.select(count, first)
                Gu = X.dot(BXc) if ncon > 0 else None
                Xc.append(X)
                if not std:
                    BX.select(count, first)
                    BXc.append(BX)
                if ncon < 1:
                    Gc = BXc.dot(Xc)
                else:
                    Gl = BXc.dot(X)
                    Gc = np.concatenate((np.concatenate((Gc, Gu), axis=1), Gl))
            if Xc.nvec() > 0 and (lcon + rcon > 0 or Gci is None):
                Gci = 2 * np.identity(Xc.nvec(), dtype=dt) - Gc
            self.lcon += lcon
            self.rcon += rcon

            # ---- termination
            if options.stopping_criteria is not None and options.stopping_criteria.satisfied(self):
                return 0
            if largest and right > 0 and self.lcon + self.rcon >= right:
                return 0
            left_done = left >= 0 and self.lcon >= left
            right_done = right >= 0 and self.rcon >= right
            if left_done and right_done:
                return 0
            if sigma is not None:
                if right_done and lcon < nx:
                    li, ei = lmd[lcon], err_lmd[0, lcon]
                    if li > 0 and ei != -1.0 and ei < li / 4:
                        return 4
                if left_done and nx - rcon - 1 >= 0:
                    li, ei = lmd[nx - rcon - 1], err_lmd[0, nx - rcon - 1]
                    if li < 0 and ei != -1.0 and ei < -li / 4:
                        return 5
            if eigenvectors.nvec() > options.max_quota * eigenvectors.dimension():
                return 1

            # ---- search directions: preconditioned residuals of ALL ny = nx iterates
            ny, lmd_y = nx, lmd[:nx].copy()
            select_all(ny, W, Y)
            if not pro:
                if opP is None:
                    Y, W = W, Y         # the residuals ARE the search directions: swap, no copy
                    if std:
                        BY = Y
                else:
                    opP.apply(W, Y)
            # the active block shrinks by the locked pairs
            first, nx_act = lcon, nx - lcon - rcon
            leftX -= lcon
            rightX -= rcon
            XAX = XAX[first:first + nx_act, first:first + nx_act]
            XBX = XBX[first:first + nx_act, first:first + nx_act]
            for V in ((X, AX) if std else (X, AX, BX)):
                V.select(nx_act, first)

            if nz > 0:          # conjugate Y against the previous search directions Z
                select_all(nz, Z, AZ, BZ)
                if batched:     # ZAY, ZBY and the two sets of norms: one round trip
                    rb = Y.reduction_batch()
                    rb.gram([Y], [AZ, BZ])
                    rb.dots(Y, Y)
                    rb.dots(Z, Z)
                    G2, dy, dz = rb.run()
                    ZAY, ZBY = G2[:nz], G2[nz:]
                    sy, sz = np.sqrt(np.abs(dy)), np.sqrt(np.abs(dz))
                else:
                    ZAY = W.dot(AZ) if pro else Y.dot(AZ)
                    ZBY = Y.dot(BZ)
                    sy = np.sqrt(np.abs(Y.dots(Y)))
                    sz = np.sqrt(np.abs(Z.dots(Z)))
                Num = ZAY - ZBY * lmd_y[None, :]
                Den = np.asarray(lmdz)[:, None] - lmd_y[None, :]
                with np.errstate(divide='ignore', invalid='ignore'):
                    Beta = np.where(np.abs(Num) >= 100 * (sy[None, :] / sz[:, None]) * np.abs(Den), 0.0, Num / Den)
                Beta = np.nan_to_num(Beta).astype(dt)
                Y.add(Z, -1.0, Beta)
                if pro:
                    W.add(BZ, -1.0, Beta)
                    BY.select(ny)
                    W.copy(BY)
            elif pro:
                BY.select(ny)
                W.copy(BY)

            if nx_act > 0:      # B-orthogonalise Y against the active X ...
                Q = Y.dot(BX)
                Y.add(X, -1.0, Q)
                if pro:
                    BY.add(BX, -1.0, Q)
            if Xc.nvec() > 0:   # ... and against the locked vectors
                Q = np.dot(Gci, Y.dot(BXc))
                Y.add(Xc, -1.0, Q)
                if pro:
                    BY.add(BXc, -1.0, Q)

            # ---- B-Gram matrix of (X, Y), Y normalised
            if batched and std:
                # [X | Y]^H Y in one pass and one round trip; the normalisation of Y is applied to the
                # small matrices on the host and to the block on the device
                rb = Y.reduction_batch()
                rb.gram([Y], [X, Y] if nx_act > 0 else [Y])
                G2 = rb.run()[0]
                YBY = G2[nx_act:]
                sn = np.sqrt(np.abs(np.real(YBY.diagonal())))
                Y.scale(sn)
                sn = np.where(sn == 0, 1.0, sn)
                YBY = YBY / sn[:, None] / sn[None, :]
                if nx_act > 0:
                    XBY = G2[:nx_act] / sn[None, :]
                    GB = np.block([[XBX, XBY], [XBY.conj().T, YBY]])
                else:
                    GB = YBY
            else:
                if std:
                    Y.scale(np.sqrt(np.abs(Y.dots(Y))))
                else:
                    BY.select(ny)
                    if not pro:
                        opB.apply(Y, BY)
                    s = np.sqrt(np.abs(BY.dots(Y)))
                    Y.scale(s)
                    BY.scale(s)
                YBY = BY.dot(Y)
                if nx_act > 0:
                    XBY = BY.dot(X)
                    GB = np.block([[XBX, XBY], [XBY.conj().T, YBY]])
                else:
                    GB = YBY

            U, ind, dropped = _pivoted_cholesky(GB, nx_act, 1e-3 if single else 1e-8)
            if dropped > 0 and verb > 0:
                print('dropped %d search directions out of %d' % (dropped, ny))
            ny_old = ny
            ny -= dropped
            if ny < 1:
                if verb > -1:
                    print('no search directions left, terminating')
                return 3
            nxy = nx_act + ny
            U = U[:nxy, :nxy]
            # Pivoting permutes the search directions.  Only when some are DROPPED are the kept ones
            # gathered (into the scratch block, which then changes places with Y); otherwise Y stays
            # where it is and the permutation is applied to the small host matrices instead.
            perm = np.asarray(ind[nx_act:nxy], dtype=np.int64) - nx_act
            if dropped > 0:
                W.select(ny)
                Y.select(ny_old)
                Y.copy(W, perm)
                Y, W = W, Y
                if std:
                    BY = Y
                else:
                    W.select(ny)
                    BY.select(ny_old)
                    BY.copy(W, perm)
                    BY, W = W, BY
                perm = np.arange(ny, dtype=np.int64)
            select_all(ny, Y, AY, BY)

            # ---- A-Gram matrix of (X, Y) and the Rayleigh-Ritz problem in the Cholesky basis
            if pro:
                opA.apply(BY, AY)
                YAY = AY.dot(BY)
                XAY = AY.dot(BX) if nx_act > 0 else None
            else:
                opA.apply(Y, AY)
                if batched and nx_act > 0:      # [X | Y]^H AY: one pass, one round trip
                    rb = AY.reduction_batch()
                    rb.gram([AY], [X, Y])
                    G2 = rb.run()[0]
                    XAY, YAY = G2[:nx_act], G2[nx_act:]
                else:
                    YAY = AY.dot(Y)
                    XAY = AY.dot(X) if nx_act > 0 else None
            YAY = YAY[np.ix_(perm, perm)]                     # to the pivoted order of U
            if nx_act > 0:
                XAY = XAY[:, perm]
            GA = np.block([[XAX, XAY], [XAY.conj().T, YAY]]) if nx_act > 0 else YAY
            Uh = U.conj().T
            G = sla.solve_triangular(Uh, sla.solve_triangular(Uh, GA, lower=True).conj().T, lower=True)
            G = (G + G.conj().T) / 2
            G = G.astype(np.complex128 if G.dtype.kind == 'c' else np.float64)
            lmdxy, Q = sla.eigh(G)

            # ---- how many Ritz vectors continue on each side: slots freed by locking are
            # refilled from the inner Ritz vectors until wanted + spare pairs are covered
            free_left = cap_left - leftX
            free_right = (m - cap_left) - rightX
            if left < 0:
                shift_left = free_left
            elif lcon > 0:
                shift_left = min(max(0, left_total - self.lcon - leftX), free_left)
            else:
                shift_left = 0
            if right < 0:
                shift_right = free_right
            elif rcon > 0:
                shift_right = min(max(0, right_total - self.rcon - rightX), free_right)
            else:
                shift_right = 0
            if shift_left + shift_right > ny:
                shift_left = min(shift_left, int(round(left_ratio * ny)))
                shift_right = min(shift_right, ny - shift_left)
            if left > 0 and lcon > 0 and self.lcon >= left:
                if verb > 0:
                    print('left-hand side converged')
                leftX_new = 0                     # the right part takes over the left capacity
                rightX_new = min(nxy, cap_left + rightX + shift_right)
                cap_left = cap_left + rightX + shift_right - rightX_new
                left_ratio = 0.0
            elif right > 0 and rcon > 0 and self.rcon >= right:
                if verb > 0:
                    print('right-hand side converged')
                still_free = free_left - shift_left
                leftX_new = min(nxy, m - still_free)
                rightX_new = 0
                cap_left = still_free + leftX_new
                left_ratio = 1.0
            else:
                leftX_new = leftX + shift_left
                rightX_new = rightX + shift_right
            nx_new = leftX_new + rightX_new

            # predicted decrements / rotation of the continuing iterates (old iterate i <-> new i
            # counted from the same margin)
            lft0, rgt0 = min(leftX, leftX_new), min(rightX, rightX_new)
            sel_old = np.concatenate((np.arange(lft0), np.arange(nxy - rgt0, nxy))).astype(np.int64)
            QYX = Q[nx_act:, sel_old]
            lmd_sel = lmdxy[sel_old]
            if rec == RECORDS:
                hist.dlmd[:, :-1] = hist.dlmd[:, 1:]
            else:
                rec += 1

            # ---- history bookkeeping for the new block composition
            src = np.full((nx_new,), -1, dtype=np.int64)
            for i in range(min(leftX, leftX_new)):
                src[i] = first + i
            for i in range(min(rightX, rightX_new)):
                src[nx_new - 1 - i] = first + nx_act - 1 - i
            # the Y-components of the new X measure how far the iterates still move
            dX_new = np.ones((nx_new,), dtype=np.float32)
            pred = np.zeros((nx_new,), dtype=np.float32)
            pos_new = np.concatenate((np.arange(lft0), np.arange(nx_new - rgt0, nx_new))).astype(np.int64)
            if len(pos_new) and ny > 0:
                Qy = Q[nx_act:, :]
                # Ritz values of the pencil restricted to span(Y) in the same basis
                Gyy = G[nx_act:, nx_act:]
                ly = np.real(np.einsum('ij,ik,kj->j', Qy[:, sel_old].conj(), Gyy, Qy[:, sel_old]))
                wy = np.sum(np.abs(QYX) ** 2, axis=0)
                dX_new[pos_new] = np.sqrt(wy)
                with np.errstate(divide='ignore', invalid='ignore'):
                    mean_y = np.where(wy > 0, ly / wy, lmd_sel)
                pred[pos_new] = (mean_y - lmd_sel) * wy
            hist.remap(src, self)
            self.lmd[:nx_new] = lmdxy[np.concatenate((np.arange(leftX_new), np.arange(nxy - rightX_new, nxy))).astype(np.int64)]
            hist.dX[:nx_new] = dX_new
            hist.dlmd[:nx_new, rec - 1] = pred

            # ---- new X and the next "previous directions" Z from the Ritz vectors
            Q = sla.solve_triangular(U, Q)
            Qphys = Q.copy()                                  # rows back to the physical order of Y
            Qphys[nx_act + perm] = Q[nx_act:]
            Q = Qphys
            take = np.concatenate((np.arange(leftX_new), np.arange(nxy - rightX_new, nxy))).astype(np.int64)
            rest = np.arange(leftX_new, nxy - rightX_new)
            lmdx = lmdxy[take]
            lmdz = lmdxy[rest]
            nz = len(rest)
            QX, QZ = Q[:, take].astype(dt), Q[:, rest].astype(dt)

            def combine(SX, SY, out, coef):
                """out = SX * coef[:nx_act] + SY * coef[nx_act:]."""
                out.select(coef.shape[1])
                if nx_act > 0 and fused:
                    SX.combine(coef[:nx_act], SY, coef[nx_act:], out)
                elif nx_act > 0:
                    SX.multiply(np.ascontiguousarray(coef[:nx_act]), out)
                    out.add(SY, 1.0, np.ascontiguousarray(coef[nx_act:]))
                else:
                    SY.multiply(np.ascontiguousarray(coef[nx_act:]), out)

            # the new block is written into the scratch block W, which then changes places with
            # the old one (no copy back); the old storage is the scratch of the next triple
            results = {}
            names = ['AX', 'X'] if std else ['AX', 'BX', 'X']
            olds = {'AX': (AX, AY, AZ), 'BX': (BX, BY, BZ), 'X': (X, Y, Z)}
            fused2 = fused and hasattr(X, 'combine2') and nx_act > 0 and nz > 0
            for name in names:
                SX, SY, SZ = olds[name]
                if fused2:      # both results from one pass over SX and SY
                    SZ.select(QZ.shape[1])
                    W.select(QX.shape[1])
                    SX.combine2(QX[:nx_act], QZ[:nx_act], SY, QX[nx_act:], QZ[nx_act:], W, SZ)
                else:
                    if nz > 0:
                        SZ.select(m)
                        combine(SX, SY, SZ, QZ)
                    W.select(m)
                    combine(SX, SY, W, QX)
                results[name] = W
                W = SX
            AX, X = results['AX'], results['X']
            BX = X if std else results['BX']
            select_all(m, Y, AY, BY, W)

            nx, leftX, rightX = nx_new, leftX_new, rightX_new
            self.iteration += 1
        return 2

    # ------------------------------------------------------------------ error estimates
    def _kinematic_estimates(self, hist, rec, nx):
        """Eigenvector/eigenvalue errors from the geometric decay of the Ritz-value
        decrements over the last third of the history (solver.py:976-1008)."""
        if rec <= 3:
            return
        for i in range(nx):
            if hist.dX[i] > 0.01:
                self.err_X[0, i] = -1.0
                continue
            k, s = 0, 0.0
            for r in range(rec - 1, rec - rec // 3 - 2, -1):
                d = abs(hist.dlmd[i, r])
                if d == 0:
                    break
                k += 1
                s += d
            if k < 2 or s == 0:
                continue
            qi = abs(hist.dlmd[i, rec - 1]) / s
            if qi <= 0:
                continue
            qi = qi ** (1.0 / (k - 1))
            hist.acf[1, i] = hist.acf[0, i]
            hist.acf[0, i] = qi
            if qi >= 1.0:
                continue
            self.err_lmd[0, i] = abs(qi / (1 - qi) * hist.dlmd[i, rec - 1])
            qx = math.sqrt(qi)
            self.err_X[0, i] = hist.dX[i] * qx / (1 - qx)

    def _residual_estimates(self, hist, nx, leftX, rightX):
        """Asymptotic Lehmann (eigenvalues) and extended-gap Davis-Kahan (eigenvectors)
        estimates using an inner Ritz value as the pole (solver.py:1010-1048)."""
        lmd, res = self.lmd, self.res
        pole = 0
        for k in range(1, leftX):
            if hist.dX[k] > 0.01:
                break
            if lmd[k] - lmd[k - 1] > res[k]:
                pole = k
        for k in range(pole):
            gap = lmd[pole] - lmd[k]
            self.err_lmd[1, k] = res[k] * res[k] / gap
            self.err_X[1, k] = res[k] / gap
        pole = 0
        for k in range(1, rightX):
            i = nx - k - 1
            if hist.dX[i] > 0.01:
                break
            if lmd[i + 1] - lmd[i] > res[i]:
                pole = k
        for k in range(pole):
            i = nx - k - 1
            gap = lmd[i] - lmd[nx - pole - 1]
            self.err_lmd[1, i] = res[i] * res[i] / gap
            self.err_X[1, i] = res[i] / gap

    def _print_table(self, hist, nx):
        print('  eigenvalue   residual   estimated errors (kinematic/residual)      a.c.f.')
        print('                             eigenvalue            eigenvector ')
        for i in range(nx):
            print('%14e %8.1e  %8.1e / %8.1e    %.1e / %.1e  %.3e  %d'
                  % (self.lmd[i], self.res[i], self.err_lmd[0, i], self.err_lmd[1, i],
                     abs(self.err_X[0, i]), abs(self.err_X[1, i]), hist.acf[0, i], self.cnv[i]))


def _single_threaded_blas():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except Exception:
        import contextlib
        return contextlib.nullcontext()


class _Error(Exception):
    def __init__(self, value):
        self.value = value

    def __str__(self):
        return '??? ' + repr(self.value)


def _is_pair(which):
    try:
        if len(which) != 2:
            raise ValueError('which must be either integer or tuple of 2 integers')
        return True
    except TypeError:
        return False


def _default_block_size(left, right, extra, init, threads):
    """Block size when the user gives none: ~1.2 x the wanted pairs (or wanted + extra),
    rounded up to a multiple of max(threads, 8) (rule of solver.py:1690-1734)."""
    extra_left, extra_right = int(extra[0]), int(extra[1])
    init_left = int(init[0].nvec()) if init[0] is not None else 0
    init_right = int(init[1].nvec()) if init[1] is not None else 0
    unit = max(threads, 8)
    round_up = lambda k: unit * ((k - 1) // unit + 1)
    if left == 0 and right == 0:
        return 0
    if left <= 0 and right <= 0:
        if init_left == 0 and init_right == 0:
            return 2 * unit if (left < 0 and right < 0) else unit
        k = round_up(init_left + init_right)
        return max(k, 2 * unit) if (left < 0 or right < 0) else k

    def total(want, extra_, init_):
        if want <= 0:
            return 0
        return max(want + extra_, init_) if extra_ >= 0 else int(math.floor(max(want, init_) * 1.2))
    left_total, right_total = total(left, extra_left, init_left), total(right, extra_right, init_right)
    if left < 0:
        left_total = right_total
    if right < 0:
        right_total = left_total
    k = round_up(int(left_total + right_total))
    return max(k, 2 * unit) if (left < 0 or right < 0) else k


def _pivoted_cholesky(G, k, eps):
    """G = U^H U with the leading k x k block factorised without pivoting and the remaining
    columns chosen by diagonal pivoting; columns whose pivot, or whose contribution to the
    reciprocal condition number of U^H U, falls below eps are dropped (they come last).
    Returns (U, ind, dropped): U upper triangular in the permuted order `ind`, rows/columns
    of dropped directions zero.  Role of solver.py:1749-1826 (`_piv_chol`), vectorised."""
    G = np.array(G, copy=True)
    n = G.shape[0]
    ind = list(range(n))
    U = np.zeros_like(G)
    if k > 0:
        Uk = sla.cholesky(G[:k, :k])
        U[:k, :k] = Uk
        U[:k, k:] = sla.solve_triangular(Uk.conj().T, G[:k, k:], lower=True)
        S = G[k:, k:] - U[:k, k:].conj().T @ U[:k, k:]
    else:
        S = G
    r = n - k
    kept = 0
    if r > 0:
        # LAPACK ?pstrf: diagonal-pivoted Cholesky S[P, P] = Us^H Us that stops at the first
        # pivot <= eps and reports the numerical rank
        pstrf = sla.get_lapack_funcs('pstrf', (S,))
        c, piv, kept, info = pstrf(np.ascontiguousarray(S), lower=0, tol=eps)
        P = piv.astype(np.int64) - 1
        Us = np.triu(c)
        Us[kept:, :] = 0
        U[:k, k:] = U[:k, k:][:, P]
        U[k:, k:] = Us
        ind[k:] = [ind[k + int(q)] for q in P]
    # condition control: the reciprocal condition number of the kept Gram block must exceed eps.  The condition
    # number of a leading block grows with its size (interlacing), so the largest admissible `kept` is found by
    # bisection: one SVD in the common case, log2(kept) of them instead of one per dropped column otherwise.
    def well_conditioned(q):
        sv = np.linalg.svd(U[:k + q, :k + q], compute_uv=False)
        return (sv[-1] / sv[0]) ** 2 > eps
    if kept > 0 and not well_conditioned(kept):
        lo, hi = 0, kept - 1                    # the answer lies in [lo, hi]; kept = 0 needs no check
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if well_conditioned(mid):
                lo = mid
            else:
                hi = mid - 1
        kept = lo
    U[k + kept:, :] = 0
    U[:, k + kept:] = 0
    return U, ind, r - kept
