"""Partial eigenvalue problem solver for a sparse symmetric/Hermitian matrix on MI355X.

Counterpart of raleigh/interfaces/partial_hevp.py:21-257 (same signature, return
values and status codes) built on this repository's Vectors, sparse operator and
block-JCG driver.  Preconditioned mode (T given) keeps every block on the device;
shift-invert mode (T None) uses the host factorisation of algebra/hip/host_ops.py
for the operator (A - sigma B)^-1 -- the reference uses PARDISO on the host too.
"""

import time

import numpy

from ..algebra.hip import Vectors, SparseSymmetricMatrix
from ..algebra.hip.host_ops import SparseSymmetricSolver
from ..algebra.hip.shift_invert import IterativeSymmetricSolver
from ..core.solver import Problem, Solver, Options, DefaultConvergenceCriteria


def partial_hevp(A, B=None, T=None, buckling=False, sigma=0, which=6, tol=1e-4, verb=0, opt=None,
                 vectors=None, operator=None, solver=None):
    '''Computes several eigenpairs of a sparse real symmetric / Hermitian problem
    (arguments as raleigh/interfaces/partial_hevp.py:23-90).

    Extra keywords (multi-GPU): `vectors` -- a factory ``f(n, data_type=)`` returning an
    empty Vectors (e.g. row-sharded), `operator` -- a ready operator for A with
    ``apply(x, y)``, ``size()`` and ``data_type()`` (e.g. ShardedSparseMatrix); `solver` -- the
    shift-invert solver to use instead of the direct factorisation: an IterativeSymmetricSolver
    (algebra/hip/shift_invert.py: preconditioned block MINRES on the device blocks, for problems whose
    factors do not fit, and for row-sharded blocks), set up here with A (or `operator`), sigma and B
    unless it already is.  With an inexact solver the converged pairs get a final Rayleigh-Ritz step
    with A itself, so that the eigenvalue errors are the square of the eigenvector errors.

    Returns (lmd, x, status): eigenvalues ascending, eigenvectors as columns, status as
    in the reference (0 success, 1 iteration limit, 2 no search directions, 3/4 some
    requested eigenvalues may not exist, <0 error).'''
    if opt is None:
        opt = Options()
    if buckling and sigma >= 0:
        raise ValueError('sigma must be negative in buckling mode')
    make_vectors = vectors if vectors is not None else (lambda n, data_type: Vectors(n, data_type=data_type))
    if B is not None:
        opB = SparseSymmetricMatrix(A if buckling else B)
    else:
        if buckling:
            raise RuntimeError('stress stiffness matrix missing in buckling mode')
        opB = None

    inexact = None
    if T is None:       # shift-invert through a host factorisation (or an iterative solve: `solver`)
        if isinstance(A, (SparseSymmetricSolver, IterativeSymmetricSolver)):
            solver = A
            n, dtype, sigma = A.size(), A.data_type(), A.sigma()
            inexact = solver if isinstance(solver, IterativeSymmetricSolver) else None
        elif solver is not None:
            inexact = solver
            if solver.operator() is None:
                if verb > -1:
                    print('setting up the iterative linear system solver...')
                mk0 = (lambda n_, nv, data_type: _with(make_vectors(n_, data_type=data_type), nv)) if vectors is not None else None
                solver.analyse(operator if operator is not None else A, sigma, B, vectors=mk0)
                solver.factorize()
            n, dtype, sigma = solver.size(), solver.data_type(), solver.sigma()
        else:
            m, n = A.shape
            if m != n:
                raise ValueError('the matrix must be square')
            dtype = A.data.dtype.type
            solver = SparseSymmetricSolver(dtype=dtype)
            if verb > -1:
                print('setting up the linear system solver...')
            start = time.time()
            solver.analyse(A, sigma, B)
            solver.factorize()
            # estimate the factorization error on three random vectors (partial_hevp.py:126-162)
            opA_ = SparseSymmetricMatrix(A)
            b, x, y = (Vectors(n, 3, data_type=dtype) for _ in range(3))
            x.fill_random()
            opA_.apply(x, b)
            z = x
            if B is not None:
                opB_ = SparseSymmetricMatrix(B)
                opB_.apply(x, y)
                z = y
            s = x.dots(x)
            if sigma != 0:
                b.add(z, -sigma)
            solver.solve(b, y)
            y.add(x, -1)
            t = y.dots(y)
            err = numpy.amax(numpy.sqrt(abs(t / s)))
            if err > 0.01:
                if verb > -1:
                    print('factorization too inaccurate: relative error > %.1e, '
                          'consider moving shift slightly' % err)
                return None, None, -1
            if verb > -1:
                print('estimated factorization error: %.1e' % err)
                print('setup time: %.2e' % (time.time() - start))
        if inexact is not None and inexact.tol is None:
            inexact.tol = min(1e-6, max(1e-12, 0.01 * tol))
        try:
            mk = (lambda n_, nv, data_type: _with(make_vectors(n_, data_type=data_type), nv)) if vectors is not None else None
            if inexact is not None and type(which) is not tuple and not buckling:
                # an integer `which` only asks whether both signs occur (one- or two-sided search): a short Lanczos run
                # instead of the full count (a tenth of its cost; the counts below are then 0 / 1 flags, not numbers)
                has_neg, has_pos = solver.signs(vectors=mk)
                neg, pos = int(has_neg), int(has_pos)
            elif inexact is not None:
                neg, pos = solver.inertia(vectors=mk)
            else:
                neg, pos = solver.inertia()
        except RuntimeError as err:
            if verb > -1:
                print('%s' % err)
            return None, None, -1
        if verb > -1 and (inexact is None or type(which) is tuple or buckling):
            print('positive eigenvalues: %d' % pos)
            print('negative eigenvalues: %d' % neg)
        if type(which) is tuple:
            if len(which) != 2:
                raise ValueError('which must be either integer or tuple of 2 integers')
            which = (min(which[0], neg), min(which[1], pos))
        elif buckling:
            which = (neg, 0) if which < neg else (neg, which - neg)
        elif neg < 1:
            which = (0, which)
        elif pos < 1:
            which = (which, 0)
        eigenvectors = make_vectors(n, data_type=dtype)
        evp = Problem(eigenvectors, solver) if B is None else Problem(eigenvectors, solver, opB, 'pro')
        evp_solver = Solver(evp)
    else:               # preconditioned iterations, everything on the device
        if buckling:
            raise ValueError('preconditioning for buckling problem not supported')
        opA = operator if operator is not None else SparseSymmetricMatrix(A)
        n = opA.size()
        dtype = numpy.dtype(opA.data_type()).type
        eigenvectors = make_vectors(n, data_type=dtype)
        # A x = lambda B x, as the reference's docstring says (partial_hevp.py:32-36).  The reference itself passes the string
        # 'gen' as Problem's `prod` argument here (partial_hevp.py:214), which makes the type 'pro' (solver.py:240-249): its
        # preconditioned generalized mode returns the eigenvalues of A B x = lambda x (tests/golden/known_answers.json:
        # hevp_gen_lap10_ilu5_reference_returns_AB).  The documented problem is solved here.
        evp = Problem(eigenvectors, opA) if B is None else Problem(eigenvectors, opA, opB)
        evp_solver = Solver(evp)
        if T is not True:          # T=True: no preconditioner (identity), blocks never leave the GPU
            evp_solver.set_preconditioner(T)
        sigma = None
        if type(which) is tuple:
            raise ValueError('which must be integer if preconditioning is used')
        which = (which, 0)

    opt.convergence_criteria = DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('k eigenvector error', tol)
    opt.sigma = sigma
    start = time.time()
    status = evp_solver.solve(eigenvectors, opt, which=which)
    if status < 0:
        return None, None, status
    solve_time = time.time() - start
    if T is None:
        lmd = sigma / (1 - 1 / evp_solver.eigenvalues) if buckling else sigma + 1. / evp_solver.eigenvalues
    else:
        lmd = evp_solver.eigenvalues
    ind = numpy.argsort(-lmd) if buckling else numpy.argsort(lmd)
    lmd = lmd[ind]
    if verb > -1:
        print('iterations: %d, solve time: %.2e' % (evp_solver.iteration, solve_time))
    partial_hevp.last = {'iterations': evp_solver.iteration, 'solve_time': solve_time,
                         'residual_norms': evp_solver.residual_norms[ind] if len(ind) else None,
                         'convergence_status': evp_solver.convergence_status[ind] if len(ind) else None,
                         'eigenvector_errors': (evp_solver.eigenvector_errors.kinematic[ind],
                                                evp_solver.eigenvector_errors.residual[ind]) if len(ind) else None}
    if inexact is not None:
        partial_hevp.last.update(inner_solves=inexact.solves, inner_iterations=inexact.iterations,
                                 inner_columns_applied=inexact.columns_applied)
    if inexact is not None and not buckling and eigenvectors.nvec() > 0:
        # (the per-pair records above are those of the iteration, in its order; the refined pairs are a rotation of them)
        lmd, ind = _refine(eigenvectors, inexact.operator(), opB, lmd)
    x = eigenvectors.data().T
    if eigenvectors.nvec() > 0:
        x = x[:, ind]
    return lmd, x, status


def _with(v, nv):
    """An empty Vectors grown to nv vectors (the factories hand out empty blocks)."""
    return v.new_vectors(nv)


def _refine(x, op_a, op_b, lmd):
    """Rayleigh-Ritz with A (and B) itself in the span of the converged vectors: the inexact shift-invert
    operator delivers eigenvectors to its inner tolerance; the Ritz values of the ORIGINAL pencil in their span
    err by the square of that.  X is rotated in place; returns the eigenvalues (ascending) and the identity
    order."""
    import scipy.linalg as sla
    k = x.nvec()
    ax = x.new_vectors(k)
    op_a.apply(x, ax)
    ga = ax.dot(x)
    if op_b is None:
        gb = x.dot(x)
    else:
        bx = x.new_vectors(k)
        op_b.apply(x, bx)
        gb = bx.dot(x)
    ga, gb = 0.5 * (ga + ga.conj().T), 0.5 * (gb + gb.conj().T)
    lam, q = sla.eigh(ga, gb)
    x.multiply(q.astype(x.data_type()), ax)
    ax.copy(x)
    return lam, numpy.arange(k)
