"""CPU tier: the inexact shift-invert operator (raleigh_amd/algebra/hip/shift_invert.py: preconditioned block MINRES
behind the surface of the reference's SparseSymmetricSolver, raleigh/algebra/sparse_mkl.py:51-119) over tests/fake_lib.py:
the linear solves against dense ones, the Lanczos inertia count against the dense spectrum, and partial_hevp's
shift-invert mode (raleigh/interfaces/partial_hevp.py:103-200) against closed-form / dense eigenvalues to 1e-10."""

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

import fake_lib


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


def hermitian(N, dtype):
    from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues
    from oracle.sparse import lap3d
    n = N ** 3
    if np.dtype(dtype).kind == 'c':
        return hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n).astype(dtype), hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
    A = sp.csr_matrix(lap3d(N, N, N, 1.0, 1.01, 1.02)).astype(dtype)
    return A, np.linalg.eigvalsh(A.toarray().astype(np.float64))


class _Inverse:
    """A dense Hermitian positive definite preconditioner on device Vectors (test helper)."""

    def __init__(self, mat):
        self.inv = np.linalg.inv(mat)

    def apply(self, x, y):
        y.fill(np.ascontiguousarray((self.inv @ x.data().T).T))


@pytest.mark.parametrize('dtype', [np.float64, np.complex128])
@pytest.mark.parametrize('precond', [False, True])
def test_block_minres_solves_an_indefinite_system(dtype, precond):
    """K = A - sigma I with 11 negative eigenvalues, 16 right-hand sides of which one is zero and one a multiple of
    another: every column to 1e-9 in the 2-norm against the dense solve, and the block narrows instead of breaking down."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.shift_invert import ShiftedOperator, block_minres
    A, exact = hermitian(8, dtype)
    n = A.shape[0]
    sigma = 0.5 * (exact[10] + exact[11])
    rng = np.random.default_rng(0)
    m = 16
    b = rng.standard_normal((m, n)).astype(dtype)
    if np.dtype(dtype).kind == 'c':
        b = b + 1j * rng.standard_normal((m, n))
    b[3] = 2 * b[2]
    b[5] = 0
    K = ShiftedOperator(SparseSymmetricMatrix(A), sigma)
    B, X = Vectors(b.copy()), Vectors(n, m, data_type=dtype)
    M = _Inverse(A.toarray()) if precond else None
    info = block_minres(K, B, X, precond=M, tol=1e-11, max_iter=300)
    assert info.converged
    assert info.columns_applied <= 14 * info.iterations          # the zero and the dependent column cost nothing
    ref = np.linalg.solve(A.toarray() - sigma * np.eye(n), b.T).T
    err = np.linalg.norm(X.data() - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)
    assert np.max(err[np.arange(m) != 5]) < 1e-9 and np.max(np.abs(X.data()[5])) < 1e-15
    # a per-column tolerance: loose columns stop the iteration no earlier than the tight ones need
    tol = np.full(m, 1e-3)
    tol[0] = 1e-11
    info2 = block_minres(K, B, X, precond=M, tol=tol, max_iter=300)
    assert info2.converged and info2.residuals[0] <= 1e-11 and info2.iterations <= info.iterations


def test_block_minres_reports_failure_and_zero_right_hand_sides():
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.shift_invert import ShiftedOperator, IterativeSymmetricSolver, block_minres
    A, exact = hermitian(6, np.float64)
    n = A.shape[0]
    K = ShiftedOperator(SparseSymmetricMatrix(A), 0.5 * (exact[40] + exact[41]))
    B, X = Vectors(n, 4), Vectors(n, 4)
    X.fill_random()
    info = block_minres(K, B, X)                 # b = 0: x = 0, no operator application
    assert info.converged and info.iterations == 0 and np.all(X.data() == 0)
    B.fill_random()
    info = block_minres(K, B, X, tol=1e-12, max_iter=3)
    assert not info.converged and info.iterations == 3 and np.max(info.residuals) > 1e-12
    sol = IterativeSymmetricSolver(dtype=np.float64, max_iter=3)
    sol.analyse(A, 0.5 * (exact[40] + exact[41]))
    sol.factorize()
    with pytest.raises(RuntimeError, match='did not reach'):
        sol.solve(B, X)
    with pytest.raises(ValueError, match='positive definite'):
        IterativeSymmetricSolver(dtype=np.float64, preconditioner='chebyshev').factorize()


@pytest.mark.parametrize('below', [0, 7, 23])
def test_lanczos_inertia_count(below):
    """The number of negative eigenvalues of A - sigma I from the probe solve's projected operator (copies of converged Ritz
    values filtered by their weight in the first block) = the dense count."""
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    A, exact = hermitian(9, np.complex128)
    n = A.shape[0]
    sigma = 0.5 * (exact[below - 1] + exact[below]) if below else 0.5 * exact[0]
    np.random.seed(2)
    sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=6, ratio=20.0)
    sol.analyse(A, sigma)
    sol.factorize()
    assert sol.inertia() == (below, n - below)


@pytest.mark.parametrize('dtype', [np.float64, np.complex128])
def test_partial_hevp_inexact_shift_invert(dtype):
    """partial_hevp(sigma=..., solver=IterativeSymmetricSolver): the eigenvalues nearest an interior shift to 1e-10 of
    the closed-form / dense spectrum, residuals small, both forms of `which`."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    A, exact = hermitian(10, dtype)
    sigma = 0.5 * (exact[12] + exact[13])
    for which, want in ((8, exact[np.argsort(np.abs(exact - sigma))[:6]]), ((3, 4), exact[10:17])):
        opt = Options()
        opt.block_size = 16
        np.random.seed(1)
        sol = IterativeSymmetricSolver(dtype=dtype, pos_def=True, degree=8, ratio=20.0)
        lmd, x, status = partial_hevp(A, sigma=sigma, which=which, tol=1e-6, verb=-1, opt=opt, solver=sol)
        assert status == 0
        for e in want:
            assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e), (which, e)
        r = A @ x - x * lmd
        assert np.max(np.linalg.norm(r, axis=0)) < 1e-5 * np.abs(exact).max()
        assert np.allclose(x.conj().T @ x, np.eye(len(lmd)), atol=1e-8)
        assert partial_hevp.last['inner_solves'] == sol.solves > 2 and sol.iterations > sol.solves


def test_partial_hevp_inexact_shift_invert_generalized():
    """A x = lambda B x with a diagonal positive B through the inexact operator (the 'pro' form of
    raleigh/interfaces/partial_hevp.py:196-200: (A - sigma B)^-1 B): eigenvalues nearest the shift against scipy's
    dense generalized solver."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    A, _ = hermitian(8, np.float64)
    n = A.shape[0]
    rng = np.random.default_rng(4)
    B = sp.diags([1.0 + rng.random(n)], [0], format='csr')
    exact = sla.eigh(A.toarray(), B.toarray(), eigvals_only=True)
    sigma = 0.5 * (exact[9] + exact[10])
    opt = Options()
    opt.block_size = 16
    np.random.seed(1)
    sol = IterativeSymmetricSolver(dtype=np.float64, pos_def=True, degree=8, ratio=20.0)
    lmd, x, status = partial_hevp(A, B=B, sigma=sigma, which=(3, 3), tol=1e-7, verb=-1, opt=opt, solver=sol)
    assert status == 0
    for e in exact[7:13]:
        assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e)
    r = A @ x - (B @ x) * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-5 * np.abs(exact).max()


@pytest.mark.parametrize('dtype', [np.float32, np.complex64])
def test_single_precision_blocks(dtype):
    """Blocks of float32 / complex64: the solve to 1e-4 (the Gram matrices the block is split by are only good to a few
    float32 eps, so directions below 3e-3 of their block's largest are dropped) and the eigenvalues nearest the shift to
    1e-5 (fp32 sigma class of SURVEY 8c)."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.shift_invert import ShiftedOperator, IterativeSymmetricSolver, block_minres
    from oracle.sparse import lap3d
    A64 = sp.csr_matrix(lap3d(10, 10, 10, 1.0, 1.01, 1.02))
    exact = np.linalg.eigvalsh(A64.toarray())
    A = A64.astype(dtype)
    n = A.shape[0]
    sigma = float(0.5 * (exact[10] + exact[11]))
    rng = np.random.default_rng(0)
    b = rng.standard_normal((8, n)).astype(dtype)
    B, X = Vectors(b.copy()), Vectors(n, 8, data_type=dtype)
    info = block_minres(ShiftedOperator(SparseSymmetricMatrix(A), sigma), B, X, tol=1e-4, max_iter=400)
    assert info.converged
    r = (A64 - sigma * sp.identity(n)) @ X.data().T.astype(np.complex128) - b.T
    assert np.max(np.linalg.norm(r, axis=0) / np.linalg.norm(b, axis=1)) < 5e-4
    np.random.seed(1)
    sol = IterativeSymmetricSolver(dtype=dtype, pos_def=True, degree=6, ratio=20.0, tol=1e-5)
    opt = Options()
    opt.block_size = 16
    lmd, x, status = partial_hevp(A, sigma=sigma, which=6, tol=1e-3, verb=-1, opt=opt, solver=sol)
    assert status == 0
    for e in exact[np.argsort(np.abs(exact - sigma))[:4]]:
        assert np.min(np.abs(lmd - e)) < 1e-5 * abs(e)


def test_spectrum_bound_of_a_ready_operator():
    """A ready operator (no matrix to take a Gershgorin bound from): the upper end of the Chebyshev interval from twelve
    Lanczos steps -- above the largest eigenvalue, within 30 % of it -- and the solve goes through."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    A, exact = hermitian(9, np.complex128)
    n = A.shape[0]
    np.random.seed(3)
    sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=6, ratio=20.0)
    sol.analyse(SparseSymmetricMatrix(A), 0.5 * (exact[6] + exact[7]))
    sol.factorize()
    assert exact[-1] < sol.hi < 1.3 * exact[-1]
    b, x = Vectors(n, 5, data_type=np.complex128), Vectors(n, 5, data_type=np.complex128)
    b.fill_random()
    sol.solve(b, x)
    assert sol.last.converged and sol.inertia() == (7, n - 7)
