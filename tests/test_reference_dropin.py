"""Drop-in check (build container only): the UNMODIFIED reference solver
(raleigh/core/solver.py) drives this repository's Vectors / Matrix /
SparseSymmetricMatrix.  The arithmetic goes through tests/fake_lib.py (no GPU in this
tier), so what is verified is the interface: every method, argument convention and
host/device hand-off the reference solver relies on.  Skipped where the reference is
not mounted (it never travels to the GPU box)."""

import json
import os
import sys

import numpy as np
import pytest
import scipy.linalg as sla

REF = os.environ.get('RALEIGH_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'raleigh')),
                                reason='reference tree not present')

import fake_lib


@pytest.fixture()
def ref_solver(monkeypatch):
    monkeypatch.syspath_prepend(REF)
    real_eigh = sla.eigh

    def eigh(*a, **kw):            # scipy >= 1.14 dropped the turbo keyword the reference passes
        kw.pop('turbo', None)
        return real_eigh(*a, **kw)
    monkeypatch.setattr(sla, 'eigh', eigh)
    fake = fake_lib.install()
    import raleigh.core.solver as rs
    yield rs, fake
    fake_lib.uninstall()
    for k in [k for k in sys.modules if k == 'raleigh' or k.startswith('raleigh.')]:
        del sys.modules[k]


def test_core_solver_doctest_problem(ref_solver, golden_dir):
    """raleigh/examples/core_solver.py:65-71: diag(1..100), 6 left eigenvalues, tol 1e-8."""
    rs, fake = ref_solver
    from raleigh_amd.algebra.hip import Vectors, Matrix
    known = json.load(open(os.path.join(golden_dir, 'known_answers.json')))['core_diag100']
    np.random.seed(1)
    n = 100
    opt = rs.Options()
    opt.convergence_criteria = rs.DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('eigenvector error', 1e-8)
    opt.verbosity = -1
    v = Vectors(n, data_type=np.float64)
    A = Matrix(np.diag(np.arange(1, n + 1).astype(np.float64)))
    solver = rs.Solver(rs.Problem(v, A))
    status = solver.solve(v, opt, which=(6, 0))
    assert status == 0
    assert np.allclose(solver.eigenvalues, known['eigenvalues'], rtol=1e-10)
    assert solver.iteration == known['iterations']          # same start vectors, same arithmetic
    assert v.nvec() == 6
    x = v.data()
    assert np.allclose(x @ x.T, np.eye(6), atol=1e-8)
    assert fake.calls['gram'] > 100 and fake.calls['block_update'] > 100


def test_sparse_laplacian_with_reference_solver(ref_solver):
    rs, fake = ref_solver
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from oracle.sparse import lap3d, lap3d_eigenvalues
    A = lap3d(9, 8, 7, 1.0, 1.01, 1.02)
    n = A.shape[0]
    np.random.seed(1)
    opt = rs.Options()
    opt.convergence_criteria = rs.DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('k eigenvector error', 1e-8)
    opt.verbosity = -1
    opt.max_iter = 300
    v = Vectors(n, data_type=np.float64)
    solver = rs.Solver(rs.Problem(v, SparseSymmetricMatrix(A)))
    status = solver.solve(v, opt, which=(4, 0))
    assert status == 0
    lam = np.sort(solver.eigenvalues)[:4]
    assert np.allclose(lam, lap3d_eigenvalues(9, 8, 7, 1.0, 1.01, 1.02, 4), rtol=1e-10)
    assert fake.calls['spmm'] > 5


def test_complex_hermitian_with_reference_solver(ref_solver):
    """Hermitian complex128 problem, largest eigenvalues (the 'largest' branch of solve())."""
    rs, fake = ref_solver
    from raleigh_amd.algebra.hip import Vectors, Matrix
    rng = np.random.default_rng(2)
    n = 60
    H = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    H = (H + H.conj().T) / 2 + np.diag(np.arange(n) * 1.0)
    np.random.seed(1)
    opt = rs.Options()
    opt.convergence_criteria = rs.DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('residual', 1e-10)
    opt.verbosity = -1
    opt.max_iter = 500
    v = Vectors(n, data_type=np.complex128)
    solver = rs.Solver(rs.Problem(v, Matrix(np.ascontiguousarray(H))))
    status = solver.solve(v, opt, which=(0, 3))
    assert status == 0
    exact = np.linalg.eigvalsh(H)
    assert np.allclose(np.sort(solver.eigenvalues), exact[-3:], rtol=1e-9)


def test_reference_lower_rank_approximation_compute_and_update(ref_solver):
    """The UNMODIFIED reference LowerRankApproximation (raleigh/interfaces/lra.py:79-379: compute, then update with
    new rows) on this repository's AMatrix / Vectors: everything that path asks of the backend -- as_vectors() as a
    view the update overwrites, append along both axes, svd, orthogonalize, new_vectors(ndarray), data()."""
    rs, fake = ref_solver
    from raleigh.interfaces.lra import LowerRankApproximation as RefLRA
    from raleigh_amd.algebra.dense_matrix import AMatrix
    from raleigh_amd.interfaces import pca_error
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(600, 400, 200, pca=True)
    A0, A1 = A[:480], A[480:]
    lra = RefLRA()
    lra.compute(AMatrix(A0), opt=rs.Options(), tol=0.05, shift=True)
    em, ef = pca_error(A0, lra.mean(), lra.left(), lra.right())
    assert ef <= 0.05
    lra.update(AMatrix(A1, copy_data=True), opt=rs.Options(), tol=0.05)
    L, R = lra.left(), lra.right()
    assert L.shape[0] == 600 and R.shape[1] == 400 and L.shape[1] == R.shape[0]
    assert np.allclose(lra.mean(), A.mean(axis=0, keepdims=True), atol=1e-6)
    assert np.abs(R @ R.T - np.eye(R.shape[0])).max() < 2e-4
    em, ef = pca_error(A, lra.mean(), L, R)
    assert ef <= 0.05
