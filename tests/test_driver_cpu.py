"""CPU tier: this repository's block-JCG driver and partial_hevp (host logic over
tests/fake_lib.py) against the reference's known answers (tests/golden/known_answers.json)
and the analytic Laplacian spectrum.  Eigenvalues: 1e-10 relative (BASELINE north star)."""

import json
import os

import numpy as np
import pytest

import fake_lib


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


def known(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'known_answers.json')))


def test_core_doctest_problem(golden_dir):
    from raleigh_amd.core.solver import Options, Problem, Solver, DefaultConvergenceCriteria
    from raleigh_amd.algebra.hip import Vectors, Matrix
    k = known(golden_dir)['core_diag100']
    np.random.seed(1)
    n = 100
    opt = Options()
    opt.convergence_criteria = DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('eigenvector error', 1e-8)
    opt.verbosity = -1
    v = Vectors(n, data_type=np.float64)
    solver = Solver(Problem(v, Matrix(np.diag(np.arange(1, n + 1).astype(np.float64)))))
    assert solver.solve(v, opt, which=(6, 0)) == 0
    assert solver.block_size == k['block_size']
    assert np.allclose(solver.eigenvalues, k['eigenvalues'], rtol=1e-10)
    assert abs(solver.iteration - k['iterations']) <= 0.2 * k['iterations']
    x = v.data()
    assert np.allclose(x @ x.T, np.eye(6), atol=1e-8)


def test_partial_hevp_identity_preconditioner(golden_dir):
    from raleigh_amd.interfaces import partial_hevp
    from oracle.sparse import lap3d
    k = known(golden_dir)['hevp_lap12_id5']
    A = lap3d(12, 11, 10, 1.0, 1.01, 1.02)
    np.random.seed(1)
    from raleigh_amd.core.solver import Options
    opt = Options()
    opt.max_iter = 500
    lmd, x, status = partial_hevp(A, T=True, which=5, tol=1e-8, verb=-1, opt=opt)
    assert status == 0
    assert np.allclose(lmd[:5], k['eigenvalues'], rtol=1e-10)
    r = A @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-5


def test_partial_hevp_shift_invert_config1(golden_dir):
    """BASELINE config 1: lap3d(30,30,30,1,1.01,1.02), sigma = 0, 6 eigenvalues, tol 1e-6."""
    from raleigh_amd.interfaces import partial_hevp
    from oracle.sparse import lap3d
    k = known(golden_dir)['hevp_lap30_si6']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    np.random.seed(1)
    lmd, x, status = partial_hevp(A, sigma=0, which=6, tol=1e-6, verb=-1)
    assert status == 0
    assert np.allclose(lmd[:6], k['eigenvalues'], rtol=1e-10)


def test_partial_hevp_ilu_config3_mode(golden_dir):
    """Preconditioned mode of config 3 on the Laplacian stand-in: 10 smallest with the ILUT preconditioner."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    from oracle.sparse import lap3d
    k = known(golden_dir)['hevp_lap30_ilu10']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    np.random.seed(1)
    T = IncompleteLU(A)
    T.factorize()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    assert status == 0
    assert np.allclose(lmd[:10], k['eigenvalues'], rtol=1e-8)
    r = A @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) <= 10 * max(k['residual_norms'])


@pytest.mark.parametrize('dt', [np.complex128, np.float32])
def test_dense_both_ends_and_largest(dt):
    from raleigh_amd.core.solver import Options, Problem, Solver, DefaultConvergenceCriteria
    from raleigh_amd.algebra.hip import Vectors, Matrix
    rng = np.random.default_rng(4)
    n = 80
    H = rng.standard_normal((n, n))
    if dt == np.complex128:
        H = H + 1j * rng.standard_normal((n, n))
    H = ((H + H.conj().T) / 2 + np.diag(np.arange(n) * 2.0)).astype(dt)
    exact = np.linalg.eigvalsh(H.astype(np.complex128))
    tol = 1e-9 if dt == np.complex128 else 2e-4
    for which, expect in (((3, 2), np.concatenate((exact[:3], exact[-2:]))), (4, None)):
        np.random.seed(1)
        opt = Options()
        opt.convergence_criteria = DefaultConvergenceCriteria()
        opt.convergence_criteria.set_error_tolerance('residual', 1e-10 if dt == np.complex128 else 1e-5)
        opt.verbosity = -1
        opt.max_iter = 600
        v = Vectors(n, data_type=dt)
        solver = Solver(Problem(v, Matrix(np.ascontiguousarray(H))))
        status = solver.solve(v, opt, which=which)
        assert status == 0
        got = np.sort(solver.eigenvalues)
        if expect is None:      # `which` largest in modulus
            expect = exact[np.argsort(-np.abs(exact))[:which]]
            assert len(got) >= which
            for e in expect:
                assert np.min(np.abs(got - e)) < tol * np.abs(e)
        else:
            assert np.allclose(got, np.sort(expect), rtol=tol)


def test_generalized_problem():
    from raleigh_amd.core.solver import Options, Problem, Solver, DefaultConvergenceCriteria
    from raleigh_amd.algebra.hip import Vectors, Matrix
    import scipy.linalg as sla
    rng = np.random.default_rng(6)
    n = 70
    A = np.diag(np.arange(1.0, n + 1))
    C = rng.standard_normal((n, n)) * 0.05
    B = np.eye(n) * 2.0 + (C + C.T) / 2
    exact = sla.eigh(A, B, eigvals_only=True)
    np.random.seed(1)
    opt = Options()
    opt.convergence_criteria = DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('kinematic eigenvector error', 1e-8)
    opt.verbosity = -1
    opt.max_iter = 600
    v = Vectors(n, data_type=np.float64)
    solver = Solver(Problem(v, Matrix(A), Matrix(np.ascontiguousarray(B))))
    assert solver.solve(v, opt, which=(4, 0)) == 0
    assert np.allclose(np.sort(solver.eigenvalues), exact[:4], rtol=1e-9)


def test_small_problem_falls_through_to_dense_rr():
    from raleigh_amd.core.solver import Options, Problem, Solver
    from raleigh_amd.algebra.hip import Vectors, Matrix
    n = 12
    np.random.seed(1)
    A = np.diag(np.arange(1.0, n + 1))
    opt = Options()
    opt.verbosity = -1
    v = Vectors(n, data_type=np.float64)
    solver = Solver(Problem(v, Matrix(A)))
    assert solver.solve(v, opt, which=(3, 0)) == 0
    assert np.allclose(np.sort(solver.eigenvalues), np.arange(1.0, n + 1), rtol=1e-10)


def test_pca_npc_against_reference_known_answer(golden_dir):
    """pca(generate(600, 400, 200, pca=True), npc=30): singular values vs the exact SVD of the
    shifted data (1e-3 relative, the svtol class) and the reference's error figures."""
    from raleigh_amd.interfaces import pca, pca_error
    from oracle.pca_data import generate
    k = known(golden_dir)['pca_600x400_npc30']
    np.random.seed(1)
    A, sigma, u, v = generate(600, 400, 200, pca=True)
    mean, trans, comps = pca(A, npc=30)
    assert mean.shape == (1, 400) and trans.shape == (600, 30) and comps.shape == (30, 400)
    assert np.allclose(mean, A.mean(axis=0, keepdims=True), atol=1e-6)
    sv = np.linalg.norm(trans, axis=0)
    exact = np.array(k['sigma_exact'])
    assert np.max(np.abs(sv - exact) / exact[0]) < 1e-3
    assert np.allclose(comps @ comps.T, np.eye(30), atol=1e-3)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef <= 1.05 * k['ef'] and em <= 1.2 * k['em']


def test_pca_tolerance_and_transposed():
    from raleigh_amd.interfaces import pca, pca_error
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(300, 500, 120, pca=True)       # fewer samples than features
    mean, trans, comps = pca(A, tol=0.1)
    em, ef = pca_error(A, mean, trans, comps)
    assert ef <= 0.1 * 1.02
    assert trans.shape[1] == comps.shape[0] and comps.shape[1] == 500
    # fewer samples than features: the components are A_s^T v, orthonormal only after the refinement
    # pca.py:146-147 asks for (2.6e-2 off without it)
    k = comps.shape[0]
    assert np.abs(comps @ comps.T - np.eye(k)).max() < 1e-5
    mean, trans, comps = pca(A, npc=20)
    assert np.abs(comps @ comps.T - np.eye(20)).max() < 1e-5
    G = trans.T @ trans
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-5 * G[0, 0]


def test_complex_hermitian_shift_invert_config5_shape():
    """BASELINE config 5 in miniature: Hermitian complex128 sparse matrix (Laplacian + i * skew
    first-neighbour term), eigenvalues nearest an interior shift by shift-invert."""
    import scipy.sparse as sp
    from raleigh_amd.interfaces import partial_hevp
    from oracle.sparse import lap3d
    A = lap3d(12, 12, 11, 1.0, 1.01, 1.02)
    n = A.shape[0]
    S = sp.diags([np.full(n - 1, 0.3)], [1], shape=(n, n))
    H = sp.csr_matrix(A.astype(np.complex128) + 1j * S - 1j * S.T)
    exact = np.linalg.eigvalsh(H.toarray())
    sigma = 0.5 * (exact[40] + exact[41])
    np.random.seed(1)
    lmd, x, status = partial_hevp(H, sigma=sigma, which=8, tol=1e-8, verb=-1)
    assert status == 0 and len(lmd) >= 8
    nearest = exact[np.argsort(np.abs(exact - sigma))[:8]]
    for e in nearest[:6]:
        assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e)
    r = H @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-5


def test_inertia_with_a_zero_diagonal_entry():
    """An indefinite matrix with a zero diagonal entry: SuperLU interchanges rows even with diag_pivot_thresh = 0,
    diag(U) then has the wrong signs, so method='superlu' must refuse to report an inertia.  The L D L^H factors
    (default) take a 2 x 2 pivot there (what PARDISO's mtype -2 does, mkl_wrap.py:354-384): the inertia is exact and
    partial_hevp maps `which` correctly."""
    import scipy.sparse as sp
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    from raleigh_amd.interfaces import partial_hevp
    n = 40
    d = np.linspace(1.0, 3.0, n)
    d[0] = 0.0                      # zero pivot in the first elimination step
    A = sp.diags([np.ones(n - 1), d, np.ones(n - 1)], [-1, 0, 1], format='csr')
    ev = np.linalg.eigvalsh(A.toarray())
    solver = SparseSymmetricSolver(method='superlu')
    solver.analyse(A, 0.0)
    solver.factorize()
    with pytest.raises(RuntimeError):
        solver.inertia()
    lmd, x, status = partial_hevp(solver, which=3, verb=-1)
    assert status == -1 and lmd is None
    solver = SparseSymmetricSolver()
    solver.analyse(A, 0.0)
    solver.factorize()
    assert solver.inertia() == (int(np.sum(ev < 0)), int(np.sum(ev > 0)))
    np.random.seed(1)
    lmd, x, status = partial_hevp(A, sigma=0.0, which=3, tol=1e-9, verb=-1)
    assert status == 0
    near = ev[np.argsort(np.abs(ev))[:3]]                # (the solver may return more than it was asked for)
    assert len(lmd) >= 3 and all(np.min(np.abs(lmd - v)) < 1e-8 for v in near)
    # a definite shift of the same matrix: both factorisations count alike
    for method in ('superlu', 'ldlt'):
        solver = SparseSymmetricSolver(method=method)
        solver.analyse(A, -1.0)
        solver.factorize()
        assert solver.inertia() == (int(np.sum(ev < -1.0)), n - int(np.sum(ev < -1.0)))


def test_device_chebyshev_preconditioner():
    """Polynomial preconditioner built from the operator itself: same eigenvalues, far fewer iterations."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip import SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    from oracle.sparse import lap3d, lap3d_eigenvalues
    A = lap3d(20, 19, 18, 1.0, 1.01, 1.02)
    ana = lap3d_eigenvalues(20, 19, 18, 1.0, 1.01, 1.02, 6)
    iters = {}
    for name, T in (('none', True), ('cheb', ChebyshevPreconditioner(SparseSymmetricMatrix(A),
                                                                    gershgorin_upper_bound(A), ratio=100, degree=6))):
        np.random.seed(1)
        opt = Options()
        opt.max_iter = 2000
        lmd, x, status = partial_hevp(A, T=T, which=6, tol=1e-7, verb=-1, opt=opt)
        assert status == 0
        assert np.max(np.abs(lmd[:6] - ana) / ana) < 1e-10
        iters[name] = partial_hevp.last['iterations']
    assert iters['cheb'] * 3 < iters['none']


def test_fused_chebyshev_step_equals_unfused():
    """rlh_spmm_cheb (p = cy y + cp p + cb (b - A y), p in place) against the same step written out."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from oracle.sparse import lap3d
    A = lap3d(9, 8, 7, 1.0, 1.01, 1.02)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    y0, p0, b0 = (rng.standard_normal((5, n)) for _ in range(3))
    op = SparseSymmetricMatrix(A)
    y, p, b = Vectors(y0.copy()), Vectors(p0.copy()), Vectors(b0.copy())
    op.cheb_step(y, p, b, 1.3, -0.3, -1.7)
    want = 1.3 * y0 - 0.3 * p0 - 1.7 * (b0 - (A @ y0.T).T)
    assert np.allclose(p.data(), want, rtol=1e-13, atol=1e-12)
    assert np.array_equal(y.data(), y0) and np.array_equal(b.data(), b0)


def test_mixed_precision_chebyshev_preconditioner():
    """The polynomial evaluated in float32 (converted on the device): fp64 eigenvalues are unaffected."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip import SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    from oracle.sparse import lap3d, lap3d_eigenvalues
    A = lap3d(20, 19, 18, 1.0, 1.01, 1.02)
    ana = lap3d_eigenvalues(20, 19, 18, 1.0, 1.01, 1.02, 6)
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 500
    T = ChebyshevPreconditioner(None, gershgorin_upper_bound(A), ratio=100, degree=6,
                                low_precision_op=SparseSymmetricMatrix(A.astype(np.float32)))
    lmd, x, status = partial_hevp(A, T=T, which=6, tol=1e-7, verb=-1, opt=opt)
    assert status == 0
    assert np.max(np.abs(lmd[:6] - ana) / ana) < 1e-10
    assert partial_hevp.last['iterations'] < 60


def test_bf16_storage_chebyshev_preconditioner():
    """Preconditioner work blocks in bfloat16 (float32 arithmetic): same eigenvalues, about the same
    iteration count as float32 storage."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip import SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    from raleigh_amd import _lib
    from oracle.sparse import lap3d, lap3d_eigenvalues
    A = lap3d(20, 19, 18, 1.0, 1.01, 1.02)
    ana = lap3d_eigenvalues(20, 19, 18, 1.0, 1.01, 1.02, 6)
    its = {}
    for storage in (None, 'bf16'):
        np.random.seed(1)
        opt = Options()
        opt.max_iter = 500
        T = ChebyshevPreconditioner(None, gershgorin_upper_bound(A), ratio=100, degree=6,
                                    low_precision_op=SparseSymmetricMatrix(A.astype(np.float32)), storage=storage)
        lmd, x, status = partial_hevp(A, T=T, which=6, tol=1e-7, verb=-1, opt=opt)
        assert status == 0
        assert np.max(np.abs(lmd[:6] - ana) / ana) < 1e-10
        its[storage] = partial_hevp.last['iterations']
    assert _lib.library().calls.get('spmm_cheb_bf16', 0) > 10
    assert its['bf16'] <= its[None] + 3


def test_host_round_trips_per_iteration(fake, monkeypatch):
    """Fused Gram pairs and batched reductions: in the steady state -- before the
    first pair is locked -- the driver synchronises with the device five times per iteration (Ritz-pair
    check + residual norms; conjugation coefficients; projection on X; Gram of (X, Y); A-Gram of
    (X, Y)) where the reference's call sequence makes 13 blocking calls (raleigh/core/solver.py:854-861,
    968-974, 1321-1339, 1360, 1376-1381, 1444-1447), and the results do not change."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip import Vectors
    from oracle.sparse import lap3d
    A = lap3d(12, 11, 10, 1.0, 1.01, 1.02)

    def run():
        np.random.seed(1)
        opt = Options()
        opt.max_iter = 30
        before = dict(fake.calls)
        lmd, x, status = partial_hevp(A, T=True, which=5, tol=1e-13, verb=-1, opt=opt)
        its = partial_hevp.last['iterations']
        return its, fake.calls.get('sync', 0) - before.get('sync', 0), evp_values()

    def evp_values():
        return None
    its, syncs, _ = run()
    assert its == 30                                   # the tolerance is out of reach: no locking, steady state
    assert syncs <= 5 * its + 8                        # 5 per iteration + the set-up (start block, initial Rayleigh-Ritz)
    assert fake.calls.get('gram_multi', 0) >= 3 * its  # the stacked Grams ran
    # the same run with one blocking call per reduction, as the reference issues them
    monkeypatch.delattr(Vectors, 'reduction_batch')
    its2, syncs2, _ = run()
    assert its2 == its and syncs2 >= 11 * its


def test_batched_and_unbatched_drivers_agree(monkeypatch):
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip import Vectors
    from oracle.sparse import lap3d, lap3d_eigenvalues
    A = lap3d(12, 11, 10, 1.0, 1.01, 1.02)
    exact = lap3d_eigenvalues(12, 11, 10, 1.0, 1.01, 1.02, 6)
    out = []
    for batched in (True, False):
        if not batched:
            monkeypatch.delattr(Vectors, 'reduction_batch')
        np.random.seed(1)
        from raleigh_amd.core.solver import Options
        opt = Options()
        opt.max_iter = 500
        lmd, x, status = partial_hevp(A, T=True, which=6, tol=1e-8, verb=-1, opt=opt)
        assert status == 0 and np.allclose(lmd[:6], exact, rtol=1e-10)
        out.append((lmd[:6], partial_hevp.last['iterations']))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-12)
    assert abs(out[0][1] - out[1][1]) <= max(2, out[1][1] // 20)


def test_shift_invert_moves_no_block_across_pcie(fake, golden_dir):
    """Shift-invert with the solves on the device (SuperLU factors + permutations as a chain of
    level-scheduled triangular solves): the reference's known answer for config 1 (17 iterations, six
    eigenvalues) with no block upload / download between the set-up and the final read-back -- and the
    same eigenvalues with the host solve, which moves two blocks per application."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    from oracle.sparse import lap3d
    k = known(golden_dir)['hevp_lap30_si6']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    out = {}
    for device in (True, False):
        solver = SparseSymmetricSolver(device=device, method='ldlt' if device else 'superlu')
        solver.analyse(A, 0.0)
        solver.factorize()
        np.random.seed(1)
        before = fake.calls.get('block_transfer', 0)
        lmd, x, status = partial_hevp(solver, which=6, tol=1e-6, verb=-1)
        out[device] = (lmd, partial_hevp.last['iterations'], fake.calls.get('block_transfer', 0) - before)
        assert status == 0 and np.allclose(lmd[:6], k['eigenvalues'], rtol=1e-10)
    assert abs(out[True][1] - 17) <= 3
    # the pivot order (twice) and the two arrays of D^-1 at set-up, start block in, eigenvectors out: nothing per iteration
    assert out[True][2] <= 6
    assert out[False][2] >= 2 * out[False][1]             # the host solve: two transfers per application


def test_pivoted_cholesky_condition_control():
    """The kept block of the pivoted factor is well conditioned (reciprocal condition number of U^H U above eps),
    reproduces its block of the permuted Gram matrix, and everything dropped is zero -- for spectra decaying over 2 to
    14 orders of magnitude, with and without an unpivoted leading block (the bisection over the number of kept
    columns replaces the reference's one-at-a-time search, solver.py:1749-1826)."""
    from raleigh_amd.core.solver import _pivoted_cholesky
    rng = np.random.default_rng(7)
    seen_drop = False
    for trial in range(16):
        n, k = 24, int(rng.integers(0, 8))
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        d = np.logspace(0, -rng.uniform(2, 14), n)
        G = (Q * d) @ Q.T
        G = (G + G.T) / 2
        eps = 1e-8
        U, ind, dropped = _pivoted_cholesky(G, k, eps)
        kept = n - k - dropped
        seen_drop = seen_drop or dropped > 0
        Gp = G[np.ix_(ind, ind)]
        if k + kept > 0:
            blk = U[:k + kept, :k + kept]
            sv = np.linalg.svd(blk, compute_uv=False)
            assert (sv[-1] / sv[0]) ** 2 > eps
            assert np.allclose(blk.T @ blk, Gp[:k + kept, :k + kept], atol=1e-10 * np.abs(G).max())
        assert not U[k + kept:].any() and not U[:, k + kept:].any()
    assert seen_drop
