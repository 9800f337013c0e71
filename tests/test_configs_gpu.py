"""The BASELINE.json configurations VERDICT r01 lists as never run on the GPU, as -m gpu tests:
config 3 (FE-like shipsec5 surrogate, n = 179 860, ~55 entries per row, fp64, 10 smallest with ILU),
config 5 (complex128 Hermitian operator, block of 64, pairs nearest an interior shift) and config 2
at full size (pca of a dense 20 000 x 20 000 fp32 matrix, 200 components)."""

import json
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def real_library():
    from raleigh_amd import _lib
    _lib.set_library(None)
    L = _lib.lib()
    import ctypes
    assert isinstance(L, ctypes.CDLL), 'the GPU tier must run on the native library'
    yield


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_config3_fe_surrogate_ilu_ten_smallest(golden_dir):
    """partial_hevp(A, T=IncompleteLU, which=10) on the config-3 surrogate, every block on the device
    (operator in the interleaved windowed layout, ILUT factors applied by level-scheduled triangular
    solves): eigenvalues against SciPy's shift-invert eigsh (fixture) to 1e-10, residuals, block 16."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    from raleigh_amd.synthetic import fe_surrogate
    exact = np.array(json.load(open(os.path.join(golden_dir, 'fe_surrogate_eigs.json')))['eigenvalues'])
    A = fe_surrogate()
    np.random.seed(1)
    T = IncompleteLU(A)
    t0 = time.time()
    T.factorize()
    t_fact = time.time() - t0
    t0 = time.time()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    t_solve = time.time() - t0
    print('config 3: ILUT %.1f s (fill %.2f, levels %s), solve %.2f s, %d iterations'
          % (t_fact, T.fill, T.levels, t_solve, partial_hevp.last['iterations']))
    assert status == 0 and len(lmd) >= 10
    assert np.max(np.abs(lmd[:10] - exact[:10]) / exact[:10]) < 1e-10
    r = A @ x[:, :10] - x[:, :10] * lmd[:10]
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-4 * exact[9]
    assert partial_hevp.last['iterations'] < 100


def test_config1_lap30_device_ilu_matches_reference_iterations(golden_dir):
    """The reference's known answer for lap3d(30, 30, 30), which = 10, T = ILU (MKL dcsrilut + two
    mkl_dcsrtrsv per vector on the host): 27 iterations and ten eigenvalues.  Here the same ILUT runs
    on the host once and the triangular solves on the device: same eigenvalues to 1e-10, the iteration
    count within 20 %."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    from oracle.sparse import lap3d
    k = json.load(open(os.path.join(golden_dir, 'known_answers.json')))['hevp_lap30_ilu10']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    np.random.seed(1)
    T = IncompleteLU(A)
    T.factorize()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    assert status == 0
    assert np.allclose(lmd[:10], k['eigenvalues'], rtol=1e-10)
    assert abs(partial_hevp.last['iterations'] - 27) <= 0.2 * 27
    r = A @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) <= 10 * max(k['residual_norms'])


@pytest.mark.parametrize('key', ['s', 'd', 'c', 'z'])
@pytest.mark.parametrize('m', [1, 5, 16, 70])
def test_triangular_chain_vs_scipy(key, m):
    """Level-scheduled triangular solves of the ILUT factors of a small operator, every type, block
    sizes that fill 1 .. 64 lanes per row and more than 64 pieces, with row permutations."""
    import scipy.sparse.linalg as sla
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.precond import TriangularChain, ilut
    from oracle.sparse import lap3d
    dt = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}[key]
    A = lap3d(9, 8, 7, 1.0, 1.01, 1.02)
    n = A.shape[0]
    if key in 'cz':
        S = sp.diags([np.full(n - 1, 40.0)], [1])
        A = sp.csr_matrix(A.astype(np.complex128) + 1j * S - 1j * S.T)
    lo, up = ilut(A, 1e-8, 9)
    rng = np.random.default_rng(m)
    b = rng.standard_normal((m, n)).astype(dt)
    if key in 'cz':
        b = b + 1j * rng.standard_normal((m, n)).astype(dt)
    pin, pout = rng.permutation(n), rng.permutation(n)
    for perms in ((None, None), (pin, pout)):
        chain = TriangularChain([(lo, True, True), (up, False, False)], dt, *perms)
        assert chain.levels[0] > 5 and chain.levels[1] > 5
        B, X = Vectors(b.copy()), Vectors(n, m, data_type=dt)
        chain.solve(B, X)
        w = b.T.astype(np.complex128 if key in 'cz' else np.float64)
        if perms[0] is not None:
            w = w[pin]
        w = sla.spsolve_triangular(sp.csr_matrix(lo + sp.identity(n)), w, lower=True)
        w = sla.spsolve_triangular(sp.csr_matrix(up), w, lower=False)
        ref = np.zeros_like(w)
        if perms[1] is not None:
            ref[pout] = w
        else:
            ref = w
        assert rel(X.data(), ref.T) < (2e-5 if key in 'sc' else 1e-12)
        chain.solve(B, B)                      # in place
        assert rel(B.data(), ref.T) < (2e-5 if key in 'sc' else 1e-12)


def test_config5_complex_hermitian_block64_shift_invert():
    """BASELINE config 5 on one GPU, at the size a direct factorisation allows in a test (the
    reference factorises with PARDISO on the host; a 126^3 complex 3-D operator has ~10^9 factor
    entries): Hermitian lap3d + i skew, complex128, block of 64 vectors, 20 eigenpairs nearest an
    interior shift by shift-invert, eigenvalues against the closed-form spectrum to 1e-10."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues
    N = 24
    n = N ** 3
    H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n)
    exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
    sigma = 0.5 * (exact[n // 3] + exact[n // 3 + 1])
    opt = Options()
    opt.block_size = 64
    np.random.seed(1)
    lmd, x, status = partial_hevp(H, sigma=sigma, which=20, tol=1e-8, verb=-1, opt=opt)
    assert status == 0 and len(lmd) >= 20
    nearest = exact[np.argsort(np.abs(exact - sigma))[:16]]
    for e in nearest:
        assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e)
    r = H @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-6 * np.max(np.abs(exact))


def test_config5_full_size_inexact_shift_invert():
    """BASELINE config 5 end to end AT ITS SIZE on one GPU: n = 126^3 = 2 000 376, complex128, block of 64, the 20 eigenpairs
    nearest a shift with 40 eigenvalues below it, by inexact shift-invert (block MINRES + Chebyshev polynomial on the device
    blocks; no factorisation exists at this size): eigenvalues against the closed-form spectrum to 1e-10, the Lanczos inertia
    count exact, residuals of the returned pairs small."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues
    N, below = 126, 40
    n = N ** 3
    H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n)
    exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
    sigma = 0.5 * (exact[below - 1] + exact[below])
    opt = Options()
    opt.block_size = 64
    np.random.seed(1)
    sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=16, ratio=250.0)
    lmd, x, status = partial_hevp(H, sigma=sigma, which=20, tol=1e-6, verb=-1, opt=opt, solver=sol)
    assert status == 0 and len(lmd) >= 20
    assert sol.inertia() == (below, n - below)
    for e in exact[np.argsort(np.abs(exact - sigma))[:20]]:
        assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e)
    keep = np.argsort(np.abs(lmd - sigma))[:20]
    r = H @ x[:, keep] - x[:, keep] * lmd[keep]
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-6 * np.max(np.abs(exact))
    assert partial_hevp.last['iterations'] < 20 and sol.iterations < 400


def test_config5_full_size_block64_operations():
    """Config 5 at its full size on one GPU (n = 126^3 = 2 000 376 rows, complex128, m = 64: blocks of
    2.05 GB): the operations of one solver iteration checked through size-independent properties --
    the operator on closed-form eigenvectors, Hermitian Gram, Gram diagonal = dots, linearity of the
    block update."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import hermitian_lap3d_rows, lap3d_coefficients
    N, m, skew = 126, 64, 0.3
    n = N ** 3
    H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n, skew=skew)
    op = CsrOperator(H)
    assert op.layout()[0] == 'wide'
    # planes of 126^2 = 15.5 uniform row blocks: the stacks are cut plane by plane (16 blocks of 992 / 993 rows), 63 pairs of planes
    assert op.stacks()[0] == 63 * 16
    cx, cy, cz = lap3d_coefficients(N, N, N, 1.0, 1.01, 1.02)
    # eigenvectors of the Kronecker sum: x-factor phase^j sin(j k pi / (N + 1)), phase = conj(b) / |b|, b = -cx + i skew
    j = np.arange(1, N + 1)
    b = -cx + 1j * skew
    ph = (np.conj(b) / abs(b)) ** j

    def mode(kx, ky, kz):
        vx = ph * np.sin(j * kx * np.pi / (N + 1))
        vy = np.sin(j * ky * np.pi / (N + 1))
        vz = np.sin(j * kz * np.pi / (N + 1))
        lam = (2 * cx + 2 * abs(b) * np.cos(kx * np.pi / (N + 1)) + 2 * cy - 2 * cy * np.cos(ky * np.pi / (N + 1))
               + 2 * cz - 2 * cz * np.cos(kz * np.pi / (N + 1)))
        return (vz[:, None, None] * vy[None, :, None] * vx[None, None, :]).ravel(), lam
    X, Y = Vectors(n, m, data_type=np.complex128), Vectors(n, m, data_type=np.complex128)
    X.fill_random()
    lams = []
    ks = [(1, 1, 1), (2, 5, 3), (126, 126, 126), (17, 60, 101)]
    for i, k in enumerate(ks):
        v, lam = mode(*k)
        X.select(1, 7 * i)
        X.fill(v.reshape(1, n))
        lams.append(lam)
    X.select(m)
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
    for i, lam in enumerate(lams):
        Y.select(1, 7 * i)
        X.select(1, 7 * i)
        y, x = Y.data()[0], X.data()[0]
        assert np.linalg.norm(y - lam * x) < 1e-12 * abs(lam) * np.linalg.norm(x), (i, lam)
    X.select(m)
    Y.select(m)
    G = X.dot(X)
    assert rel(G, G.conj().T) < 1e-14
    assert rel(np.diag(G), X.dots(X)) < 1e-13
    # self-adjointness of the operator on random blocks: <Y, X> with Y = A X is Hermitian
    XAX = Y.dot(X)
    assert rel(XAX, XAX.conj().T) < 1e-12
    # linearity of the block update against the Gram: (X q)^H X = q^H (X^H X)
    rng = np.random.default_rng(5)
    q = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))) / m
    W = Vectors(n, m, data_type=np.complex128)
    X.multiply(q, W)
    assert rel(X.dot(W), q.conj().T @ G) < 1e-12


def test_config2_pca_20k_full_size():
    """BASELINE config 2: pca() of a dense 20 000 x 20 000 fp32 matrix, 200 components, on one GPU.
    The data are generated from factors (U with a constant first column, so the mean-shifted
    matrix is exactly sum_{k >= 1} s_k u_k v_k^T): singular values within 1e-3 sigma_max of the
    generator's (the reference's svtol class), principal components orthonormal, wall time bounded."""
    from raleigh_amd.interfaces import pca
    M = N = 20000
    r, npc = 400, 200
    rng = np.random.default_rng(1)
    U = rng.standard_normal((M, r)).astype(np.float32)
    U[:, 0] = 1.0
    V = rng.standard_normal((N, r)).astype(np.float32)
    U, _ = np.linalg.qr(U)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(min(M, N)).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    A = np.ascontiguousarray((U * s) @ V.T, dtype=np.float32)
    np.random.seed(1)
    t0 = time.time()
    mean, trans, comps = pca(A, npc=npc)
    el = time.time() - t0
    sv = np.linalg.norm(trans, axis=0)
    print('config 2: pca 20000 x 20000 npc=200 in %.2f s, %d iterations' % (el, pca.last['iterations']))
    assert trans.shape == (M, npc) and comps.shape == (npc, N)
    assert np.max(np.abs(sv - s[1:npc + 1])) < 1e-3 * s[1]
    assert np.max(np.abs(comps @ comps.T - np.eye(npc))) < 1e-3
    assert el < 5.0


def test_pca_singular_values_to_1e5_with_a_tight_tolerance():
    """SURVEY 8(c) states 1e-5 relative for fp32 singular values; the default svtol = 1e-3 (the reference's) stops the
    solver earlier (the configs above assert that class).  With svtol = 1e-9 the same kernels deliver the stated
    accuracy: fp32 data, 50 components of a 6 000 x 4 000 matrix against the generator's spectrum."""
    from raleigh_amd.interfaces import pca
    M, N, r, npc = 6000, 4000, 300, 50
    rng = np.random.default_rng(3)
    U = rng.standard_normal((M, r)).astype(np.float32)
    U[:, 0] = 1.0
    V = rng.standard_normal((N, r)).astype(np.float32)
    U, _ = np.linalg.qr(U)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(N).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    A = np.ascontiguousarray((U * s) @ V.T, dtype=np.float32)
    exact = np.linalg.svd((A - A.mean(axis=0, keepdims=True)).astype(np.float64), compute_uv=False)[:npc]
    np.random.seed(1)
    mean, trans, comps = pca(A, npc=npc, svtol=1e-9)
    sv = np.linalg.norm(trans.astype(np.float64), axis=0)
    assert np.max(np.abs(sv - exact) / exact) < 1e-5
    assert np.max(np.abs(comps @ comps.T - np.eye(npc))) < 2e-5


def test_config4_row_shard_pca_full_size():
    """BASELINE config 4 is pca() of 500 000 x 40 000 fp32 rows over 8 GPUs: every GPU holds a 62 500 x 40 000
    row shard (10 GB).  This is ONE such shard at full size on one GPU, built on the device from factors
    (rows = (U s) V^T by rlh_dense_apply with 62 500 right-hand sides; U's first column constant, so the
    mean-shifted shard is exactly sum_{k >= 1} s_k u_k v_k^T): the two dense products at that shape against
    unit vectors (exact: every sum has one non-zero term) and against each other (<A x, y> = <x, A^T y>),
    then 200 principal components against the generator's singular values."""
    from raleigh_amd.algebra.hip import Vectors, Matrix
    from raleigh_amd.algebra.dense_matrix import AMatrix
    from raleigh_amd.interfaces import pca
    from raleigh_amd import _lib
    M, N, r, npc, m = 62500, 40000, 256, 200, 128
    rng = np.random.default_rng(4)
    U = rng.standard_normal((M, r)).astype(np.float32)
    U[:, 0] = 1.0
    V = rng.standard_normal((N, r)).astype(np.float32)
    U, _ = np.linalg.qr(U)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(N).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    rows = Vectors(N, M, data_type=np.float32)                       # the shard: M vectors of dimension N
    Matrix(np.ascontiguousarray(V)).apply(Vectors(np.ascontiguousarray(U * s)), rows)
    A = AMatrix(rows)
    op = A.as_operator()
    assert op.shape() == (M, N)
    # unit vectors: A^T e_i is row i, A e_j is column j -- to the last bit
    pick_i = np.array([0, 1, 31249, 62499] + list(rng.integers(0, M, m - 4)))
    E = np.zeros((m, M), dtype=np.float32)
    E[np.arange(m), pick_i] = 1.0
    w = Vectors(N, m, data_type=np.float32)
    op.apply(Vectors(E), w, transp=True)
    got = w.data()
    for k in (0, 1, 2, 3, 77):
        assert np.array_equal(got[k], rows.data(int(pick_i[k])))
    pick_j = np.array([0, 39999] + list(rng.integers(0, N, m - 2)))
    E = np.zeros((m, N), dtype=np.float32)
    E[np.arange(m), pick_j] = 1.0
    y = Vectors(M, m, data_type=np.float32)
    op.apply(Vectors(E), y)
    col = y.data()
    for k in (0, 3, 77):
        assert np.array_equal(col[:, int(pick_i[k])], got[k][pick_j])
    # adjointness on random blocks, and the time of the pair
    x = Vectors(N, m, data_type=np.float32)
    x.fill_random()
    z = Vectors(M, m, data_type=np.float32)
    z.fill_random()
    op.apply(x, y)
    op.apply(z, w, transp=True)
    lhs, rhs = z.dot(y), w.dot(x)                # [i, j]: <z_j, A x_i> and <A^T z_j, x_i>
    assert np.linalg.norm(lhs - rhs) <= 2e-4 * np.linalg.norm(lhs)
    _lib.synchronize()
    t0 = time.time()
    for _ in range(3):
        op.apply(x, y)
        op.apply(y, w, transp=True)
    _lib.synchronize()
    pair = (time.time() - t0) / 3
    print('config 4 shard 62500 x 40000 x %d: A x + A^T y in %.2f ms = %.1f TFLOP/s, %.2f TB/s of matrix reads'
          % (m, pair * 1e3, 4.0 * M * N * m / pair / 1e12, 2.0 * M * N * 4 / pair / 1e12))
    np.random.seed(1)
    t0 = time.time()
    mean, trans, comps = pca(A, npc=npc)
    el = time.time() - t0
    sv = np.linalg.norm(trans, axis=0)
    print('config 4 shard: pca npc=%d in %.2f s, %d iterations' % (npc, el, pca.last['iterations']))
    assert trans.shape == (M, npc) and comps.shape == (npc, N)
    assert np.max(np.abs(sv - s[1:npc + 1])) < 1e-3 * s[1]
    assert np.max(np.abs(comps @ comps.T - np.eye(npc))) < 1e-3


def test_triangular_solve_fails_loudly_when_it_gives_up():
    """The persistent triangular solve bounds every spin; a launch that gives up (forced here by a 10-microsecond
    limit, in a process of its own: the error word is sticky) leaves NaNs and the next synchronising call raises."""
    import subprocess
    import sys
    code = r'''
import numpy as np, scipy.sparse as sp
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
from raleigh_amd.algebra.hip.precond import IncompleteLU
from oracle.sparse import lap3d
A = lap3d(40, 40, 40, 1.0, 1.01, 1.02)
T = IncompleteLU(A); T.factorize()
x, y = Vectors(A.shape[0], 8), Vectors(A.shape[0], 8)
x.fill_random()
T.apply(x, y)
try:
    _lib.check(_lib.lib().rlh_sync())
    print('NO ERROR')
except _lib.RlhError as e:
    print('RAISED', 'gave up' in str(e))
'''
    env = dict(os.environ, RLH_SPTRSV_SPIN_TICKS='1000', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert 'RAISED True' in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
