"""The symmetric indefinite factorisation behind the direct shift-invert operator (rlh_ldlt_factor: host-only entry
points of the real library -- no GPU needed), checked against dense LAPACK: P A P^T = L D L^H to rounding, the inertia
against eigvalsh (what PARDISO reports in iparm[21], iparm[22]: mkl_wrap.py:354-489), on the matrices that need 2 x 2 and
delayed pivots (zero diagonal, saddle point), Hermitian ones, singular ones, and through the solver class with the
device calls on the NumPy stand-in."""

import os

import numpy as np
import pytest
import scipy.sparse as scs

from raleigh_amd import _lib
from raleigh_amd.algebra.hip.ldlt import ldlt
from tests import fake_lib


@pytest.fixture(autouse=True)
def fake():
    lib = fake_lib.FakeLib()
    _lib.set_library(lib)
    yield lib
    _lib.set_library(None)


def _rand_sym(rng, n, dens, cplx=False, shift=0.0):
    m = scs.random(n, n, dens, random_state=rng, format='csr')
    if cplx:
        m = m + 1j * scs.random(n, n, dens, random_state=rng, format='csr')
    m = m + m.conj().T
    return (m + shift * scs.identity(n)).tocsr()


def _lap3d(N):
    I = scs.identity(N)
    T = scs.diags([-1, 2, -1], [-1, 0, 1], shape=(N, N))
    return (scs.kron(scs.kron(T, I), I) + scs.kron(scs.kron(I, T), I) + scs.kron(scs.kron(I, I), T)).tocsr()


def _check(a, **kw):
    a = scs.csr_matrix(a)
    n = a.shape[0]
    f = ldlt(a, **kw)
    assert sorted(f.order) == list(range(n))
    low = f.lower
    rows = np.repeat(np.arange(n), np.diff(low.indptr))
    assert np.all(low.indices < rows)                                    # strictly lower, pivot order
    first = np.flatnonzero(f.block == 1)
    assert np.all(f.block[first + 1] == 2) and np.sum(f.block == 2) == len(first) == f.info['two_by_two']
    inside = set(zip(first + 1, first))                                  # no entry inside a 2 x 2 pivot
    assert not inside.intersection(zip(rows.tolist(), low.indices.tolist()))
    L = low + scs.identity(n, format='csr')
    R = L @ f.block_diagonal() @ L.conj().T - a[f.order][:, f.order]
    assert abs(R).max() <= 1e-11 * max(abs(a).max(), 1e-300)
    up = f.lower_transposed()                                             # L^H as the library hands it out: the same entries
    assert up.has_sorted_indices and abs(up - low.conj().T).max() == 0 if n else True
    chk = up.copy()
    chk.has_sorted_indices = False
    chk.sort_indices()
    assert np.array_equal(chk.indices, up.indices)
    ev = np.linalg.eigvalsh(a.toarray())
    assert f.inertia() == (int((ev < 0).sum()), int((ev > 0).sum()))
    assert f.info['perturbed'] == 0
    return f


def test_small_and_structured_cases():
    _check(np.array([[2.0]]))
    _check(np.array([[-3.0]]))
    f = _check(np.array([[0.0, 1.0], [1.0, 0.0]]))
    assert f.info['two_by_two'] == 1
    f = _check(scs.kron(scs.identity(5), np.array([[0.0, 1.0], [1.0, 0.0]])))
    assert f.info['two_by_two'] == 5
    f = _check(scs.kron(scs.identity(4), np.array([[0.0, 1j], [-1j, 0.0]])))          # Hermitian, purely imaginary coupling
    assert f.info['two_by_two'] == 4
    _check(scs.diags([1.0, -2.0, 3.0, -4.0]))
    f = ldlt(scs.csr_matrix((0, 0)))
    assert f.lower.shape == (0, 0) and f.inertia() == (0, 0)


@pytest.mark.parametrize('cplx', [False, True])
def test_random_indefinite(cplx):
    rng = np.random.default_rng(5 + cplx)
    seen_2x2 = seen_delay = 0
    for n, dens, shift in ((50, 0.1, 0.0), (300, 0.02, 0.0), (300, 0.02, 0.3), (200, 0.03, -0.7), (120, 0.3, 0.0)):
        f = _check(_rand_sym(rng, n, dens, cplx, shift))
        seen_2x2 += f.info['two_by_two']
        seen_delay += f.info['delayed']
    assert seen_2x2 > 0 and seen_delay > 0          # the zero-diagonal cases cannot be factorised without either


def test_saddle_point_needs_delayed_pivots():
    """[[K, B^T], [B, 0]]: the constraint rows have no pivot of their own until their K-neighbours are eliminated --
    SuperLU in symmetric mode interchanges rows here (and PARDISO perturbs); the multifrontal factorisation delays."""
    rng = np.random.default_rng(11)
    K = _rand_sym(rng, 200, 0.03, shift=5.0)
    B = scs.random(60, 200, 0.05, random_state=rng, format='csr')
    f = _check(scs.bmat([[K, B.T], [B, None]], format='csr'))
    assert f.inertia()[0] == 60
    # constraints ordered FIRST by a user permutation: every one of them starts with a zero pivot
    perm = np.concatenate([np.arange(200, 260), np.arange(200)])
    f = _check(scs.bmat([[K, B.T], [B, None]], format='csr'), perm=perm)
    assert f.info['two_by_two'] + f.info['delayed'] > 0


def test_shifted_laplacians_and_positive_definite_mode():
    a = _lap3d(10)
    f = _check(a - 1.3 * scs.identity(1000))
    assert f.info['two_by_two'] == 0 or f.info['two_by_two'] > 0        # (either is legitimate: the inertia is what counts)
    f = _check(a, pivot_threshold=0.0)
    assert f.inertia() == (0, 1000) and f.info['two_by_two'] == 0 and f.info['delayed'] == 0
    # the ordering does its work: far fewer entries than the band of the natural order
    assert f.info['nnz_l'] < 0.5 * 1000 * 100


def test_upper_triangle_is_what_is_read_and_duplicates_are_summed():
    rng = np.random.default_rng(3)
    a = _rand_sym(rng, 80, 0.1, shift=0.2)
    junk = scs.tril(scs.random(80, 80, 0.2, random_state=rng), -1)
    f1, f2 = ldlt(a), ldlt(scs.triu(a) + junk)
    assert np.array_equal(f1.order, f2.order) and np.array_equal(f1.lower.toarray(), f2.lower.toarray())
    f4 = ldlt(scs.tril(a))                                               # the lower triangle alone: mirrored
    assert np.array_equal(f4.order, f1.order) and np.allclose(f4.lower.toarray(), f1.lower.toarray(), atol=1e-14)
    up = scs.triu(a, format='coo')
    twice = scs.csr_matrix((np.concatenate([up.data, up.data]) * 0.5, (np.concatenate([up.row, up.row]),
                                                                        np.concatenate([up.col, up.col]))), shape=a.shape)
    f3 = ldlt(twice)
    assert np.allclose(f3.lower.toarray(), f1.lower.toarray(), atol=1e-13)


def test_singular_matrices_are_reported():
    nodes = 50                                                           # graph Laplacian of a path: the last pivot is 0.0
    g = scs.diags([-np.ones(nodes - 1), np.r_[1.0, 2 * np.ones(nodes - 2), 1.0], -np.ones(nodes - 1)], [-1, 0, 1], format='csr')
    f = ldlt(g)
    assert f.info['perturbed'] == 1 and f.inertia()[0] == 0
    f = ldlt(scs.csr_matrix((5, 5)))                                     # the zero matrix
    assert f.info['perturbed'] == 5
    f = ldlt(scs.csr_matrix(np.array([[1.0, 1.0], [1.0, 1.0]])))
    assert f.info['perturbed'] == 1
    # badly scaled but regular: nothing perturbed (pivots are measured against their own columns)
    f = _check(scs.diags([1e20, 1.0, 1e-20, -1e-10]))
    assert f.info['perturbed'] == 0


def test_bad_arguments():
    a = scs.identity(4, format='csr')
    with pytest.raises(_lib.RlhError):
        ldlt(a, perm=[0, 1, 1, 2])
    with pytest.raises(_lib.RlhError):
        ldlt(a, pivot_threshold=0.9)
    with pytest.raises(_lib.RlhError):
        ldlt(scs.csr_matrix(np.array([[1.0, np.inf], [np.inf, 1.0]])))
    with pytest.raises(_lib.RlhError):
        ldlt(scs.csr_matrix(np.array([[1.0, 2.0, 0.0], [2.0, np.nan, 1.0], [0.0, 1.0, 3.0]])))
    with pytest.raises(ValueError):
        ldlt(scs.csr_matrix((3, 4)))


def test_solver_class_on_the_stand_in(fake):
    """SparseSymmetricSolver (default method) end to end with the device calls on the NumPy stand-in: solve() against
    SciPy, real factors on complex blocks and float32 blocks, the pos_def switch, the singular-shift refusal."""
    import scipy.sparse.linalg as sla
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    rng = np.random.default_rng(2)
    K = _rand_sym(rng, 150, 0.04, shift=3.0)
    B = scs.random(30, 150, 0.08, random_state=rng, format='csr')
    S = scs.bmat([[K, B.T], [B, None]], format='csr')
    for dtype, tol in ((np.float64, 1e-10), (np.complex128, 1e-10), (np.float32, 2e-3)):
        solver = SparseSymmetricSolver(dtype=dtype)
        solver.analyse(S, 0.25)
        solver.factorize()
        assert solver.inertia()[0] == int((np.linalg.eigvalsh(S.toarray()) < 0.25).sum())
        b = Vectors(180, 4, data_type=dtype)
        x = Vectors(180, 4, data_type=dtype)
        b.fill_random()
        solver.solve(b, x)
        ref = sla.spsolve((S - 0.25 * scs.identity(180)).tocsc(), b.data().T.astype(np.complex128 if dtype == np.complex128 else np.float64))
        assert np.max(np.abs(x.data().T - ref)) <= tol * np.max(np.abs(ref))
    H = _rand_sym(rng, 90, 0.05, cplx=True, shift=0.4)
    solver = SparseSymmetricSolver(dtype=np.complex128)
    solver.analyse(H, 0.1)
    solver.factorize()
    b, x = Vectors(90, 3, data_type=np.complex128), Vectors(90, 3, data_type=np.complex128)
    b.fill_random()
    solver.solve(b, x)
    ref = sla.spsolve((H - 0.1 * scs.identity(90)).tocsc(), b.data().T)
    assert np.max(np.abs(x.data().T - ref)) <= 1e-10 * np.max(np.abs(ref))
    with pytest.raises(ValueError):                                      # complex factors, real blocks
        solver.solve(Vectors(90, 1, data_type=np.float64), Vectors(90, 1, data_type=np.float64))
    spd = SparseSymmetricSolver(pos_def=True)
    spd.analyse(_lap3d(6), 0.0)
    spd.factorize()
    assert spd.inertia() == (0, 216)
    spd.analyse(_lap3d(6), 1.0)                                          # not positive definite after all
    with pytest.raises(RuntimeError):
        spd.factorize()
    nodes = 50                                                           # graph Laplacian of a path: exactly singular
    g = scs.diags([-np.ones(nodes - 1), np.r_[1.0, 2 * np.ones(nodes - 2), 1.0], -np.ones(nodes - 1)], [-1, 0, 1], format='csr')
    sing = SparseSymmetricSolver()
    sing.analyse(g, 0.0)
    with pytest.raises(RuntimeError, match='singular'):
        sing.factorize()


@pytest.mark.parametrize('isa', ['0', '1'])
def test_front_update_variants_agree(isa):
    """The rank-64 update of a front exists per instruction set (chosen once per process: baseline x86-64, AVX2 + FMA,
    AVX-512); the other two run here in processes of their own, real and complex, against dense LAPACK like the default."""
    import subprocess
    import sys
    code = '''
import numpy as np, scipy.sparse as scs
from raleigh_amd.algebra.hip.ldlt import ldlt
rng = np.random.default_rng(8)
for cplx in (False, True):
    m = scs.random(400, 400, 0.03, random_state=rng, format='csr')
    if cplx:
        m = m + 1j * scs.random(400, 400, 0.03, random_state=rng, format='csr')
    a = (m + m.conj().T + 0.2 * scs.identity(400)).tocsr()
    f = ldlt(a)
    L = f.lower + scs.identity(400, format='csr')
    R = L @ f.block_diagonal() @ L.conj().T - a[f.order][:, f.order]
    assert abs(R).max() <= 1e-11 * abs(a).max(), abs(R).max()
    ev = np.linalg.eigvalsh(a.toarray())
    assert f.inertia() == (int((ev < 0).sum()), int((ev > 0).sum()))
    assert f.info['max_front'] > 100
print('ok')
'''
    env = dict(os.environ, RLH_LDLT_ISA=isa)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]
