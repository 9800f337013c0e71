"""GPU tier of the direct shift-invert operator on L D L^H factors (rlh_ldlt_factor on the host, L^-1 / D^-1 / L^-H on the
device through the C ABI): solves against SciPy for every block type, matrices that need 2 x 2 and delayed pivots, the
inertia against LAPACK, and partial_hevp in shift-invert mode against the reference's known answer and a dense solve."""

import json
import os

import numpy as np
import pytest
import scipy.sparse as scs
import scipy.sparse.linalg as sla

pytestmark = pytest.mark.gpu


def _rand_sym(rng, n, dens, cplx=False, shift=0.0):
    m = scs.random(n, n, dens, random_state=rng, format='csr')
    if cplx:
        m = m + 1j * scs.random(n, n, dens, random_state=rng, format='csr')
    m = m + m.conj().T
    return (m + shift * scs.identity(n)).tocsr()


def _saddle(rng, nk, nc):
    K = _rand_sym(rng, nk, 6.0 / nk, shift=4.0)
    B = scs.random(nc, nk, 4.0 / nk, random_state=rng, format='csr') + scs.eye(nc, nk, format='csr')   # (no empty row)
    return scs.bmat([[K, B.T], [B, None]], format='csr')


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.complex128, 1e-10), (np.float32, None), (np.complex64, None)])
def test_solve_with_two_by_two_and_delayed_pivots(dtype, tol):
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    rng = np.random.default_rng(17)
    cplx = np.dtype(dtype).kind == 'c'
    pairs = 2.0 * scs.kron(scs.identity(600), np.array([[0.0, 1.0], [1.0, 0.0]]))     # zero diagonal, no zero row
    for a, sigma in ((_saddle(rng, 1500, 350), 0.0), ((_rand_sym(rng, 1200, 0.004, cplx=cplx) + pairs).tocsr(), 0.0),
                     (_rand_sym(rng, 800, 0.008, cplx=cplx, shift=0.3), 0.11)):
        n = a.shape[0]
        solver = SparseSymmetricSolver(dtype=dtype)
        solver.analyse(a, sigma)
        solver.factorize()
        info = solver.factors().info
        ev = np.linalg.eigvalsh(a.toarray())
        assert solver.inertia() == (int((ev < sigma).sum()), int((ev > sigma).sum()))
        b, x = Vectors(n, 5, data_type=dtype), Vectors(n, 5, data_type=dtype)
        b.fill_random()
        solver.solve(b, x)
        wide = np.complex128 if cplx else np.float64
        shifted = (a - sigma * scs.identity(n)).tocsc().astype(wide)
        bh = b.data().T.astype(wide)
        ref = sla.spsolve(shifted, bh)

        def good(xh):
            if tol is not None:
                return np.max(np.abs(xh - ref)) <= tol * np.max(np.abs(ref))
            # single precision factors (threshold pivoting allows growth): judged by the backward error
            return np.max(np.abs(shifted @ xh - bh)) <= 2e-4 * (np.max(np.abs(bh)) + abs(shifted).sum(axis=1).max() * np.max(np.abs(xh)))
        assert good(x.data().T.astype(wide))
        solver.solve(b, b)                                                # in place
        assert good(b.data().T.astype(wide))
        if sigma == 0.0:
            assert info['two_by_two'] + info['delayed'] > 0               # zero diagonal entries: no 1 x 1 pivot there


def test_block_diagonal_kernel_against_numpy():
    """rlh_bdiag_solve on its own, through the raw ABI: every type, a block whose leading dimension exceeds n, 2 x 2
    pivots at both ends."""
    import ctypes
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.memory import DeviceBuffer
    L = _lib.lib()
    rng = np.random.default_rng(4)
    n, m = 1003, 7
    shift = np.zeros(n, dtype=np.int32)
    i = 0
    while i < n - 1:
        if rng.random() < 0.3 or i == 0 or i == n - 2:
            shift[i], shift[i + 1] = 1, -1
            i += 2
        else:
            i += 1
    for dtype in (np.float32, np.float64, np.complex64, np.complex128):
        coef = rng.standard_normal((n, 2)).astype(dtype)
        if np.dtype(dtype).kind == 'c':
            coef = coef + 1j * rng.standard_normal((n, 2)).astype(dtype)
        x = Vectors(n, m, data_type=dtype)
        x.fill_random()
        old = x.data().copy()
        cb, sb = DeviceBuffer(coef.nbytes, zero=False), DeviceBuffer(shift.nbytes, zero=False)
        _lib.check(L.rlh_h2d(cb.ptr, _lib.host_ptr(coef), coef.nbytes))
        _lib.check(L.rlh_h2d(sb.ptr, _lib.host_ptr(shift), shift.nbytes))
        _lib.check(L.rlh_bdiag_solve(_lib.dtype_code(dtype), n, cb.ptr, sb.ptr, m, x.data_ptr(), x.ld()))
        want = old * coef[:, 0] + old[:, np.arange(n) + shift] * np.where(shift != 0, coef[:, 1], 0)
        eps = 1e-5 if np.dtype(dtype).name in ('float32', 'complex64') else 1e-13      # (two products and a sum: cancellation)
        assert np.allclose(x.data(), want, rtol=eps, atol=eps * np.max(np.abs(want)))
    with pytest.raises(_lib.RlhError):
        _lib.check(L.rlh_bdiag_solve(1, n, None, sb.ptr, m, x.data_ptr(), x.ld()))
    with pytest.raises(_lib.RlhError):
        _lib.check(L.rlh_bdiag_solve(1, n, cb.ptr, sb.ptr, m, x.data_ptr(), n - 1))


def test_shift_invert_known_answer_and_both_factorisations_agree():
    """The reference's answer for lap3d 30^3, six eigenvalues nearest 0 by shift-invert (PARDISO there): the L D L^H
    factors and SuperLU's give the same eigenvalues in the same number of iterations; the L D L^H ones hold half the
    entries (one triangle is factorised)."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    from oracle.sparse import lap3d
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'known_answers.json')) as fh:
        k = json.load(fh)['hevp_lap30_si6']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    seen = {}
    for method in ('ldlt', 'superlu'):
        solver = SparseSymmetricSolver(method=method)
        solver.analyse(A, 0.0)
        solver.factorize()
        assert solver.inertia() == (0, 27000)
        np.random.seed(1)
        lmd, x, status = partial_hevp(solver, which=6, tol=1e-6, verb=-1)
        assert status == 0 and np.allclose(lmd[:6], k['eigenvalues'], rtol=1e-10)
        seen[method] = partial_hevp.last['iterations']
    assert abs(seen['ldlt'] - seen['superlu']) <= 1 and abs(seen['ldlt'] - 17) <= 3


def test_interior_eigenvalues_of_a_saddle_point_matrix():
    """Eigenvalues of [[K, B^T], [B, 0]] nearest 0 on both sides: SuperLU's symmetric mode has no inertia here (row
    interchanges) and partial_hevp used to give up; with 2 x 2 / delayed pivots the count is exact."""
    from raleigh_amd.interfaces import partial_hevp
    rng = np.random.default_rng(23)
    S = _saddle(rng, 900, 200)
    ev = np.linalg.eigvalsh(S.toarray())
    np.random.seed(1)
    lmd, x, status = partial_hevp(S, sigma=0.0, which=(3, 4), tol=1e-9, verb=-1)
    assert status == 0
    want = np.r_[np.sort(ev[ev < 0])[-3:], np.sort(ev[ev > 0])[:4]]
    assert all(np.min(np.abs(lmd - v)) < 1e-8 * max(1.0, abs(v)) for v in want)
    r = S @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-6
