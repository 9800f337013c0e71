"""CPU tier of the incomplete-LU preconditioner: the library's host factorisation (rlh_ilut_factor
needs no GPU) against the pure-Python restatement of ILUT in oracle/ilut.py, and the host logic of
IncompleteLU / TriangularChain through the C-ABI stand-in."""

import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle.ilut import ilut as ilut_oracle
from oracle.sparse import lap3d
from tests import fake_lib


@pytest.fixture(autouse=True)
def fake():
    from raleigh_amd import _lib
    lib = fake_lib.FakeLib()
    _lib.set_library(lib)
    yield lib
    _lib.set_library(None)


@pytest.mark.parametrize('tol,maxfil', [(1e-6, 6), (1e-3, 2), (0.0, 1000), (1e-6, 0)])
def test_ilut_factor_matches_oracle_real(tol, maxfil):
    from raleigh_amd.algebra.hip.precond import ilut
    A = lap3d(7, 6, 5, 1.0, 1.01, 1.02)
    n = A.shape[0]
    lo, up = ilut(A, tol, maxfil)
    lo_o, up_o = ilut_oracle(A, tol, maxfil)
    assert lo.nnz == lo_o.nnz and up.nnz == up_o.nnz
    assert abs(lo - lo_o).max() <= 1e-15 * abs(lo_o).max() if lo.nnz else True
    assert abs(up - up_o).max() <= 1e-15 * abs(up_o).max()
    assert np.diff(lo.indptr).max() <= maxfil and np.diff(up.indptr).max() <= maxfil + 1
    if tol == 0.0:                  # no dropping: the exact LU
        assert abs((lo + sp.identity(n)) @ up - A).max() < 1e-12 * abs(A).max()


def test_ilut_factor_matches_oracle_complex_and_irregular():
    from raleigh_amd.algebra.hip.precond import ilut
    A = lap3d(6, 5, 4, 1.0, 1.01, 1.02)
    n = A.shape[0]
    S = sp.diags([np.full(n - 1, 0.3)], [1])
    H = sp.csr_matrix(A.astype(np.complex128) + 1j * S - 1j * S.T)
    lo, up = ilut(H, 1e-4, 5)
    lo_o, up_o = ilut_oracle(H, 1e-4, 5)
    assert lo.nnz == lo_o.nnz and up.nnz == up_o.nnz
    assert abs(lo - lo_o).max() < 1e-14 and abs(up - up_o).max() < 1e-12
    rng = np.random.default_rng(3)
    R = sp.random(150, 150, density=0.05, random_state=5, format='csr')
    B = sp.csr_matrix(R + R.T + sp.diags(10 + rng.random(150)))
    lo, up = ilut(B, 1e-3, 8)
    lo_o, up_o = ilut_oracle(B, 1e-3, 8)
    assert abs(lo - lo_o).max() < 1e-14 and abs(up - up_o).max() < 1e-13


def test_ilut_rejects_bad_input():
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip.precond import ilut
    A = sp.lil_matrix(sp.identity(5))
    A[2, 2] = 0.0
    with pytest.raises(_lib.RlhError, match='empty'):
        ilut(sp.csr_matrix(A), 1e-6, 3)


def test_triangular_chain_host_logic():
    import scipy.sparse.linalg as sla
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.precond import TriangularChain, ilut
    A = lap3d(6, 5, 4, 1.0, 1.01, 1.02)
    n = A.shape[0]
    lo, up = ilut(A, 1e-8, 7)
    rng = np.random.default_rng(1)
    b = rng.standard_normal((5, n))
    perm = rng.permutation(n)
    chain = TriangularChain([(lo, True, True), (up, False, False)], np.float64, perm, perm)
    B, X = Vectors(b.copy()), Vectors(n, 5)
    chain.solve(B, X)
    w = sla.spsolve_triangular(sp.csr_matrix(lo + sp.identity(n)), b.T[perm], lower=True)
    w = sla.spsolve_triangular(sp.csr_matrix(up), w, lower=False)
    ref = np.zeros_like(w)
    ref[perm] = w
    assert np.allclose(X.data(), ref.T, rtol=1e-12, atol=1e-14)
    assert chain.algorithmic_bytes(5) == (lo.nnz + up.nnz - n) * 12 + 2 * n * 5 * 8
    with pytest.raises(ValueError):
        chain.solve(B, Vectors(n, 4))
    with pytest.raises(ValueError):
        chain.solve(Vectors(b.astype(np.float32)), Vectors(n, 5, data_type=np.float32))


def test_incomplete_lu_reproduces_the_reference_iteration_count(golden_dir):
    """lap3d(30, 30, 30), which = 10 with the ILUT preconditioner: the reference (MKL dcsrilut +
    mkl_dcsrtrsv) takes 27 iterations (tests/golden/known_answers.json); this repository's ILUT with
    the same tol / maxfil must land within 20 % and give the same eigenvalues."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    k = json.load(open(os.path.join(golden_dir, 'known_answers.json')))['hevp_lap30_ilu10']
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    np.random.seed(1)
    T = IncompleteLU(A)
    T.factorize()                    # tol = 1e-6, max_fill = 1: the reference's defaults
    assert 1.5 < T.fill < 2.5
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    assert status == 0
    assert np.allclose(lmd[:10], k['eigenvalues'], rtol=1e-10)
    assert abs(partial_hevp.last['iterations'] - 27) <= 0.2 * 27
