"""Synthetic inputs of the BASELINE configurations (raleigh_amd/synthetic.py) and the Matrix-Market
reader: pure host code, no library calls."""

import gzip

import numpy as np
import pytest
import scipy.sparse as sp

from raleigh_amd.synthetic import (fe_surrogate, hermitian_lap3d_rows, hermitian_lap3d_eigenvalues, lap3d_rows,
                                   read_matrix_market)
from oracle.sparse import lap3d


def test_lap3d_rows_match_the_kronecker_form():
    A = lap3d(7, 6, 5, 1.0, 1.01, 1.02)
    assert abs(lap3d_rows(7, 6, 5, 1.0, 1.01, 1.02, 0, 210) - A).max() < 1e-12
    blk = lap3d_rows(7, 6, 5, 1.0, 1.01, 1.02, 50, 133)
    assert blk.shape == (83, 210) and abs(blk - A[50:133]).max() < 1e-12


def test_hermitian_operator_and_its_closed_form_spectrum():
    n = 6 * 5 * 4
    H = hermitian_lap3d_rows(6, 5, 4, 1.0, 1.01, 1.02, 0, n)
    assert abs(H - H.conj().T).max() == 0
    exact = hermitian_lap3d_eigenvalues(6, 5, 4, 1.0, 1.01, 1.02)
    assert np.allclose(np.linalg.eigvalsh(H.toarray()), exact, rtol=1e-13)
    rows = hermitian_lap3d_rows(6, 5, 4, 1.0, 1.01, 1.02, 17, 90)
    assert abs(rows - H[17:90]).max() == 0
    # the global-superdiagonal variant of the CPU-tier config-5 test couples consecutive x-lines too
    G = hermitian_lap3d_rows(6, 5, 4, 1.0, 1.01, 1.02, 0, n, within_lines=False)
    assert G.nnz > H.nnz and abs(G - G.conj().T).max() == 0


def test_fe_surrogate_small_grid_properties():
    A = fe_surrogate(grid=(7, 6, 5), dof=2)
    n = 2 * 7 * 6 * 5
    assert A.shape == (n, n) and abs(A - A.T).max() == 0
    assert np.diff(A.indptr).max() == 58                     # 29 coupled nodes x 2 unknowns
    d = A.diagonal()
    off = np.asarray(abs(A).sum(axis=1)).ravel() - d
    assert np.all(d > off)                                   # strictly diagonally dominant: positive definite
    assert np.all(np.linalg.eigvalsh(A.toarray()) > 0)
    B = fe_surrogate(grid=(7, 6, 5), dof=2)
    assert (A != B).nnz == 0                                 # a pure function of its arguments


def test_matrix_market_reader(tmp_path):
    rng = np.random.default_rng(4)
    R = sp.random(30, 30, density=0.15, random_state=2, format='coo')
    S = sp.coo_matrix(sp.triu(R + R.T + sp.identity(30)))    # stored triangle of a symmetric matrix

    def write(path, header, rows, cols, vals=None, comment=True, opener=open):
        with opener(path, 'wt') as fh:
            fh.write(header + '\n')
            if comment:
                fh.write('% a comment line\n')
            fh.write('%d %d %d\n' % (S.shape[0], S.shape[1], len(rows)))
            for k in range(len(rows)):
                if vals is None:
                    fh.write('%d %d\n' % (rows[k] + 1, cols[k] + 1))
                elif np.iscomplexobj(vals):
                    fh.write('%d %d %.17g %.17g\n' % (rows[k] + 1, cols[k] + 1, vals[k].real, vals[k].imag))
                else:
                    fh.write('%d %d %.17g\n' % (rows[k] + 1, cols[k] + 1, vals[k]))
    full = sp.csr_matrix(S + sp.triu(S, 1).T)
    p = tmp_path / 'sym.mtx'
    write(p, '%%MatrixMarket matrix coordinate real symmetric', S.col, S.row, S.data)   # lower triangle, as SuiteSparse stores it
    A = read_matrix_market(p)
    assert A.shape == (30, 30) and abs(A - full).max() < 1e-15 and A.has_sorted_indices
    pz = tmp_path / 'sym.mtx.gz'
    write(pz, '%%MatrixMarket matrix coordinate real symmetric', S.col, S.row, S.data, opener=gzip.open)
    assert abs(read_matrix_market(pz) - full).max() < 1e-15
    zv = S.data + 1j * np.where(S.row == S.col, 0.0, rng.standard_normal(S.nnz))
    ph = tmp_path / 'herm.mtx'
    write(ph, '%%MatrixMarket matrix coordinate complex hermitian', S.col, S.row, np.conj(zv))
    H = read_matrix_market(ph)
    U = sp.coo_matrix((zv, (S.row, S.col)), shape=S.shape)
    assert abs(H - (U + sp.triu(U, 1).conj().T)).max() < 1e-15 and abs(H - H.conj().T).max() < 1e-15
    pp = tmp_path / 'pat.mtx'
    write(pp, '%%MatrixMarket matrix coordinate pattern general', R.row, R.col, None, comment=False)
    P = read_matrix_market(pp)
    assert P.nnz == sp.csr_matrix((np.ones(R.nnz), (R.row, R.col)), shape=R.shape).nnz and P.max() >= 1.0
    bad = tmp_path / 'bad.mtx'
    bad.write_text('%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n')
    with pytest.raises(ValueError, match='coordinate'):
        read_matrix_market(bad)
