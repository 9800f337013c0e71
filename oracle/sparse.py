"""Sparse symmetric/Hermitian operator and the Laplacian test matrices.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).
"""

import numpy as np
import scipy.sparse as sp

from . import ops


def lap1d(n, a):
    """1-D Dirichlet Laplacian, n interior points on (0, a): tridiag(-1, 2, -1)/h^2
    with h = a/(n+1).  Same matrix as raleigh/examples/laplace.py:10-13."""
    h = a / (n + 1.0)
    c = 1.0 / (h * h)
    return sp.diags([-c, 2 * c, -c], [-1, 0, 1], shape=(n, n), format='csr')


def lap3d(nx, ny, nz, ax, ay, az):
    """7-point Laplacian, x fastest.  Same matrix as
    raleigh/examples/laplace.py:16-27 (Kronecker sum of three lap1d)."""
    ix, iy, iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    a = (sp.kron(iz, sp.kron(iy, lap1d(nx, ax)))
         + sp.kron(iz, sp.kron(lap1d(ny, ay), ix))
         + sp.kron(lap1d(nz, az), sp.kron(iy, ix)))
    a = sp.csr_matrix(a)
    a.sort_indices()
    return a


def lap3d_eigenvalues(nx, ny, nz, ax, ay, az, k):
    """Analytic k smallest eigenvalues of lap3d:
    sum_d (4/h_d^2) sin^2(j_d pi / (2 (n_d + 1)))."""
    def ev(n, a):
        h = a / (n + 1.0)
        j = np.arange(1, n + 1)
        return 4.0 / (h * h) * np.sin(j * np.pi / (2.0 * (n + 1))) ** 2
    lx, ly, lz = ev(nx, ax), ev(ny, ay), ev(nz, az)
    cut = min(len(lx), k), min(len(ly), k), min(len(lz), k)
    s = (lx[:cut[0], None, None] + ly[None, :cut[1], None]
         + lz[None, None, :cut[2]]).ravel()
    return np.sort(s)[:k]


class SparseSymmetricMatrix:
    """Restates raleigh/algebra/sparse_mkl.py:16-48: keeps triu(A) and applies
    the full symmetric/Hermitian matrix to every vector of a block."""

    def __init__(self, matrix):
        u = sp.triu(matrix, format='csr')
        u.sort_indices()
        self._u = u

    def size(self):
        return self._u.shape[0]

    def data_type(self):
        return self._u.data.dtype

    def csr(self):
        return self._u

    def apply(self, x, y):
        xa = x.data() if hasattr(x, 'data') and callable(x.data) else x
        ya = y.data() if hasattr(y, 'data') and callable(y.data) else y
        ya[...] = ops.csr_sym_apply(self._u, xa)
