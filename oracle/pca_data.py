"""Synthetic low-rank data of the reference's PCA examples.

TEST INFRASTRUCTURE ONLY.  Restates raleigh/examples/pca/generate_matrix.py:50-77:
A = U diag(s) V^T with orthonormal random U (first column constant when pca=True),
V, and singular values s = scale * t^-alpha for sorted uniform t, normalised to s[0] = 1."""

import numpy as np
import scipy.linalg as sla


def generate(m, n, rank, dtype=np.float32, scale=1.0, alpha=0.75, pca=False):
    k = min(m, n)
    s = np.sort(np.random.rand(k).astype(dtype))
    s = dtype(scale) * s ** (-alpha)
    s = (s / s[0])[:rank]
    u = np.random.randn(m, rank).astype(dtype)
    if pca:
        u[:, 0] = 1.0
    v = np.random.randn(n, rank).astype(dtype)
    u, _ = sla.qr(u, mode='economic')
    v, _ = sla.qr(v, mode='economic')
    a = np.dot(u * s, v.transpose())
    return a, s, u, v
