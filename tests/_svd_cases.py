"""Vectors.svd() on blocks that have lost rank -- the situation the solver calls it in (raleigh/core/solver.py:877-885;
the reference's backends use LAPACK / cuSOLVER gesvd there: dense_cblas.py:256-281, dense_cublas.py:537-591) -- shared by
the CPU tier (fake library) and the GPU tier.  Checker: numpy.linalg.svd of the same block in double precision."""

import numpy as np


def block(n, m, dt, cond=None, rank=None, seed=0):
    """n x m block with singular values 1 ... 1/cond (geometric), or exactly `rank` non-zero ones."""
    rng = np.random.default_rng(seed)
    cplx = np.dtype(dt).kind == 'c'
    def rnd(*shape):
        a = rng.standard_normal(shape)
        return a + 1j * rng.standard_normal(shape) if cplx else a
    u, _ = np.linalg.qr(rnd(n, m))
    v, _ = np.linalg.qr(rnd(m, m))
    if rank is not None:
        s = np.zeros(m)
        s[:rank] = np.linspace(1.0, 0.5, rank)
    else:
        s = np.logspace(0, -np.log10(cond), m)
    x = (u * s) @ v.conj().T
    return np.ascontiguousarray(x.T.astype(dt)), s          # vectors as rows, like Vectors(ndarray)


def check(dt, cond=None, rank=None, n=3000, m=24):
    from raleigh_amd.algebra.hip import Vectors
    xt, s_true = block(n, m, dt, cond, rank)
    single = np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64))
    wide = np.complex128 if np.dtype(dt).kind == 'c' else np.float64
    X = Vectors(xt.copy())
    np.random.seed(3)
    sigma, q = X.svd()
    w = X.data().astype(wide).T                               # n x m, should be orthonormal
    exact = np.linalg.svd(xt.astype(wide).T, compute_uv=False)
    eps = np.finfo(np.float32 if single else np.float64).eps
    assert sigma.shape == (m,) and q.shape == (m, m)
    assert np.all(np.diff(sigma) <= 1e-6 * sigma[0])
    # singular values to 1e-10 sigma_max in double (the review's figure), a few hundred eps in single
    assert np.max(np.abs(sigma - exact)) <= (300 * eps if single else 1e-10) * exact[0]
    # orthonormal whatever the rank
    assert np.abs(w.conj().T @ w - np.eye(m)).max() <= (300 * eps if single else 1e-12)
    # reconstruction: X = W diag(sigma) V^H, with the second return value the reference's conj(v) = V
    rec = (w * sigma.astype(np.float64)) @ q.astype(wide).conj().T
    assert np.linalg.norm(rec - xt.astype(wide).T) <= (300 * eps if single else 1e-13) * np.linalg.norm(xt)
    assert np.abs(q.astype(wide).conj().T @ q.astype(wide) - np.eye(m)).max() <= (100 * eps if single else 1e-13)


CASES = [(np.float64, 1e8, None), (np.float64, 1e12, None), (np.float64, None, 9), (np.complex128, 1e10, None),
         (np.complex128, None, 5), (np.float32, 1e4, None), (np.float32, None, 9), (np.complex64, 1e3, None),
         (np.float64, 10.0, None)]
