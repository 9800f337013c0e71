"""Functional NumPy restatement of the RALEIGH Vectors/Matrix operations.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Every block is a 2-D array of shape (nvec, dim): one vector per ROW, exactly
the storage convention of the reference (``raleigh/algebra/dense_ndarray.py:52-83``),
which is bit-identical to a column-major dim x nvec matrix with ld = dim.
All functions are pure: they return new arrays and never alias their inputs.
Citations are relative to the reference tree.
"""

import numpy as np


def _cj(a):
    return a.conj() if np.iscomplexobj(a) else a


def gram(x, y):
    """``x.dot(y)``: G[i, j] = sum_r conj(y[i, r]) * x[j, r], shape (ny, nx).

    Reference: raleigh/algebra/dense_numpy.py:78-82 (NumPy),
    raleigh/algebra/dense_cblas.py:105-124 (gemm ConjTrans, then conj).
    """
    return _cj(y) @ x.T


def dots(x, y):
    """``x.dots(y)``: v[i] = sum_r conj(y[i, r]) * x[i, r].

    Reference: raleigh/algebra/dense_numpy.py:68-76.
    """
    return np.einsum('ir,ir->i', _cj(y), x).astype(x.dtype, copy=False)


def dots_transp(x, y):
    """``x.dots(y, transp=True)``: w[r] = sum_i conj(y[i, r]) * x[i, r].

    Reference: raleigh/algebra/dense_numpy.py:55-66.
    """
    return np.einsum('ir,ir->r', _cj(y), x).astype(x.dtype, copy=False)


def multiply(x, q):
    """``x.multiply(q, out)``: out[j, :] = sum_i q[i, j] * x[i, :].

    Reference: raleigh/algebra/dense_numpy.py:84-93 (``numpy.dot(q.T, x)``).
    """
    return (q.T @ x).astype(x.dtype, copy=False)


def add_q(self_, other, s, q):
    """``self.add(other, s, q)``: self[j, :] += s * sum_i q[i, j] * other[i, :].

    Reference: raleigh/algebra/dense_numpy.py:95-102.
    """
    return (self_ + s * (q.T @ other)).astype(self_.dtype, copy=False)


def axpy(self_, other, s):
    """``self.add(other, s)`` with scalar s: self += s * other.

    Reference: raleigh/algebra/dense_numpy.py:98-99.
    """
    return (self_ + s * other).astype(self_.dtype, copy=False)


def axpy_cols(self_, other, s):
    """``self.add(other, s)`` with 1-D s: self[i, :] += s[i] * other[i, :].

    Reference: raleigh/algebra/dense_numpy.py:103-105.
    """
    s = np.asarray(s)
    return (self_ + s[:, None] * other).astype(self_.dtype, copy=False)


def copy_cols(x_all, ind):
    """``x.copy(y, ind)``: gathers vectors ``ind`` (absolute indices into all of
    x's storage, NOT relative to its selection).

    Reference: raleigh/algebra/dense_numpy.py:35-42.
    """
    return x_all[np.asarray(ind, dtype=np.int64), :].copy()


def scale_cols(x, s, multiply=False):
    """``x.scale(s, multiply)``: x[i] *= s[i], or x[i] /= s[i] skipping s[i]==0.

    Reference: raleigh/algebra/dense_numpy.py:44-52.
    """
    s = np.asarray(s)
    out = x.copy()
    if multiply:
        out *= s[:, None].astype(x.dtype)
    else:
        nz = s != 0
        out[nz] = (out[nz] / s[nz, None]).astype(x.dtype)
    return out


def dense_apply(a, x, transp=False):
    """``Matrix(a).apply(x, y, transp)``: rows of y are A x_i (or A^H x_i).

    Reference: raleigh/algebra/dense_numpy.py:153-175.  For transp the
    reference conjugates x, multiplies by a (un-transposed twice) and
    conjugates y, i.e. y = conj(conj(x) @ a) = x @ conj(a).
    """
    if transp:
        return (x @ _cj(a)).astype(x.dtype, copy=False)
    return (x @ a.T).astype(x.dtype, copy=False)


def orthogonalize(x, other):
    """``x.orthogonalize(other)``: q = gram(x, other); x -= q^T other; returns
    (new_x, q).  Reference: raleigh/algebra/dense_numpy.py:117-123.
    """
    q = gram(x, other)
    return (x - q.T @ other).astype(x.dtype, copy=False), q


def svd(x):
    """``x.svd()``: thin SVD x^T = W^T-ish in the reference's row convention:
    x (m x n) = v @ diag(sigma) @ w with w (m x n) having orthonormal rows;
    w replaces x and (sigma, conj(v)) is returned.

    Reference: raleigh/algebra/dense_numpy.py:125-128.
    """
    v, sigma, w = np.linalg.svd(x, full_matrices=False)
    return w.astype(x.dtype, copy=False), sigma, _cj(v)


def csr_sym_apply(upper_csr, x):
    """``SparseSymmetricMatrix.apply``: y_i = A x_i with A given by its upper
    triangle (symmetric for real, Hermitian for complex data).

    Reference: raleigh/algebra/sparse_mkl.py:18-48 stores triu(A) 1-based and
    calls mkl_?csrmm with matdescra 'SUNF' / 'HUNF'
    (raleigh/algebra/mkl_wrap.py:211-276), i.e. A = U + U^H - diag(U).
    """
    import scipy.sparse as sp
    u = sp.csr_matrix(upper_csr)
    d = sp.diags(u.diagonal())
    full = u + u.conj().T - d
    return np.ascontiguousarray((full @ x.T).T).astype(x.dtype, copy=False)


def uniform_block(seed, n, m, dtype, row0=0, col0=0):
    """The counter-based generator behind rlh_fill_random (not in the reference, which draws
    numpy.random.rand on the host: dense_cublas.py:119-131): element (i, j) of the (m, n) block
    is 2 u - 1, u = the top 53 (float64) or 24 (float32) bits of splitmix64(seed + (col0 + j) * K1
    + (row0 + i + 1) * K2) scaled to [0, 1); bit-exact restatement in uint64 arithmetic."""
    dtype = np.dtype(dtype)
    real = np.dtype({np.complex64: np.float32, np.complex128: np.float64}.get(dtype.type, dtype.type))
    with np.errstate(over='ignore'):
        col = np.uint64(seed) + (np.uint64(col0) + np.arange(m, dtype=np.uint64)) * np.uint64(0x632BE59BD9B4E019)
        row = (np.uint64(row0) + np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = col[:, None] + row[None, :]
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    if real == np.float64:
        u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        out = 2.0 * u - 1.0
    else:
        u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        out = np.float32(2) * u - np.float32(1)
    return out.astype(dtype)


def bf16_round(x):
    """float32 values rounded to the nearest bfloat16 (ties to even), returned as float32 --
    the storage rounding of rlh_bf16_pack / rlh_spmm_cheb_bf16 (not in the reference)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def bf16_bits(x):
    """The 16-bit patterns of bf16_round(x)."""
    return (bf16_round(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def bf16_from_bits(h):
    return (np.asarray(h, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)
