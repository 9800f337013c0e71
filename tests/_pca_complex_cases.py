"""Complex PCA checks shared by the CPU tier (fake library) and the GPU tier:
with the mean shift the transposed product is A^H - conj(a) e^T, so every rank-one term that rides an
A^H product or a dot against the mean takes conj(a).  Checked against NumPy: the operator itself, the
optimal rank-k error (Eckart-Young on the centred data) for samples >= features and the transposed
case, and the row-norm ('m') stopping rule with the shift.

Convention (the reference's: lra.py:436-440 returns right_v.data(), truncated_svd.py:127 returns v.T): the rows
of `comps` are the right singular vectors THEMSELVES, so for complex data A_s ~ trans @ conj(comps).  The
reference holds no complex PCA fixture, hence NumPy as the checker here."""

import numpy as np


def data(m, n, dt, rank=12, seed=3):
    rng = np.random.RandomState(seed)
    rt = np.float32 if dt == np.complex64 else np.float64
    def cplx(*shape):
        return (rng.randn(*shape) + 1j * rng.randn(*shape)).astype(dt)
    s = (2.0 ** -np.arange(rank)).astype(rt)
    A = (cplx(m, rank) * s) @ cplx(rank, n) + 1e-3 * cplx(m, n)
    A += (3.0 + 2.0j) * cplx(1, n)            # a complex mean, far from zero
    return np.ascontiguousarray(A.astype(dt))


def centred(A):
    return A - A.mean(axis=0, keepdims=True)


def operator_against_numpy(m, n, dt):
    """_OperatorSVD.apply = A_s^H A_s x (or A_s A_s^H x) for a complex block."""
    from raleigh_amd.algebra.dense_matrix import AMatrix
    from raleigh_amd.interfaces.pca import _OperatorSVD
    A = data(m, n, dt)
    As = centred(A.astype(np.complex128))
    mat = AMatrix(A, arch='hip')
    op = mat.as_operator()
    transp = m < n
    dim = m if transp else n
    v = op.new_vectors(dim, 3)
    rng = np.random.RandomState(5)
    x = (rng.randn(3, dim) + 1j * rng.randn(3, dim)).astype(dt)
    v.fill(x)
    y = op.new_vectors(dim, 3)
    _OperatorSVD(mat, v, transp, shift=True).apply(v, y)
    xs = x.astype(np.complex128).T
    want = (As @ (As.conj().T @ xs)) if transp else (As.conj().T @ (As @ xs))
    got = y.data().T
    tol = 2e-4 if dt == np.complex64 else 1e-11
    assert np.linalg.norm(got - want) <= tol * np.linalg.norm(want)


def pca_is_optimal(m, n, dt, k=6):
    from raleigh_amd.interfaces.pca import pca
    A = data(m, n, dt)
    mean, trans, comps = pca(A, npc=k)
    assert trans.shape[0] == m and comps.shape[1] == n and comps.shape[0] >= k
    kk = comps.shape[0]
    As = centred(A.astype(np.complex128))
    sg = np.linalg.svd(As, compute_uv=False)
    best = np.sqrt(np.sum(sg[kk:] ** 2)) / np.sqrt(np.sum(sg ** 2))
    ef = np.linalg.norm(As - trans @ comps.conj()) / np.linalg.norm(As)
    single = dt == np.complex64
    assert np.abs(mean.ravel() - A.mean(axis=0)).max() <= (1e-5 if single else 1e-13) * np.abs(A).max()
    assert ef <= best * 1.02 + (2e-6 if single else 1e-12), (ef, best)
    eye = np.eye(kk)
    assert np.abs(comps.conj() @ comps.T - eye).max() <= (2e-5 if single else 1e-9)
    s = np.linalg.norm(trans, axis=0)[:k]          # (sigma carries the solver's default svtol = 1e-3 class)
    assert np.max(np.abs(s - sg[:k])) <= 1e-3 * sg[0], np.max(np.abs(s - sg[:k])) / sg[0]


def row_norm_rule_with_shift(m, n, dt):
    """tol with the 'm' norm on mean-shifted complex data: every row of A_s - L R within tol of the largest row."""
    from raleigh_amd.core.solver import Options
    from raleigh_amd.interfaces.pca import pca
    A = data(m, n, dt)
    opt = Options()
    opt.block_size = 16             # (the default of 128 exceeds what block JCG accepts at this size: dense fall-back)
    mean, trans, comps = pca(A, tol=0.05, norm='m', opt=opt)
    As = centred(A)
    D = As - trans @ comps.conj()
    rows = lambda a: np.sqrt((np.abs(a) ** 2).sum(1))
    assert rows(D).max() <= 0.05 * 1.01 * rows(As).max()
    assert comps.shape[0] < min(m, n)


def update_matches_one_shot(m0, m1, n, dt, k=8):
    """pca(A1, have=pca(A0)) on complex data (missing item 6): the mean of ALL rows, orthonormal
    components, and an error close to the optimal rank-k one of the stacked, centred data."""
    from raleigh_amd.interfaces.pca import pca
    A = data(m0 + m1, n, dt, rank=10, seed=7)
    A0, A1 = A[:m0], A[m0:]
    mean, trans, comps = pca(A0, npc=k)
    mean, trans, comps = pca(A1, have=(mean, trans, comps))
    single = dt == np.complex64
    kk = comps.shape[0]
    assert kk == k and trans.shape == (m0 + m1, kk) and comps.shape == (kk, n)
    assert np.abs(mean.ravel() - A.mean(axis=0)).max() <= (2e-5 if single else 1e-12) * np.abs(A).max()
    assert np.abs(comps.conj() @ comps.T - np.eye(kk)).max() <= (1e-4 if single else 1e-8)
    As = centred(A.astype(np.complex128))
    sg = np.linalg.svd(As, compute_uv=False)
    best = np.sqrt(np.sum(sg[kk:] ** 2)) / np.sqrt(np.sum(sg ** 2))
    ef = np.linalg.norm(As - trans @ comps.conj()) / np.linalg.norm(As)
    assert ef <= 1.25 * best + (1e-5 if single else 1e-10), (ef, best)
    G = trans.conj().T @ trans
    d = np.real(np.diag(G))
    assert np.abs(G - np.diag(d)).max() <= 1e-3 * d[0]
