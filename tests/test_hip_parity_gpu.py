"""GPU tier: the HIP kernels of librlhip.so against the golden vectors of the
reference and against the CPU oracle on seeded inputs, through the C ABI.
Tolerances: fp64/complex128 1e-13 relative (golden cases) and 1e-12 (long
reductions); fp32/complex64 2e-5."""

import numpy as np
import pytest
import scipy.sparse as sp

import _backend_cases as cases
from oracle import ops
from oracle.sparse import lap3d, lap3d_eigenvalues

pytestmark = pytest.mark.gpu

KEYS = ['s', 'd', 'c', 'z']
DT = cases.DT


@pytest.fixture(scope='module', autouse=True)
def real_library():
    from raleigh_amd import _lib
    _lib.set_library(None)
    L = _lib.lib()                    # raises if the .so or the GPU is missing
    import ctypes
    assert isinstance(L, ctypes.CDLL), 'native library not loaded'
    yield L


def rnd(shape, key, rng):
    a = rng.standard_normal(shape)
    if key in 'cz':
        a = a + 1j * rng.standard_normal(shape)
    return a.astype(DT[key])


def tol_for(key, n):
    return (3e-5 if key in 'sc' else 1e-12)


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('shape', [(5, 257), (16, 192)])
def test_ops_golden(golden_dir, key, shape):
    cases.ops_case(golden_dir, key, shape)


@pytest.mark.parametrize('key', KEYS)
def test_matrix_golden(golden_dir, key):
    cases.matrix_case(golden_dir, key)


def test_sparse_golden(golden_dir):
    cases.sparse_case(golden_dir)


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('n,mx,my', [(1, 1, 1), (63, 3, 2), (1000, 8, 8), (4097, 16, 5), (20011, 32, 32),
                                     (20000, 33, 7), (9999, 70, 40), (5000, 130, 3), (7777, 64, 64), (3001, 100, 129)])
def test_gram_vs_oracle(key, n, mx, my):
    """Panels of 1/2/4 tiles, 128 x 128 panels with one part per wave (windows of more than 64 real
    columns), ragged tails, odd n (unaligned columns when ld is not padded)."""
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(n + mx)
    x, y = rnd((mx, n), key, rng), rnd((my, n), key, rng)
    X, Y = Vectors(x), Vectors(y)
    g = X.dot(Y)
    ref = ops.gram(x.astype(np.complex128 if key in 'cz' else np.float64),
                   y.astype(np.complex128 if key in 'cz' else np.float64))
    assert g.shape == (my, mx)
    assert cases.rel(g, ref) < tol_for(key, n)
    gs = X.dot(X)
    xs = x.astype(np.complex128 if key in 'cz' else np.float64)
    assert cases.rel(gs, ops.gram(xs, xs)) < tol_for(key, n)
    d = X.dots(X)
    assert cases.rel(d, ops.dots(xs, xs)) < tol_for(key, n)
    # bitwise reproducibility of the deterministic reduction
    assert np.array_equal(g, X.dot(Y))


@pytest.mark.parametrize('n,mx,my', [(2048, 64, 64), (7777, 64, 64), (5003, 40, 50), (100001, 33, 64), (40000, 64, 37)])
def test_gram_complex128_interleaved_view(n, mx, my, monkeypatch):
    """gram_z_dma_kernel: the two-operand Gram of complex128 blocks of 33 .. 64 vectors as a real Gram of Y_r against
    [X_r | J X_r] on LDS-DMA staging -- ragged widths (clamped columns), row counts that leave a partial chunk (zeroed
    rows), windows inside wider blocks (other leading dimensions); against the oracle, against the workgroup kernel it
    replaces (RLH_GRAM_ZDMA is read once per process: the other kernel is reached through a self-Gram-shaped request it
    does not take), reproducible bit for bit."""
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(n + mx)
    x, y = rnd((mx, n), 'z', rng), rnd((my, n), 'z', rng)
    X, Y = Vectors(x), Vectors(y)
    g = X.dot(Y)
    ref = ops.gram(x, y)
    assert g.shape == (my, mx)
    assert cases.rel(g, ref) < 1e-13 * max(1.0, np.sqrt(n) / 30)
    assert np.array_equal(g, X.dot(Y))
    # conj symmetry with the operands swapped (another launch of the same kernel)
    assert cases.rel(Y.dot(X), ref.conj().T) < 1e-13 * max(1.0, np.sqrt(n) / 30)
    # windows of wider blocks
    wide = Vectors(n, mx + 7, data_type=np.complex128)
    wide.select(mx, 5)
    wide.fill(x)
    assert np.array_equal(wide.dot(Y), g)


@pytest.mark.parametrize('key', ['d', 's'])
@pytest.mark.parametrize('n,mx,my', [(31, 32, 32), (64, 17, 32), (30001, 24, 17), (15000, 32, 64), (12345, 20, 50),
                                     (100003, 32, 48), (4096, 32, 33), (70001, 18, 18), (20001, 64, 64), (9000, 40, 20),
                                     (33333, 50, 64), (5000, 9, 3), (777, 12, 12)])
def test_gram_streaming_kernel_shapes(key, n, mx, my, monkeypatch):
    """The wave-private streaming Gram (real types, <= 32 columns on the right, <= 64 on the left): fewer rows
    than one tile, ragged widths (clamped columns), the 64-column left window, row counts that leave a partial
    tile; plain, self and stacked (windows made of blocks with DIFFERENT leading dimensions) requests."""
    from raleigh_amd.algebra.hip import Vectors
    monkeypatch.setenv('RLH_GRAM_STREAM', '1')         # (the default)
    rng = np.random.default_rng(n + 3 * mx + my)
    x, y = rnd((mx, n), key, rng), rnd((my, n), key, rng)
    X, Y = Vectors(x), Vectors(y)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    tol = tol_for(key, n)
    g = X.dot(Y)
    assert g.shape == (my, mx)
    assert cases.rel(g, ops.gram(x64, y64)) < tol
    assert cases.rel(X.dot(X), ops.gram(x64, x64)) < tol
    assert np.array_equal(g, X.dot(Y))                 # fixed summation order
    # stacked: [Y1 | Y2]^H X with Y2 a window of a wider block (another leading dimension / offset)
    m1 = my // 2
    wide = Vectors(n, my + 5, data_type=DT[key])
    wide.select(my - m1, 3)
    wide.fill(y[m1:])
    Y1 = Vectors(y[:m1].copy())
    rb = X.reduction_batch()
    rb.gram([X], [Y1, wide])
    rb.dots(X, X)
    got, dd = rb.run()
    assert got.shape == (my, mx)
    assert cases.rel(got, ops.gram(x64, y64)) < tol
    assert cases.rel(dd, ops.dots(x64, x64)) < tol


@pytest.mark.parametrize('key', ['d', 's'])
@pytest.mark.parametrize('n,m', [(20011, 32), (4099, 24), (1000, 17)])
def test_gram_streaming_kernel_shared_block(key, n, m, monkeypatch):
    """Stacked requests whose right block is also one of the left blocks ([X | Y]^H Y, [AX | X]^H X): the streaming
    kernel stages that block once and reads both fragment sides out of the same image."""
    from raleigh_amd.algebra.hip import Vectors
    monkeypatch.setenv('RLH_GRAM_STREAM', '1')
    rng = np.random.default_rng(n + m)
    x, y = rnd((m, n), key, rng), rnd((32, n), key, rng)
    X, Y = Vectors(x), Vectors(y)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    tol = tol_for(key, n)
    for rights, lefts, r64, l64 in (([X], [Y, X], x64, np.concatenate((y64, x64))),       # the shared block second
                                    ([Y], [Y, X], y64, np.concatenate((y64, x64)))):      # ... and first (offset 0)
        rb = X.reduction_batch()
        rb.gram(rights, lefts)
        got, = rb.run()
        assert got.shape == (l64.shape[0], r64.shape[0])
        assert cases.rel(got, ops.gram(r64, l64)) < tol


@pytest.mark.parametrize('key', KEYS)
def test_unaligned_leading_dimension(key):
    """A Vectors view of a C-ordered Matrix has ld = padded row length but a window of a
    matrix with odd row count exercises the scalar-load paths through raw ABI calls."""
    import ctypes
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip.memory import DeviceBuffer
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n, m, ld = 1001, 6, 1003            # ld*es not a multiple of 16 for every dtype but z
    x, y = rnd((m, ld), key, rng), rnd((m, ld), key, rng)
    es = x.itemsize
    bx, by = DeviceBuffer(x.nbytes + 16), DeviceBuffer(y.nbytes + 16)
    off = 4 if key == 's' else 8        # mis-align the base as well
    _lib.check(L.rlh_h2d(bx.ptr + off, _lib.host_ptr(x), x.nbytes))
    _lib.check(L.rlh_h2d(by.ptr + off, _lib.host_ptr(y), y.nbytes))
    code = _lib.dtype_code(DT[key])
    g = np.zeros((m, m), dtype=DT[key])
    _lib.check(L.rlh_gram(code, n, m, bx.ptr + off, ld, m, by.ptr + off, ld, None, _lib.host_ptr(g)))
    assert cases.rel(g, ops.gram(x[:, :n], y[:, :n])) < tol_for(key, n)
    d = np.zeros((m,), dtype=DT[key])
    _lib.check(L.rlh_dots(code, n, m, bx.ptr + off, ld, by.ptr + off, ld, None, _lib.host_ptr(d)))
    assert cases.rel(d, ops.dots(x[:, :n], y[:, :n])) < tol_for(key, n)
    s = rnd((m,), key, rng)
    _lib.check(L.rlh_axpy_cols(code, n, m, _lib.host_ptr(s), bx.ptr + off, ld, by.ptr + off, ld))
    out = np.zeros_like(y)
    _lib.check(L.rlh_d2h(_lib.host_ptr(out), by.ptr + off, y.nbytes))
    assert cases.rel(out[:, :n], ops.axpy_cols(y[:, :n], x[:, :n], s)) < tol_for(key, n)
    assert np.array_equal(out[:, n:], y[:, n:])        # padding untouched
    _lib.check(L.rlh_copy(code, n, m, bx.ptr + off, ld, by.ptr + off, ld))
    _lib.check(L.rlh_d2h(_lib.host_ptr(out), by.ptr + off, y.nbytes))
    assert np.array_equal(out[:, :n], x[:, :n]) and np.array_equal(out[:, n:], y[:, n:])


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('n,k,m', [(1, 1, 1), (777, 5, 3), (10000, 16, 16), (30011, 32, 32), (4096, 40, 9),
                                   (3000, 3, 70), (2048, 300, 20), (5, 8, 8), (33, 12, 20), (1023, 64, 64),
                                   (9001, 9, 64), (20002, 128, 17), (66, 37, 33)])
def test_block_update_vs_oracle(key, n, k, m):
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(n + k + m)
    x, w, q = rnd((k, n), key, rng), rnd((m, n), key, rng), rnd((k, m), key, rng)
    X = Vectors(x)
    W = Vectors(w.copy())
    X.multiply(q, W)
    big = np.complex128 if key in 'cz' else np.float64
    ref = ops.multiply(x.astype(big), q.astype(big))
    scale = np.linalg.norm(ref)
    assert np.linalg.norm(W.data() - ref) / scale < tol_for(key, n) * 10
    W = Vectors(w.copy())
    alpha = -0.5
    W.add(X, alpha, q.T.copy().T)        # F-ordered q
    ref = ops.add_q(w.astype(big), x.astype(big), alpha, q.astype(big))
    assert np.linalg.norm(W.data() - ref) / np.linalg.norm(ref) < tol_for(key, n) * 10


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('M,N,m', [(300, 200, 7), (1000, 513, 33), (257, 1025, 128), (2000, 100, 70), (3001, 2200, 96)])
def test_dense_apply_vs_oracle(key, M, N, m):
    """Y = A X and Y = A^H X for row- and column-major A, every type: float on the 32x32x2 matrix-core kernels, double
    and the complex types on dense_mfma16_kernel (real planes on the 16x16x4 shapes; conj(A) by the sign of the
    im plane), ragged tile edges in all three dimensions."""
    from raleigh_amd.algebra.hip import Vectors, Matrix
    rng = np.random.default_rng(M + N)
    a, x, z = rnd((M, N), key, rng), rnd((m, N), key, rng), rnd((m, M), key, rng)
    tol = 2e-4 if key in 'sc' else 1e-12
    big = np.complex128 if key in 'cz' else np.float64
    for arr in (np.ascontiguousarray(a), np.asfortranarray(a)):
        A = Matrix(arr)
        y = Vectors(M, m, data_type=DT[key])
        A.apply(Vectors(x), y)
        assert cases.rel(y.data(), ops.dense_apply(a.astype(big), x.astype(big))) < tol
        w = Vectors(N, m, data_type=DT[key])
        A.apply(Vectors(z), w, transp=True)
        assert cases.rel(w.data(), ops.dense_apply(a.astype(big), z.astype(big), True)) < tol


def test_dense_apply_valu_fallback_agrees(monkeypatch):
    """The VALU kernel kept for unaligned layouts (RLH_DENSE_VALU=1 forces it) against the matrix-core kernel."""
    from raleigh_amd.algebra.hip import Vectors, Matrix
    rng = np.random.default_rng(12)
    a, x = rnd((700, 450), 'z', rng), rnd((40, 450), 'z', rng)
    A, X = Matrix(a), Vectors(x)
    y1, y2 = Vectors(700, 40, data_type=np.complex128), Vectors(700, 40, data_type=np.complex128)
    A.apply(X, y1)
    monkeypatch.setenv('RLH_DENSE_VALU', '1')
    A.apply(X, y2)
    assert cases.rel(y1.data(), y2.data()) < 1e-13


@pytest.mark.parametrize('m', [1, 5, 16, 32, 40])
def test_spmm_lap3d_vs_oracle(m):
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    A = lap3d(23, 19, 17, 1.0, 1.01, 1.02)
    n = A.shape[0]
    rng = np.random.default_rng(m)
    x = rng.standard_normal((m, n))
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, m)
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < 1e-13


def test_spmm_irregular_rows():
    """Ragged CSR: empty rows, one dense row, random pattern (FE-like irregularity)."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    rng = np.random.default_rng(11)
    n = 3001
    R = sp.random(n, n, density=0.004, random_state=3, format='lil')
    R[5, :] = rng.standard_normal(n)          # a full row
    R[100:164, :] = 0                          # a slice of empty rows
    U = sp.triu(sp.csr_matrix(R), format='csr')
    op = SparseSymmetricMatrix(U)
    for key in ('d', 'z'):
        if key == 'z':
            U = sp.csr_matrix(U.astype(np.complex128) + 1j * sp.triu(U, k=1))
            op = SparseSymmetricMatrix(U)
        x = rnd((9, n), key, rng)
        X, Y = Vectors(x), Vectors(n, 9, data_type=DT[key])
        op.apply(X, Y)
        assert cases.rel(Y.data(), ops.csr_sym_apply(U, x)) < 1e-12


def test_full_size_properties_fp64():
    """BASELINE roofline point (n = 10^7 scaled to 2*10^6 rows here to keep the host
    side light; the kernels take the same paths): size-independent properties.
    - Gram of known vectors: exact integer sums;
    - linearity: (X Q)^H X == Q^H (X^H X);
    - SpMM on an analytic eigenvector of the 7-point Laplacian reproduces lambda * v."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    N = 126
    A = lap3d(N, N, N, 1.0, 1.01, 1.02)
    n = A.shape[0]
    m = 32
    X = Vectors(n, m)
    # exact-integer known answer: column j holds the constant (j+1) on even rows, 0 on odd rows
    x = np.zeros((m, n))
    x[:, ::2] = np.arange(1, m + 1)[:, None]
    X.fill(x)
    G = X.dot(X)
    expect = np.outer(np.arange(1, m + 1), np.arange(1, m + 1)) * float((n + 1) // 2)
    assert np.array_equal(G, expect)
    # linearity with random data
    np.random.seed(1)
    X.fill_random()
    Q = np.random.randn(m, m)
    W = Vectors(n, m)
    X.multiply(Q, W)
    G = X.dot(X)
    lhs = X.dot(W)              # lhs[i, j] = <w_i, x_j> = sum_k Q[k, i] G[k, j]
    assert cases.rel(lhs, Q.T @ G) < 1e-12
    assert cases.rel(G, G.T) < 1e-14
    # SpMM on analytic eigenvectors
    op = SparseSymmetricMatrix(A)
    idx = np.arange(1, N + 1)
    def mode(kx, ky, kz):
        sx, sy, sz = (np.sin(k * np.pi * idx / (N + 1)) for k in (kx, ky, kz))
        return (sz[:, None, None] * sy[None, :, None] * sx[None, None, :]).ravel()
    def lam(kx, ky, kz):
        return sum(4.0 / (a / (N + 1)) ** 2 * np.sin(k * np.pi / (2 * (N + 1))) ** 2
                   for k, a in ((kx, 1.0), (ky, 1.01), (kz, 1.02)))
    modes = [(1, 1, 1), (2, 1, 1), (1, 3, 2), (5, 4, 7)]
    V = Vectors(np.array([mode(*k) for k in modes]))
    AV = Vectors(n, len(modes))
    op.apply(V, AV)
    lams = np.array([lam(*k) for k in modes])
    AV.add(V, -lams)
    res = np.sqrt(np.abs(AV.dots(AV))) / (lams * np.sqrt(np.abs(V.dots(V))))
    assert np.max(res) < 1e-12
    assert abs(lams[0] - lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02, 1)[0]) < 1e-9 * lams[0]


def test_empty_and_degenerate():
    from raleigh_amd.algebra.hip import Vectors
    e = Vectors(100, data_type=np.float64)
    x = Vectors(np.ones((3, 100)))
    assert e.nvec() == 0
    assert x.dot(e).shape == (0, 3) and e.dot(x).shape == (3, 0)
    assert e.dots(e).shape == (0,)
    x.select(0)
    x.scale(np.ones(3))
    x.zero()
    x.select(3)
    assert np.all(x.data() == 1)
    z = Vectors(0, 2, data_type=np.float32)
    assert z.dot(z).shape == (2, 2) and np.all(z.dot(z) == 0)


def test_more_than_2_31_elements_in_a_block():
    """The reference passes sizes as c_int and overflows at n*m >= 2^31 (cuda_wrap.py:143,
    dense_cublas.py:142); here every size is 64-bit.  fp32 block of 67 200 000 x 33 = 2.2e9
    elements (8.9 GB): exact integer sums located past the 2^31-element mark."""
    from raleigh_amd.algebra.hip import Vectors
    n, m = 67_200_000, 33
    assert n * m > 2 ** 31
    X = Vectors(n, m, data_type=np.float32)              # zero-filled on the device
    col = np.zeros((1, n), dtype=np.float32)
    col[0, -1000:] = 1.0
    col[0, :7] = 2.0
    for j in (0, 32):                                     # first and last vector
        X.select(1, j)
        X.fill(col)
    X.select(m)
    d = X.dots(X)
    expect = np.zeros(m, dtype=np.float32)
    expect[[0, 32]] = 1000.0 + 7 * 4.0
    assert np.array_equal(d, expect)
    g = X.dot(X)
    assert g[32, 0] == 1028.0 and g[0, 32] == 1028.0 and g[32, 32] == 1028.0 and g[1, 1] == 0.0
    W = Vectors(n, 2, data_type=np.float32)
    q = np.zeros((m, 2), dtype=np.float32)
    q[32, 0], q[0, 1], q[32, 1] = 3.0, 1.0, 1.0
    X.multiply(q, W)                                      # W0 = 3 x32, W1 = x0 + x32
    assert np.array_equal(W.dots(W), np.array([9 * 1028.0, 4 * 1028.0], dtype=np.float32))
    X.select(1, 32)
    W.select(1, 0)
    X.copy(W)
    W.add(X, -1.0)
    assert W.dots(W)[0] == 0.0


@pytest.mark.parametrize('key', ['d', 'z'])
def test_fused_chebyshev_step(key):
    """rlh_spmm_cheb: p = cy y + cp p + cb (b - A y) in one pass, p in place, against the oracle."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    A = lap3d(23, 19, 17, 1.0, 1.01, 1.02)
    if key == 'z':
        S = sp.diags([np.full(A.shape[0] - 1, 0.3)], [1], shape=A.shape)
        A = sp.csr_matrix(A.astype(np.complex128) + 1j * S - 1j * S.T)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    m = 21
    y0, p0, b0 = (rnd((m, n), key, rng) for _ in range(3))
    op = SparseSymmetricMatrix(A)
    y, p, b = Vectors(y0.copy()), Vectors(p0.copy()), Vectors(b0.copy())
    op.cheb_step(y, p, b, 1.3, -0.3, -1.7)
    t = ops.csr_sym_apply(sp.triu(A, format='csr'), y0)
    assert cases.rel(p.data(), 1.3 * y0 - 0.3 * p0 - 1.7 * (b0 - t)) < 1e-13
    assert np.array_equal(y.data(), y0) and np.array_equal(b.data(), b0)


@pytest.mark.parametrize('src,dst', [('d', 's'), ('s', 'd'), ('z', 'c'), ('c', 'z'), ('d', 'd')])
def test_convert_precision(src, dst):
    """rlh_convert / Vectors.convert_to: element-wise rounding identical to NumPy's astype, on a
    column window with different leading dimensions on the two sides."""
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(17)
    n, m = 100003, 9
    x0 = rnd((m, n), src, rng)
    X = Vectors(x0.copy())
    Y = Vectors(n, m + 2, data_type=DT[dst])
    Y.fill(np.repeat(np.arange(1, m + 3)[:, None], n, axis=1).astype(DT[dst]))
    X.select(5, 2)
    Y.select(5, 4)
    X.convert_to(Y)
    Y.select(m + 2, 0)
    y = Y.data()
    assert np.array_equal(y[4:9], x0[2:7].astype(DT[dst]))
    assert np.all(y[:4] == np.arange(1, 5)[:, None]) and np.all(y[9:] == np.arange(10, m + 3)[:, None])


# ---------------------------------------------------------------- the three device layouts of the sparse operator
@pytest.fixture(params=['sell', 'well', 'wide'])
def spmm_format(request, monkeypatch):
    """RLH_SPMM_FORMAT is read by rlh_csr_create per handle: 'sell' = sliced ELL (per-entry
    gathers), 'well' = 1024-row windowed ELL (column windows staged through the LDS, rows of at
    most 8 entries of a real type) even where the locality test would not choose it, 'wide' = the
    256-row interleaved windowed layout (any row length, any type); a matrix 'well' cannot hold
    falls to 'wide'."""
    monkeypatch.setenv('RLH_SPMM_FORMAT', request.param)
    return request.param


def expected_layout(fmt, key, max_row=7):
    if fmt == 'well' and (key in 'cz' or max_row > 8):
        return 'wide'
    return fmt


def _sym(A, key):
    """A symmetric real matrix made Hermitian with a skew imaginary part for the complex types."""
    A = sp.csr_matrix(A)
    if key in 'cz':
        S = sp.triu(A, k=1)
        A = A + 0.5j * S - 0.5j * S.T
    return sp.csr_matrix(A.astype(DT[key]))


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('m', [1, 3, 8, 21, 37])
def test_spmm_layouts_stencil(spmm_format, key, m):
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    A = _sym(lap3d(23, 19, 17, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    rng = np.random.default_rng(100 + m)
    x = rnd((m, n), key, rng)
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, m, data_type=DT[key])
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < (2e-6 if key in 'sc' else 1e-13)


@pytest.mark.parametrize('key', ['s', 'd', 'z'])
def test_spmm_layouts_irregular(spmm_format, key):
    """Ragged rows up to 30 entries, empty rows, far-away couplings (several windows per block),
    a last block of 3 rows."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    rng = np.random.default_rng(5)
    n = 5 * 1024 + 3
    band = sp.random(n, n, density=0.0, format='lil')
    rows = rng.integers(0, n, 9000)
    cols = np.clip(rows + rng.integers(-40, 41, 9000), 0, n - 1)
    far = rng.integers(0, n, 1500)
    B = sp.coo_matrix((rng.standard_normal(9000), (rows, cols)), shape=(n, n)).tocsr()
    Fm = sp.coo_matrix((rng.standard_normal(1500), (far, (far + 2500) % n)), shape=(n, n)).tocsr()
    A = sp.lil_matrix(B + B.T + Fm + Fm.T + sp.eye(n))
    A[200:300, :] = 0
    A[:, 200:300] = 0
    A[7, 7:33] = 1.5
    A[7:33, 7] = 1.5
    A = _sym(sp.csr_matrix(A), key)
    A.eliminate_zeros()
    assert np.diff(A.indptr).max() <= 32 and np.diff(A.indptr).min() == 0
    x = rnd((11, n), key, rng)
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, 11, data_type=DT[key])
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < (2e-6 if key == 's' else 1e-13)


@pytest.mark.parametrize('key', ['s', 'd'])
def test_spmm_layouts_no_locality(spmm_format, key):
    """Uniformly random pattern: the windowed layout degenerates to one wide window (forced) or
    is not chosen; both give the oracle's product."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    rng = np.random.default_rng(8)
    n = 4000
    R = sp.random(n, n, density=0.001, random_state=4, format='csr')
    A = _sym(R + R.T + sp.eye(n), key)
    x = rnd((6, n), key, rng)
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, 6, data_type=DT[key])
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < (2e-6 if key == 's' else 1e-13)


@pytest.mark.parametrize('rows', [(1100, 2300), (1101, 2302)])
@pytest.mark.parametrize('key', ['d', 'c'])
def test_spmm_layouts_halo_block(spmm_format, key, rows):
    """Row shard of an operator: columns [0, n_own) come from X, the rest from the halo block H
    (rlh_spmm's n_own / H arguments), windows straddling the two included; plain and fused
    Chebyshev forms."""
    import ctypes
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    A = _sym(lap3d(16, 15, 14, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    r0, r1 = rows                             # the shard's rows (1201 of them: the own / halo boundary
                                              # is then not a multiple of the 16-byte staging piece)
    loc = A[r0:r1]
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    # local numbering: own columns first (in order), then the halo columns in ascending global order
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(r1 - r0)
    newcol[halo] = (r1 - r0) + np.arange(len(halo))
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr),
                      shape=(r1 - r0, r1 - r0 + len(halo)))
    L.sort_indices()
    op = CsrOperator(L, n_own=r1 - r0)
    rng = np.random.default_rng(2)
    m = 13
    x = rnd((m, n), key, rng)
    X, Hb = Vectors(np.ascontiguousarray(x[:, r0:r1])), Vectors(np.ascontiguousarray(x[:, halo]))
    Y = Vectors(r1 - r0, m, data_type=DT[key])
    assert op.layout()[0] == expected_layout(spmm_format, key)
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hb.data_ptr(), Hb.ld())
    ref = (A @ x.T).T[:, r0:r1]
    tol = 2e-6 if key == 'c' else 1e-13
    assert cases.rel(Y.data(), ref) < tol
    p_0, b_0 = rnd((m, r1 - r0), key, rng), rnd((m, r1 - r0), key, rng)
    P, B = Vectors(p_0.copy()), Vectors(b_0.copy())
    op.cheb_step_ptr(m, X, P, B, 1.7, -0.7, -0.2, Hb.data_ptr(), Hb.ld())
    assert cases.rel(P.data(), 1.7 * x[:, r0:r1] - 0.7 * p_0 - 0.2 * (b_0 - ref)) < tol


@pytest.mark.parametrize('key', ['s', 'd'])
def test_fused_chebyshev_step_layouts(spmm_format, key):
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    A = _sym(lap3d(23, 19, 17, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    m = 10
    y0, p0, b0 = (rnd((m, n), key, rng) for _ in range(3))
    op = SparseSymmetricMatrix(A)
    y, p, b = Vectors(y0.copy()), Vectors(p0.copy()), Vectors(b0.copy())
    op.cheb_step(y, p, b, 1.3, -0.3, -1.7)
    t = ops.csr_sym_apply(sp.triu(A, format='csr'), y0)
    assert cases.rel(p.data(), 1.3 * y0 - 0.3 * p0 - 1.7 * (b0 - t)) < (3e-6 if key == 's' else 1e-13)
    assert np.array_equal(y.data(), y0)


def test_layout_choice(monkeypatch):
    """The layout choice: the stencil gets the 1024-row windowed layout, a matrix with a long row
    the interleaved one, a random pattern the sliced one; the environment override wins where the
    layout can hold the matrix."""
    from raleigh_amd.algebra.hip import CsrOperator
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    A = lap3d(23, 19, 17, 1.0, 1.01, 1.02)
    lay, stored, ratio = CsrOperator(A).layout()
    assert lay == 'well' and stored == 8 * 1024 * 8 and 0 < ratio < 0.9      # 8 blocks, slots padded to 8
    rng = np.random.default_rng(4)
    nr = 200000                                   # random couplings: ~12 per row, no column locality
    i, j = rng.integers(0, nr, 6 * nr), rng.integers(0, nr, 6 * nr)
    R = sp.coo_matrix((rng.standard_normal(6 * nr), (i, j)), shape=(nr, nr)).tocsr()
    lay, stored, ratio = CsrOperator(R + R.T + sp.eye(nr)).layout()
    assert lay == 'sell' and ratio > 0.9
    D = sp.lil_matrix(A)
    D[5, :40] = 1.0
    assert CsrOperator(sp.csr_matrix(D)).layout()[0] == 'wide'        # a 40-entry row: interleaved layout
    assert CsrOperator(A.astype(np.complex128)).layout()[0] == 'wide'  # complex operators too
    monkeypatch.setenv('RLH_SPMM_FORMAT', 'sell')
    assert CsrOperator(A).layout()[0] == 'sell'
    monkeypatch.setenv('RLH_SPMM_FORMAT', 'well')
    assert CsrOperator(sp.csr_matrix(D)).layout()[0] == 'wide'        # a 40-entry row does not fit 'well'
    assert CsrOperator(A.astype(np.float32)).layout()[0] == 'well'
    monkeypatch.setenv('RLH_SPMM_FORMAT', 'wide')
    lay, stored, ratio = CsrOperator(A).layout()
    assert lay == 'wide' and stored == 30 * 256 * 8                   # 30 blocks of 256 rows, one chunk of 8 slots


def test_spmm_windowed_schedule_many_blocks(monkeypatch):
    """More blocks than workgroups (335 blocks of 1024 rows): the XCD-aware launch order is a
    permutation of the blocks, so every row is computed exactly once."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    A = lap3d(70, 70, 70, 1.0, 1.01, 1.02)
    n = A.shape[0]
    rng = np.random.default_rng(9)
    x = rng.standard_normal((5, n))
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, 5)
    Y.fill(np.full((5, n), np.nan))
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < 1e-13


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('m', [1, 5, 13, 32])
@pytest.mark.parametrize('dma,pat', [('2', '1'), ('2', '0'), ('2', 'v'), ('0', '1')])
def test_spmm_stacked_blocks(monkeypatch, key, m, dma, pat):
    """The stacked windowed layout (two overlapping 1024-row blocks per workgroup, rlh_csr_stacks) forced on matrices
    that are too small to get it by default: 70 x 53 x 31 lap3d has 113 row blocks -- an odd number, so one stack has a
    single member -- and 115 010 rows, so the last block is ragged.  The LDS-DMA ring kernel (RLH_SPMM_STACK_DMA=2: for
    float32 too), and (=0, real types) the register-staged one; the complex types keep their interleaved layout beside the stacks.  Against the
    oracle, and bit for bit (real types: same entry order per row) against the unstacked kernel on the same handle."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    monkeypatch.setenv('RLH_SPMM_STACK_DMA', dma)
    # the value dictionary and the position patterns (27 distinct rows of values here), explicit entries ('0'), or the
    # value dictionary with explicit positions ('v')
    monkeypatch.setenv('RLH_SPMM_STACK_PAT', '1' if pat == 'v' else pat)
    monkeypatch.setenv('RLH_SPMM_STACK_DPAT', '0' if pat == 'v' else '1')
    # (complex: grid planes of exactly three row blocks, 111 blocks -- a stack's image must fit one slot of the ring,
    # which the skewed overlap of the other grid's planes and blocks exceeds at 8 and 16 bytes per element)
    A = _sym(lap3d(70, 53, 31, 1.0, 1.01, 1.02) if key in 'sd' else lap3d(64, 48, 37, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    rng = np.random.default_rng(40 + m)
    x = rnd((m, n), key, rng)
    op = SparseSymmetricMatrix(A)
    lay = op.layout()
    # (float64: uniform blocks pair up out of step on this grid -- planes of 70 x 53 = 3.6 blocks -- and most pairs' images
    # exceed a slot of the LDS-DMA ring, so the library cuts the blocks plane by plane: 4 blocks of 927 / 928 rows, 15 pairs
    # of planes and one plane of single blocks; float32 keeps the uniform blocks; the complex grid's planes are whole blocks)
    assert lay[0] == ('wide' if key in 'cz' else 'well') and lay[3] == {'s': 57, 'd': 64}.get(key, 56) and lay[5] < 0.8 * lay[4]
    X, Y = Vectors(x), Vectors(n, m, data_type=DT[key])
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply(X, Y)
    y = Y.data()
    tol = 2e-6 if key in 'sc' else 1e-13
    assert cases.rel(y, ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < tol
    monkeypatch.setenv('RLH_SPMM_STACK', '0')                                   # (read per call: the other layout)
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply(X, Y)
    if key in 'sd':
        assert np.array_equal(Y.data(), y)
    else:
        assert cases.rel(Y.data(), y) < tol


@pytest.mark.parametrize('key', ['d', 'c', 'z'])
@pytest.mark.parametrize('m', [5, 16])
def test_spmm_stacked_blocks_cut_plane_by_plane(monkeypatch, key, m):
    """Grid planes that are not a whole number of 1024-row blocks (50 x 50 = 2.44 blocks; BASELINE config 5: 126 x 126 = 15.5):
    the stack layout cuts its row blocks plane by plane (3 blocks of 833 / 834 rows) so that a block and its partner one
    plane on are exact translates -- 11 pairs of planes x 3 stacks of two + the last plane's 3 single blocks = 36 stacks --
    where uniform blocks pair up out of step (the library does so by itself only where the uniform stacks fall apart, as at
    126^3: RLH_SPMM_STACK_ALIGN=2 asks for it here).  Against the oracle, against the unstacked kernel of the same handle (bit for
    bit for the real type) and against a handle with uniform blocks (RLH_SPMM_STACK_ALIGN=0)."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    monkeypatch.setenv('RLH_SPMM_STACK_ALIGN', '2')
    A = _sym(lap3d(50, 50, 23, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    rng = np.random.default_rng(70 + m)
    x = rnd((m, n), key, rng)
    op = SparseSymmetricMatrix(A)
    assert op.layout()[3] == 36
    X, Y = Vectors(x), Vectors(n, m, data_type=DT[key])
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply(X, Y)
    y = Y.data()
    tol = 2e-6 if key == 'c' else 1e-13
    assert cases.rel(y, ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < tol
    monkeypatch.setenv('RLH_SPMM_STACK', '0')                                   # (read per call: the other layout)
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply(X, Y)
    assert np.array_equal(Y.data(), y) if key == 'd' else cases.rel(Y.data(), y) < tol
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    monkeypatch.setenv('RLH_SPMM_STACK_ALIGN', '0')
    uniform = SparseSymmetricMatrix(A)
    assert uniform.layout()[3] != 36
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    uniform.apply(X, Y)
    assert np.array_equal(Y.data(), y) if key == 'd' else cases.rel(Y.data(), y) < tol


@pytest.mark.parametrize('align', ['0', '2'])
def test_fused_chebyshev_step_on_stacks_complex128(monkeypatch, align):
    """The fused Chebyshev step of a complex128 operator on the stacked LDS-DMA ring (well_stack_dma_kernel<..., CHEB>: y[row]
    out of the staged image through slot 7 of the row, p and b as ordinary loads behind waits for everything): uniform
    stacks and stacks cut plane by plane (members shorter than 1024 rows: idle lanes), 64 and 5 vectors, against the oracle
    and against the interleaved kernel of the same handle; y and b untouched; RLH_SPMM_STACK_CHEB=2 makes the library
    refuse anything but the stacked kernel, so the comparison is between the two kernels for sure."""
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    monkeypatch.setenv('RLH_SPMM_STACK_ALIGN', align)
    A = _sym(lap3d(50, 50, 23, 1.0, 1.01, 1.02), 'z')
    n = A.shape[0]
    op = SparseSymmetricMatrix(A)
    assert (op.layout()[3] == 36) == (align == '2')
    up = sp.triu(A, format='csr')
    for m in (64, 5):
        rng = np.random.default_rng(90 + m)
        y0, p0, b0 = (rnd((m, n), 'z', rng) for _ in range(3))
        want = 1.3 * y0 - 0.3 * p0 - 1.7 * (b0 - ops.csr_sym_apply(up, y0))
        y, p, b = Vectors(y0.copy()), Vectors(p0.copy()), Vectors(b0.copy())
        monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '2')
        op.cheb_step(y, p, b, 1.3, -0.3, -1.7)
        got = p.data()
        assert cases.rel(got, want) < 1e-13
        assert np.array_equal(y.data(), y0) and np.array_equal(b.data(), b0)
        monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '0')
        p.fill(p0.copy())
        op.cheb_step(y, p, b, 1.3, -0.3, -1.7)
        assert cases.rel(p.data(), got) < 1e-14
    # a row that does not store its diagonal: no own column to find in the image -- the interleaved kernel as before
    B = sp.lil_matrix(A)
    B[77, 77] = 0.0
    B = sp.csr_matrix(B)
    B.eliminate_zeros()
    op2 = SparseSymmetricMatrix(B)
    y, p, b = Vectors(y0.copy()), Vectors(p0.copy()), Vectors(b0.copy())
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '2')
    with pytest.raises(_lib.RlhError):
        op2.cheb_step(y, p, b, 1.3, -0.3, -1.7)
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '1')
    op2.cheb_step(y, p, b, 1.3, -0.3, -1.7)
    assert cases.rel(p.data(), 1.3 * y0 - 0.3 * p0 - 1.7 * (b0 - ops.csr_sym_apply(sp.triu(B, format='csr'), y0))) < 1e-13


def test_fused_chebyshev_step_on_stacks_repeats_at_config5_size(monkeypatch):
    """BASELINE config 5's operator (126^3, complex128, 64 vectors, every member of a stack shorter than 1024 rows): 100 fused
    steps from the same p, each bit for bit the first (every wait of this kernel is for everything outstanding: a race would
    show as a difference once in thousands of stacks), the first against the interleaved kernel."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import hermitian_lap3d_rows
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    N, m = 126, 64
    n = N ** 3
    op = CsrOperator(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n))
    y, p, b, p0 = (Vectors(n, m, data_type=np.complex128) for _ in range(4))
    for v in (y, p0, b):
        v.fill_random()
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '2')
    first = None
    for it in range(100):
        p0.copy(p)
        op.cheb_step_ptr(m, y, p, b, 0.9, -0.2, 0.011)
        got = p.data()
        if first is None:
            first = got
        else:
            assert np.array_equal(got, first), 'fused step %d differs' % it
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '0')
    p0.copy(p)
    op.cheb_step_ptr(m, y, p, b, 0.9, -0.2, 0.011)
    assert cases.rel(p.data(), first) < 1e-14


@pytest.mark.parametrize('rows', [(0, 30000), (17000, 49000), (40000, 64000)])
def test_fused_chebyshev_step_on_stacks_complex128_row_shard(monkeypatch, rows):
    """The stacked fused step on a ROW SHARD (what every rank of a multi-GPU run of BASELINE config 5 executes): own columns
    from y, the others from a halo block; part 1 (interior rows, handed NaNs for the halo, which it must not read) + part 2
    (boundary rows) = the full step, against NumPy; RLH_SPMM_STACK_CHEB=2 refuses any other kernel."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    A = _sym(lap3d(40, 40, 40, 1.0, 1.01, 1.02), 'z')
    n = A.shape[0]
    r0, r1 = rows
    loc = sp.csr_matrix(A[r0:r1])
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    nown = r1 - r0
    n_own_pad = -(-nown // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(nown)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    nh = -(-len(halo) // 8) * 8
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    rng = np.random.default_rng(5)
    m = 64
    y0 = rnd((m, n), 'z', rng)
    p0, b0 = (rnd((m, nown), 'z', rng) for _ in range(2))

    def block(a, rows_alloc):
        pad = np.zeros((m, rows_alloc), dtype=np.complex128)
        pad[:, :a.shape[1]] = a
        return Vectors(pad)
    y, p, b = block(y0[:, r0:r1], n_own_pad), block(p0, n_own_pad), block(b0, n_own_pad)
    hgood = block(y0[:, halo], nh)
    hbad = block(np.full((m, len(halo)), np.nan, dtype=np.complex128), nh)
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '2')
    op.cheb_step_ptr(m, y, p, b, 1.3, -0.3, 0.01, hbad.data_ptr(), hbad.ld(), part=1)
    op.cheb_step_ptr(m, y, p, b, 1.3, -0.3, 0.01, hgood.data_ptr(), hgood.ld(), part=2)
    t = (sp.csr_matrix(A[r0:r1]) @ y0.T).T
    want = 1.3 * y0[:, r0:r1] - 0.3 * p0 + 0.01 * (b0 - t)
    got = p.data()[:, :nown]
    assert np.all(np.isfinite(got))
    assert cases.rel(got, want) < 1e-13
    # and all rows in one call, halo in place
    p2 = block(p0, n_own_pad)
    op.cheb_step_ptr(m, y, p2, b, 1.3, -0.3, 0.01, hgood.data_ptr(), hgood.ld(), part=0)
    assert np.array_equal(p2.data()[:, :nown], got)


@pytest.mark.parametrize('key', ['d', 's'])
def test_product_on_a_row_shard_of_the_headline_operator(monkeypatch, key):
    """The second of eight row shards of lap3d 215^3 (1 242 304 rows, one grid plane of halo rows on either side), built the way
    ShardedSparseMatrix builds it, with the library's default choices: the stacks exist, interior rows + boundary rows = the
    product of the shard's rows (SciPy), the interior part never reads the halo block (handed NaNs); float32: the operator
    takes the bfloat16 Chebyshev step."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import lap3d_rows
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    N, m = 215, 8
    n = N ** 3
    per = -(-(n // 8) // 64) * 64
    r0, r1 = per, 2 * per
    dt = DT[key]
    loc = sp.csr_matrix(lap3d_rows(N, N, N, 1.0, 1.01, 1.02, r0, r1)).astype(dt)
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    nown = r1 - r0
    n_own_pad = -(-nown // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(nown)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    nh = -(-len(halo) // 8) * 8
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    assert op.layout()[0] == 'well' and op.stacks()[0] > 0
    if key == 's':
        assert op.bf16_ready(nh)
    rng = np.random.default_rng(12)
    xown, xhalo = rnd((m, nown), key, rng), rnd((m, len(halo)), key, rng)

    def block(a, rows_alloc):
        pad = np.zeros((m, rows_alloc), dtype=dt)
        pad[:, :a.shape[1]] = a
        return Vectors(pad)
    x, y = block(xown, n_own_pad), Vectors(n_own_pad, m, data_type=dt)
    hgood, hbad = block(xhalo, nh), block(np.full((m, len(halo)), np.nan, dtype=dt), nh)
    y.fill(np.full((m, n_own_pad), np.nan, dtype=dt))
    op.apply_ptr(m, x.data_ptr(), x.ld(), y.data_ptr(), y.ld(), hbad.data_ptr(), hbad.ld(), part=1)
    op.apply_ptr(m, x.data_ptr(), x.ld(), y.data_ptr(), y.ld(), hgood.data_ptr(), hgood.ld(), part=2)
    xcat = np.zeros((m, L.shape[1]), dtype=dt)
    xcat[:, :nown] = xown
    xcat[:, n_own_pad:n_own_pad + len(halo)] = xhalo
    want = (L.astype(np.float64) @ xcat.T.astype(np.float64)).T
    got = y.data()[:, :nown]
    assert np.all(np.isfinite(got)) and cases.rel(got, want) < (3e-6 if key == 's' else 1e-13)


def test_fused_chebyshev_step_on_stacks_of_a_config5_row_shard(monkeypatch):
    """The second of eight row shards of BASELINE config 5's operator (126^3 complex128: 250 047 rows = 15.75 grid planes, one
    halo plane on either side), built the way ShardedSparseMatrix builds it, with the library's DEFAULT choices: the stacks
    exist, the fused step runs on them (RLH_SPMM_STACK_CHEB=2 refuses anything else) in the two parts of an overlapped
    exchange, and equals NumPy."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import hermitian_lap3d_rows
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    N, m = 126, 64
    n = N ** 3
    r0, r1 = n // 8, 2 * (n // 8)
    loc = sp.csr_matrix(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, r0, r1))
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    nown = r1 - r0
    n_own_pad = -(-nown // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(nown)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    nh = -(-len(halo) // 8) * 8
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    assert op.stacks()[0] > 0
    rng = np.random.default_rng(6)
    yown, p0, b0 = (rnd((m, nown), 'z', rng) for _ in range(3))
    yhalo = rnd((m, len(halo)), 'z', rng)

    def block(a, rows_alloc):
        pad = np.zeros((m, rows_alloc), dtype=np.complex128)
        pad[:, :a.shape[1]] = a
        return Vectors(pad)
    y, p, b = block(yown, n_own_pad), block(p0, n_own_pad), block(b0, n_own_pad)
    hgood = block(yhalo, nh)
    hbad = block(np.full((m, len(halo)), np.nan, dtype=np.complex128), nh)
    monkeypatch.setenv('RLH_SPMM_STACK_CHEB', '2')
    op.cheb_step_ptr(m, y, p, b, 0.9, -0.2, 0.011, hbad.data_ptr(), hbad.ld(), part=1)
    op.cheb_step_ptr(m, y, p, b, 0.9, -0.2, 0.011, hgood.data_ptr(), hgood.ld(), part=2)
    ycat = np.zeros((m, L.shape[1]), dtype=np.complex128)
    ycat[:, :nown] = yown
    ycat[:, n_own_pad:n_own_pad + len(halo)] = yhalo
    want = 0.9 * yown - 0.2 * p0 + 0.011 * (b0 - (L @ ycat.T).T)
    got = p.data()[:, :nown]
    assert np.all(np.isfinite(got)) and cases.rel(got, want) < 1e-13


def test_plane_aligned_stacks_repeat_at_config5_size(monkeypatch):
    """The stacks of BASELINE config 5's operator (126^3, complex128: row blocks cut plane by plane, EVERY member shorter than
    1024 rows, so every wait of the LDS-DMA ring is a counted one on a stack with idle lanes): 300 products of 64 vectors, each
    bit for bit the first (a miscounted wait is a race that shows once in thousands of stacks), the first against the
    interleaved kernel of the same handle."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import hermitian_lap3d_rows
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    N, m = 126, 64
    n = N ** 3
    op = CsrOperator(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n))
    assert op.stacks()[0] == 63 * 16
    X, Y = Vectors(n, m, data_type=np.complex128), Vectors(n, m, data_type=np.complex128)
    np.random.seed(11)
    X.fill_random()
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
    first = Y.data()
    for rep in range(300):
        op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
        if rep % 50 == 49:                              # (the launches in between keep the queue full)
            assert np.array_equal(Y.data(), first), rep
    monkeypatch.setenv('RLH_SPMM_STACK', '0')
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
    assert cases.rel(Y.data(), first) < 1e-13


@pytest.mark.parametrize('key', ['d', 's', 'z'])
def test_spmm_stacked_blocks_with_halo(monkeypatch, key):
    """The stacked layout on a row shard (rlh_spmm_part): the LDS-DMA kernel fetches the pieces right of the own / halo
    boundary from the halo block; part 1 (stacks without halo columns) must not touch it -- NaNs here -- and parts 1 + 2,
    as well as the whole shard in one call, give the product.  Grid planes of exactly three row blocks, so that the
    stacks' images fit a ring slot for every type."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    A = _sym(lap3d(64, 48, 30, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    r0, r1 = 3072 * 4 + 100, 3072 * 26 + 777
    loc = A[r0:r1]
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    n_own_pad = -(-(r1 - r0) // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(r1 - r0)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    ncols = -(-(n_own_pad + len(halo)) // 8) * 8
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(r1 - r0, ncols))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    assert op.stacks()[0] > 0
    rng = np.random.default_rng(13)
    m = 9
    x = rnd((m, n), key, rng)
    xo = np.zeros((m, n_own_pad), dtype=DT[key])
    xo[:, :r1 - r0] = x[:, r0:r1]
    hg = np.zeros((m, ncols - n_own_pad), dtype=DT[key])
    hg[:, :len(halo)] = x[:, halo]
    X, Hgood = Vectors(xo), Vectors(hg)
    Hbad = Vectors(np.full((m, ncols - n_own_pad), np.nan, dtype=DT[key]))
    Y = Vectors(r1 - r0, m, data_type=DT[key])
    Y.fill(np.full((m, r1 - r0), 777, dtype=DT[key]))
    ref = (A @ x.T).T[:, r0:r1]
    tol = 3e-6 if key == 's' else 1e-13
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hbad.data_ptr(), Hbad.ld(), part=1)
    y1 = Y.data()
    done = y1[0] != 777
    assert 0.5 < done.mean() < 1.0 and np.all(np.isfinite(y1[:, done])) and cases.rel(y1[:, done], ref[:, done]) < tol
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hgood.data_ptr(), Hgood.ld(), part=2)
    y12 = Y.data()
    assert cases.rel(y12, ref) < tol
    Y.fill(np.full((m, r1 - r0), 777, dtype=DT[key]))
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hgood.data_ptr(), Hgood.ld(), part=0)
    assert np.array_equal(Y.data(), y12)
    monkeypatch.setenv('RLH_SPMM_STACK', '0')               # the other layout of the same handle
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hgood.data_ptr(), Hgood.ld(), part=0)
    assert cases.rel(Y.data(), y12) < tol


@pytest.mark.parametrize('side,stacks,products,steps', [(110, 650, 150, 300), (215, 4853, 100, 1000)])
def test_stacked_kernels_repeat_bit_for_bit(monkeypatch, side, stacks, products, steps):
    """The LDS-DMA kernels wait on counted vmcnt values written by hand; a wrong count is a race that shows once in a
    thousand launches at full size (the first bfloat16 version passed every small test and died of NaNs in the 10^7-row
    solve: about one bad stack in 5 10^6).  So: many launches on operators with more stacks than CUs -- lap3d 110^3 and
    the roofline point 215^3 (1000 steps x 4853 stacks) -- every result bit for bit equal to the unstacked kernel's:
    fp64 products of 32 vectors and chained bfloat16 Chebyshev steps of 16."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    A = lap3d(side, side, side, 1.0, 1.01, 1.02)
    n = A.shape[0]
    rng = np.random.default_rng(77)
    op = SparseSymmetricMatrix(A)
    assert op.layout()[3] == stacks
    col = rng.standard_normal((1, n))
    X, Y = Vectors(n, 32), Vectors(n, 32)
    for j in range(32):
        X.select(1, j)
        X.fill(np.roll(col, 11 * j + 1, axis=1) * (1 + 0.01 * j))
    X.select(32)
    monkeypatch.setenv('RLH_SPMM_STACK', '0')
    op.apply(X, Y)
    want = Y.data()
    monkeypatch.setenv('RLH_SPMM_STACK', '1')
    for rep in range(products):
        op.apply(X, Y)
        if rep % 25 == 24 or rep < 2:                     # (the launches in between keep the queue full)
            assert np.array_equal(Y.data(), want), rep
    del X, Y, want, op
    op32 = SparseSymmetricMatrix(A.astype(np.float32))
    m = 16
    col32 = col.astype(np.float32)
    y0, p0, b0 = (ops.bf16_round(np.concatenate([np.roll(col32, 7 * j + 3 * k + 1, axis=1) for j in range(m)])) for k in range(3))

    def run(stacked, steps):
        monkeypatch.setenv('RLH_SPMM_STACK_BF16', '1' if stacked else '0')
        blocks = []
        for a in (y0, p0, b0):
            blk = Bf16Block(n, m)
            blk.pack(Vectors(a), 1.0)
            blocks.append(blk)
        y, p, b = blocks
        for _ in range(steps):
            op32.cheb_step_bf16(m, y, p, b, 1.02, -0.25, 2e-6)
            y, p = p, y
        out = Vectors(n, m, data_type=np.float32)
        y.unpack(out)
        return out.data()
    got, ref = run(True, steps), run(False, steps)
    assert np.all(np.isfinite(ref)) and np.array_equal(got, ref)


def test_spmm_stacks_only_where_they_pay(monkeypatch):
    """Default policy: no stacks on a small stencil (fewer than four row blocks per CU); stacks of two planes on a large one."""
    from raleigh_amd.algebra.hip import SparseSymmetricMatrix
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.delenv('RLH_SPMM_STACK', raising=False)
    assert SparseSymmetricMatrix(lap3d(40, 40, 40, 1.0, 1.01, 1.02)).layout()[3] == 0
    lay = SparseSymmetricMatrix(lap3d(110, 110, 110, 1.0, 1.01, 1.02)).layout()      # 1 300 row blocks
    assert lay[0] == 'well' and lay[3] == 650 and lay[5] < 0.75 * lay[4]


@pytest.mark.parametrize('key', KEYS)
def test_fill_random_bit_exact(key):
    """rlh_fill_random against its oracle restatement: bit-exact, with row / vector offsets and
    a leading dimension larger than the block."""
    import ctypes
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors
    n, m = 70001, 5
    V = Vectors(n, m + 1, data_type=DT[key])
    V.fill(np.full((m + 1, n), 7, dtype=DT[key]))
    V.select(m, 1)
    _lib.check(_lib.lib().rlh_fill_random(_lib.DTYPE_CODE[DT[key]], n, m, V.data_ptr(), V.ld(),
                                          0x123456789ABCDEF, 1000, 2))
    V.select(m + 1, 0)
    got = V.data()
    assert np.array_equal(got[1:], ops.uniform_block(0x123456789ABCDEF, n, m, DT[key], row0=1000, col0=2))
    assert np.all(got[0] == 7)


def test_fill_random_large_block_uses_device_generator():
    from raleigh_amd.algebra.hip import Vectors
    n, m = 1 << 20, 5                                   # 5 M elements >= DEVICE_RANDOM_THRESHOLD
    V = Vectors(n, m)
    np.random.seed(3)
    V.fill_random()
    np.random.seed(3)
    seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
    assert np.array_equal(V.data(), ops.uniform_block(seed, n, m, np.float64))


@pytest.mark.parametrize('key', ['d', 's'])
def test_spmm_interior_boundary_parts(spmm_format, key):
    """rlh_spmm_part / rlh_spmm_cheb_part: part 1 (rows without halo columns) must not depend on
    the halo block -- it is handed NaNs here -- and parts 1 + 2 together give the full product."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    A = _sym(lap3d(40, 40, 40, 1.0, 1.01, 1.02), key)
    n = A.shape[0]
    r0, r1 = 20000, 44001
    loc = A[r0:r1]
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    n_own_pad = -(-(r1 - r0) // 4) * 4
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(r1 - r0)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr),
                      shape=(r1 - r0, n_own_pad + len(halo)))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    rng = np.random.default_rng(12)
    m = 7
    x = rnd((m, n), key, rng)
    xo = np.zeros((m, n_own_pad), dtype=DT[key])
    xo[:, :r1 - r0] = x[:, r0:r1]
    X, Hgood = Vectors(xo), Vectors(np.ascontiguousarray(x[:, halo]))
    Hbad = Vectors(np.full((m, len(halo)), np.nan, dtype=DT[key]))
    Y = Vectors(r1 - r0, m, data_type=DT[key])
    Y.fill(np.full((m, r1 - r0), 777, dtype=DT[key]))
    ref = (A @ x.T).T[:, r0:r1]
    tol = 3e-6 if key == 's' else 1e-13
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hbad.data_ptr(), Hbad.ld(), part=1)
    y1 = Y.data()
    done = y1[0] != 777                                   # rows written by part 1
    assert np.all(np.isfinite(y1[:, done])) and cases.rel(y1[:, done], ref[:, done]) < tol
    if spmm_format in ('well', 'wide'):
        assert 0.5 < done.mean() < 1.0                    # most blocks are interior, the shard ends are not
    else:
        assert not done.any()                             # sliced layout: everything is left to part 2
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), Hgood.data_ptr(), Hgood.ld(), part=2)
    assert cases.rel(Y.data(), ref) < tol
    # fused Chebyshev step in two parts
    p0, b0 = rnd((m, r1 - r0), key, rng), rnd((m, r1 - r0), key, rng)
    P, B = Vectors(p0.copy()), Vectors(b0.copy())
    op.cheb_step_ptr(m, X, P, B, 1.2, -0.2, 0.4, Hbad.data_ptr(), Hbad.ld(), part=1)
    op.cheb_step_ptr(m, X, P, B, 1.2, -0.2, 0.4, Hgood.data_ptr(), Hgood.ld(), part=2)
    assert cases.rel(P.data(), 1.2 * x[:, r0:r1] - 0.2 * p0 + 0.4 * (b0 - ref)) < tol


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('shape', [(7, 5, 9, 4), (16, 16, 16, 16), (3, 20, 10, 27), (32, 32, 40, 24), (30, 34, 17, 33),
                                   (64, 64, 32, 32), (64, 64, 64, 64), (60, 64, 64, 50)])
def test_combine2_two_outputs_one_pass(key, shape):
    """rlh_block_update2x2 / Vectors.combine2: [A | B] = X qx + Y qy in one pass, against the oracle
    (odd row count: the 16-byte row groups end in a partial one)."""
    from raleigh_amd.algebra.hip import Vectors
    k1, k2, ma, mb = shape
    n = 20011
    rng = np.random.default_rng(k1 + ma)
    x, y = rnd((k1, n), key, rng), rnd((k2, n), key, rng)
    qxa, qxb = rnd((k1, ma), key, rng), rnd((k1, mb), key, rng)
    qya, qyb = rnd((k2, ma), key, rng), rnd((k2, mb), key, rng)
    X, Y = Vectors(x), Vectors(y)
    A, B = Vectors(n, ma, data_type=DT[key]), Vectors(n, mb, data_type=DT[key])
    X.combine2(qxa, qxb, Y, qya, qyb, A, B)
    tol = 2e-5 if key in 'sc' else 1e-13
    assert cases.rel(A.data(), ops.multiply(x, qxa) + ops.multiply(y, qya)) < tol
    assert cases.rel(B.data(), ops.multiply(x, qxb) + ops.multiply(y, qyb)) < tol
    assert np.array_equal(X.data(), x) and np.array_equal(Y.data(), y)


@pytest.mark.parametrize('key', KEYS)
def test_absmax(key):
    """rlh_absmax / Matrix.absmax: the exact maximum (no rounding involved), C- and F-ordered data,
    a leading dimension larger than the row."""
    from raleigh_amd.algebra.hip import Matrix
    rng = np.random.default_rng(31)
    a = rnd((301, 1237), key, rng)
    a[17, 1001] *= 40
    want = max(np.max(np.abs(a.real)), np.max(np.abs(a.imag)))
    assert Matrix(a).absmax() == float(want)
    assert Matrix(np.asfortranarray(a)).absmax() == float(want)


def test_bf16_pack_unpack_bit_exact():
    """rlh_bf16_pack / unpack: round-to-nearest-even bit patterns identical to the oracle's."""
    import ctypes
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    rng = np.random.default_rng(41)
    n, m = 10007, 5
    for dt in (np.float64, np.float32):
        x = (rng.standard_normal((m, n)) * 10.0 ** rng.integers(-20, 20, (m, 1))).astype(dt)
        x[0, :4] = [0.0, 1.0, 1.00390625, -3.0]                  # exact and tie cases
        X = Vectors(x)
        blk = Bf16Block(n, m)
        blk.pack(X, 0.75)
        Y = Vectors(n, m, data_type=dt)
        blk.unpack(Y)
        want = ops.bf16_round(np.float32(0.75) * x.astype(np.float32))
        assert np.array_equal(Y.data(), want.astype(dt))


@pytest.mark.parametrize('m', [1, 8, 13, 16, 32])
def test_fused_chebyshev_step_bf16(m):
    """rlh_spmm_cheb_bf16 against float32 NumPy on the SAME bfloat16 inputs: the only differences are
    the summation order of the 7 products per row and one final rounding to bfloat16."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    A = lap3d(23, 19, 17, 1.0, 1.01, 1.02).astype(np.float32)
    n = A.shape[0]
    rng = np.random.default_rng(5 + m)
    y0, p0, b0 = (ops.bf16_round(rng.standard_normal((m, n)).astype(np.float32)) for _ in range(3))
    op = SparseSymmetricMatrix(A)
    assert op.supports_bf16()
    blocks = []
    for a in (y0, p0, b0):
        blk = Bf16Block(n, m)
        blk.pack(Vectors(a), 1.0)
        blocks.append(blk)
    y, p, b = blocks
    op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01)
    out = Vectors(n, m, data_type=np.float32)
    p.unpack(out)
    t = (sp.csr_matrix(A) @ y0.T).T.astype(np.float32)
    exact = np.float32(1.3) * y0 + np.float32(-0.3) * p0 + np.float32(0.01) * (b0 - t)
    got = out.data()
    assert np.all(np.abs(got - exact) <= 2.0 ** -8 * np.abs(exact) + 1e-6)     # half a bf16 ulp + float32 noise
    chk = Vectors(n, m, data_type=np.float32)
    y.unpack(chk)
    assert np.array_equal(chk.data(), y0)                                      # inputs untouched


def test_full_size_roofline_point_fp64():
    """The BASELINE roofline point itself: n = 215^3 = 9 938 375 rows, m = 32, fp64, every block in
    HBM (start block from the device generator, so the host only ever sees m x m results).
    Size-independent checks: Gram symmetric with diag == dots and the trace of a U(-1,1) block;
    linearity (X Q)^H Y = Q^H (X^H Y); the SpMM is self-adjoint, <A X, Y> = <X, A Y>, and maps a
    constant vector to the row sums of the matrix."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    N, m = 215, 32
    A = lap3d(N, N, N, 1.0, 1.01, 1.02)
    n = A.shape[0]
    np.random.seed(5)
    X, Y, W, Z = (Vectors(n, m) for _ in range(4))
    X.fill_random()
    Y.fill_random()
    G = X.dot(X)
    d = X.dots(X)
    assert cases.rel(G, G.T) < 1e-14 and cases.rel(np.diag(G), d) < 1e-13
    assert abs(np.trace(G) / (n * m) - 1.0 / 3.0) < 1e-3                   # E[u^2] = 1/3 for U(-1, 1)
    assert np.max(np.abs(G - np.diag(np.diag(G)))) < 1e-3 * np.max(np.diag(G))   # independent columns
    Q = np.random.randn(m, m)
    X.multiply(Q, W)
    # dot(other)[i, j] = <other_i, self_j>: W = X Q  =>  W.dot(Y) = (X.dot(Y)) Q
    assert cases.rel(W.dot(Y), X.dot(Y) @ Q) < 1e-12
    op = SparseSymmetricMatrix(A)
    op.apply(X, W)                                                          # W = A X
    op.apply(Y, Z)                                                          # Z = A Y
    assert cases.rel(W.dot(Y), Z.dot(X).T) < 1e-12                          # <y_i, A x_j> = <A y_i, x_j>
    c = np.ones((2, n))
    c[1] *= 3.0                                                             # (two vectors through the host: 160 MB)
    C = Vectors(c)
    R = Vectors(n, 2)
    op.apply(C, R)
    rows = np.asarray(A.sum(axis=1)).ravel()
    got = R.data()
    scale = np.max(np.abs(A.data))                  # interior row sums cancel to ~0: compare on the entries' scale
    assert np.max(np.abs(got[0] - rows)) < 1e-13 * scale and np.max(np.abs(got[1] - 3.0 * rows)) < 3e-13 * scale


@pytest.mark.parametrize('key', ['d', 's'])
def test_spmm_layouts_agree_on_a_large_irregular_matrix(monkeypatch, key):
    """n = 3*10^5 rows, 1..27 entries per row (banded couplings of varying reach plus a few far
    ones, empty rows): more than a thousand 256-row blocks of the interleaved layout through the
    XCD-aware schedule, one to four entry chunks per row -- against the sliced kernel, which does
    one gather per entry and shares nothing with it beyond the CSR input."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    rng = np.random.default_rng(77)
    n = 300007
    reach = np.repeat(rng.integers(1, 14, n // 1000 + 1), 1000)[:n]          # couplings per row: 1 + 2 * reach
    reach[5000:6000] = 0
    rows = np.repeat(np.arange(n), 2 * reach + 1)
    offs = np.concatenate([np.arange(-r, r + 1) * (1 + (i % 7)) for i, r in enumerate(reach)])
    cols = np.clip(rows + offs, 0, n - 1)
    A = sp.coo_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    keep = np.ones(n)
    keep[7000:7100] = 0                                                         # empty rows
    A = sp.csr_matrix(sp.diags(keep) @ A).astype(DT[key])
    A.eliminate_zeros()
    assert np.diff(A.indptr).max() <= 32 and np.diff(A.indptr).min() == 0
    m = 12
    x = rnd((m, n), key, rng)
    X = Vectors(x)
    out = {}
    for fmt in ('sell', 'wide'):
        monkeypatch.setenv('RLH_SPMM_FORMAT', fmt)
        op = CsrOperator(A)
        assert op.layout()[0] == fmt
        Y = Vectors(n, m, data_type=DT[key])
        op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
        out[fmt] = Y.data()
    assert cases.rel(out['wide'], out['sell']) < (2e-6 if key == 's' else 1e-14)
    i = rng.integers(0, n, 200)                                                 # and a sample of rows against SciPy
    ref = (A[i] @ x.T).T
    assert cases.rel(out['wide'][:, i], ref) < (2e-6 if key == 's' else 1e-13)


def test_new_entry_points_reject_bad_arguments(monkeypatch):
    """Error behaviour of the extensions: a message, no launch."""
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    A = lap3d(12, 11, 10, 1.0, 1.01, 1.02)
    n = A.shape[0]
    op64 = CsrOperator(A)
    X, Y = Vectors(n, 2), Vectors(n, 2)
    with pytest.raises(_lib.RlhError, match='part must be'):
        op64.apply_ptr(2, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), part=3)
    blocks = [Bf16Block(n, 2) for _ in range(3)]
    with pytest.raises(_lib.RlhError, match='float32 operator'):
        op64.cheb_step_bf16(2, *blocks, 1.0, 0.0, 1.0)                       # float64 operator
    monkeypatch.setenv('RLH_SPMM_FORMAT', 'sell')
    op32s = CsrOperator(A.astype(np.float32))
    with pytest.raises(_lib.RlhError, match='windowed layout'):
        op32s.cheb_step_bf16(2, *blocks, 1.0, 0.0, 1.0)                      # sliced layout
    monkeypatch.delenv('RLH_SPMM_FORMAT')
    op32 = CsrOperator(A.astype(np.float32))
    with pytest.raises(_lib.RlhError, match='P is updated in place|bad block pointers'):
        op32.cheb_step_bf16(2, blocks[0], blocks[0], blocks[1], 1.0, 0.0, 1.0)   # p aliases y
    with pytest.raises(_lib.RlhError, match='real blocks only'):
        blocks[0].pack(Vectors(n, 2, data_type=np.complex128))


@pytest.mark.parametrize('rows', [(20000, 44000), (20003, 43998)])
def test_fused_chebyshev_step_bf16_row_shard(rows):
    """rlh_spmm_cheb_bf16_part on a row shard: own columns from y, the rest from a bfloat16 halo block;
    part 1 (handed NaNs for the halo) + part 2 = the full step; against float32 NumPy on the same
    bfloat16 inputs."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    A = lap3d(40, 40, 40, 1.0, 1.01, 1.02).astype(np.float32)
    n = A.shape[0]
    r0, r1 = rows
    loc = sp.csr_matrix(A[r0:r1])
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    nown = r1 - r0
    n_own_pad = -(-nown // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(nown)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + len(halo)))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    assert op.layout()[0] == 'well'
    rng = np.random.default_rng(3)
    m = 12
    y0 = ops.bf16_round(rng.standard_normal((m, n)).astype(np.float32))
    p0, b0 = (ops.bf16_round(rng.standard_normal((m, nown)).astype(np.float32)) for _ in range(2))

    def block(a, rows_alloc):
        blk = Bf16Block(rows_alloc, m)
        pad = np.zeros((m, rows_alloc), dtype=np.float32)
        pad[:, :a.shape[1]] = a
        blk.pack(Vectors(pad), 1.0)
        return blk
    nh = -(-len(halo) // 8) * 8
    y, p, b = block(y0[:, r0:r1], n_own_pad), block(p0, n_own_pad), block(b0, n_own_pad)
    hgood = block(y0[:, halo], nh)
    hbad = block(np.full((m, len(halo)), np.nan, dtype=np.float32), nh)
    op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01, hbad.ptr(), hbad.ld, part=1)
    op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01, hgood.ptr(), hgood.ld, part=2)
    out = Vectors(n_own_pad, m, data_type=np.float32)
    p.unpack(out)
    t = (sp.csr_matrix(A[r0:r1]) @ y0.T).T.astype(np.float32)
    exact = np.float32(1.3) * y0[:, r0:r1] + np.float32(-0.3) * p0 + np.float32(0.01) * (b0 - t)
    got = out.data()[:, :nown]
    assert np.all(np.isfinite(got))
    assert np.all(np.abs(got - exact) <= 2.0 ** -8 * np.abs(exact) + 1e-6)


@pytest.mark.parametrize('m', [1, 7, 8, 13, 16, 29])
def test_fused_chebyshev_step_bf16_on_stacks(monkeypatch, m):
    """The bfloat16 step on the stacked layout (LDS-DMA ring of eight slots, groups of eight vectors; forced on a
    matrix too small to get the stacks by default: 113 row blocks, so one stack of a single member, ragged last block):
    against float32 NumPy on the same bfloat16 inputs, and bit for bit against the unstacked kernel."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    A = lap3d(70, 53, 31, 1.0, 1.01, 1.02).astype(np.float32)
    n = A.shape[0]
    rng = np.random.default_rng(50 + m)
    y0, p0, b0 = (ops.bf16_round(rng.standard_normal((m, n)).astype(np.float32)) for _ in range(3))
    op = SparseSymmetricMatrix(A)
    assert op.supports_bf16() and op.layout()[3] == 57

    def run():
        blocks = []
        for a in (y0, p0, b0):
            blk = Bf16Block(n, m)
            blk.pack(Vectors(a), 1.0)
            blocks.append(blk)
        y, p, b = blocks
        op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01)
        out = Vectors(n, m, data_type=np.float32)
        p.unpack(out)
        chk = Vectors(n, m, data_type=np.float32)
        y.unpack(chk)
        assert np.array_equal(chk.data(), y0)
        return out.data()
    got = run()
    A64 = sp.csr_matrix(A).astype(np.float64)
    t = (A64 @ y0.T.astype(np.float64)).T
    exact = 1.3 * y0 + -0.3 * p0 + 0.01 * (b0 - t)
    # half a bfloat16 ulp of the result + the float32 rounding of the sum (entries of 1e4 here: a result that sits
    # on a rounding tie may go either way)
    mag = 1.3 * np.abs(y0) + 0.3 * np.abs(p0) + 0.01 * (np.abs(b0) + (abs(A64) @ np.abs(y0).T.astype(np.float64)).T)
    assert np.all(np.abs(got - exact) <= 2.0 ** -8 * np.abs(exact) + 8 * 2.0 ** -24 * mag)
    monkeypatch.setenv('RLH_SPMM_STACK', '0')
    assert np.array_equal(run(), got)


def test_fused_chebyshev_step_bf16_on_stacks_row_shard(monkeypatch):
    """... and on a row shard: pieces right of the own / halo boundary from the bfloat16 halo block, part 1 (NaNs for
    the halo) + part 2 = the whole step.  Planes of three row blocks (64 x 48)."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.algebra.hip.sparse import Bf16Block
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    monkeypatch.setenv('RLH_SPMM_STACK', '2')
    A = lap3d(64, 48, 30, 1.0, 1.01, 1.02).astype(np.float32)
    n = A.shape[0]
    r0, r1 = 3072 * 4 + 104, 3072 * 26 + 777
    loc = sp.csr_matrix(A[r0:r1])
    used = np.unique(loc.indices)
    halo = used[(used < r0) | (used >= r1)]
    nown = r1 - r0
    n_own_pad = -(-nown // 8) * 8
    newcol = np.full(n, -1, dtype=np.int64)
    newcol[r0:r1] = np.arange(nown)
    newcol[halo] = n_own_pad + np.arange(len(halo))
    nh = -(-len(halo) // 8) * 8
    L = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
    L.sort_indices()
    op = CsrOperator(L, n_own=n_own_pad)
    assert op.layout()[0] == 'well' and op.stacks()[0] > 0
    rng = np.random.default_rng(4)
    m = 12
    y0 = ops.bf16_round(rng.standard_normal((m, n)).astype(np.float32))
    p0, b0 = (ops.bf16_round(rng.standard_normal((m, nown)).astype(np.float32)) for _ in range(2))

    def block(a, rows_alloc):
        blk = Bf16Block(rows_alloc, m)
        pad = np.zeros((m, rows_alloc), dtype=np.float32)
        pad[:, :a.shape[1]] = a
        blk.pack(Vectors(pad), 1.0)
        return blk
    y, p, b = block(y0[:, r0:r1], n_own_pad), block(p0, n_own_pad), block(b0, n_own_pad)
    hgood = block(y0[:, halo], nh)
    hbad = block(np.full((m, len(halo)), np.nan, dtype=np.float32), nh)
    op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01, hbad.ptr(), hbad.ld, part=1)
    op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 0.01, hgood.ptr(), hgood.ld, part=2)
    out = Vectors(n_own_pad, m, data_type=np.float32)
    p.unpack(out)
    A64 = sp.csr_matrix(A[r0:r1]).astype(np.float64)
    t = (A64 @ y0.T.astype(np.float64)).T
    exact = 1.3 * y0[:, r0:r1] + -0.3 * p0 + 0.01 * (b0 - t)
    mag = 1.3 * np.abs(y0[:, r0:r1]) + 0.3 * np.abs(p0) + 0.01 * (np.abs(b0) + (abs(A64) @ np.abs(y0).T.astype(np.float64)).T)
    got = out.data()[:, :nown]
    assert np.all(np.isfinite(got))
    assert np.all(np.abs(got - exact) <= 2.0 ** -8 * np.abs(exact) + 8 * 2.0 ** -24 * mag)


# ---------------------------------------------------------------- interleaved layout: wide rows, every type
@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('m,nv', [(1, 0), (5, 0), (16, 0), (33, 0), (16, 8), (40, 32), (7, 16), (9, 4)])
def test_spmm_wide_rows_fe_like(monkeypatch, key, m, nv):
    """FE-like operator (58 entries per interior row: 8 entry chunks, three column windows per block)
    through the interleaved layout, plain and fused Chebyshev forms, every vectors-per-pass variant
    (RLH_WIDE_NV forces one where the type has it), block sizes that leave a short last pass."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import fe_surrogate
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    if nv:
        monkeypatch.setenv('RLH_WIDE_NV', str(nv))
    A = _sym(fe_surrogate(grid=(9, 11, 13), dof=2), key)
    n = A.shape[0]
    assert np.diff(A.indptr).max() == 58
    op = CsrOperator(A)
    assert op.layout()[0] == 'wide'
    rng = np.random.default_rng(m + nv)
    x = rnd((m, n), key, rng)
    X, Y = Vectors(x), Vectors(n, m, data_type=DT[key])
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
    ref = (A @ x.T).T
    tol = 3e-6 if key in 'sc' else 1e-13
    assert cases.rel(Y.data(), ref) < tol
    p0, b0 = rnd((m, n), key, rng), rnd((m, n), key, rng)
    P, B = Vectors(p0.copy()), Vectors(b0.copy())
    op.cheb_step_ptr(m, X, P, B, 1.3, -0.3, -0.7)
    assert cases.rel(P.data(), 1.3 * x - 0.3 * p0 - 0.7 * (b0 - ref)) < tol
    assert np.array_equal(X.data(), x)


@pytest.mark.parametrize('key', ['d', 's'])
@pytest.mark.parametrize('odd', [False, True])
def test_spmm_wide_row_pairs(monkeypatch, key, odd):
    """Rows that share their column pattern pairwise (the two unknowns of a node) go to one thread as a pair -- one
    position, two values per entry, the staged x values read once for both (wide_spmm_kernel<..., K = 2>): bit for bit the
    result of the one-row-per-thread layout (RLH_WIDE_PAIR=0), plain and fused Chebyshev forms, an odd number of rows
    (the last row has no partner), and a matrix whose rows do NOT pair (one entry removed) takes the other layout."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import fe_surrogate
    monkeypatch.delenv('RLH_SPMM_FORMAT', raising=False)
    A = sp.csr_matrix(fe_surrogate(grid=(9, 11, 13), dof=2).astype(DT[key]))
    if odd:
        A = sp.csr_matrix(A[:-1, :-1])
    n = A.shape[0]
    m = 19
    rng = np.random.default_rng(5)
    x, p0, b0 = rnd((m, n), key, rng), rnd((m, n), key, rng), rnd((m, n), key, rng)
    X, B = Vectors(x), Vectors(b0)
    results = []
    for pair in ('1', '0'):
        monkeypatch.setenv('RLH_WIDE_PAIR', pair)
        op = CsrOperator(A)
        assert op.layout()[0] == 'wide'
        Y, P = Vectors(n, m, data_type=DT[key]), Vectors(p0.copy())
        Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
        op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
        op.cheb_step_ptr(m, X, P, B, 1.3, -0.3, -0.7)
        results.append((Y.data(), P.data()))
    ref = (A.astype(np.float64) @ x.T.astype(np.float64)).T
    assert cases.rel(results[0][0], ref) < (3e-6 if key == 's' else 1e-13)
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    # one entry less in one row: its pair no longer shares the pattern, the layout falls back to one row per thread
    monkeypatch.setenv('RLH_WIDE_PAIR', '1')
    C = sp.lil_matrix(A)
    r = 200
    c = int(A[r].indices[3])
    C[r, c] = 0.0
    C[c, r] = 0.0
    C = sp.csr_matrix(C)
    C.eliminate_zeros()
    opc = CsrOperator(C)
    Y = Vectors(n, m, data_type=DT[key])
    opc.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
    assert cases.rel(Y.data(), (C.astype(np.float64) @ x.T.astype(np.float64)).T) < (3e-6 if key == 's' else 1e-13)


def test_spmm_wide_padding_never_touches_foreign_columns(monkeypatch):
    """A non-finite entry of x in a column a row does not reference must not reach that row
    (padding slots point at the row's own first entry): the reference CSR product only touches
    referenced columns (mkl_wrap.py:246-276)."""
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    for fmt in ('well', 'wide'):
        monkeypatch.setenv('RLH_SPMM_FORMAT', fmt)
        n = 3000
        A = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format='lil')
        A[10, 10:14] = 1.0
        A[10:14, 10] = 1.0
        A = sp.csr_matrix(A)                    # rows of 2 .. 5 entries: every block has padding slots
        x = np.ones((2, n))
        x[:, 0] = np.inf                        # column 0 is referenced by rows 0 and 1 only
        op = CsrOperator(A)
        X, Y = Vectors(x), Vectors(n, 2)
        op.apply_ptr(2, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
        y = Y.data()
        assert not np.any(np.isfinite(y[:, :2])) and np.all(np.isfinite(y[:, 2:]))
        assert np.allclose(y[:, 2:], (A @ np.ones(n))[2:])


def test_config3_surrogate_apply_full_size():
    """BASELINE config 3 stand-in (FE-like surrogate for shipsec5, n = 179 860, ~55
    entries per row): the operator application on the full matrix against the oracle, m = 16 (the
    block size of `which = 10`)."""
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from raleigh_amd.synthetic import fe_surrogate
    A = fe_surrogate()
    n = A.shape[0]
    assert n == 179860 and 54 < A.nnz / n < 57
    rng = np.random.default_rng(3)
    x = rng.standard_normal((16, n))
    op = SparseSymmetricMatrix(A)
    X, Y = Vectors(x), Vectors(n, 16)
    op.apply(X, Y)
    assert cases.rel(Y.data(), ops.csr_sym_apply(sp.triu(A, format='csr'), x)) < 1e-13


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('M,N,m,transp', [(300, 200, 7, False), (1000, 513, 33, True), (4000, 2500, 128, False),
                                           (2500, 6000, 70, True)])
def test_dense_apply_with_rank_one_epilogue(key, M, N, m, transp):
    """rlh_dense_apply_r1: Y = Op(A) X - u c^T with the rank-one term folded into the tile store (no K
    split) or into the split-K reduction (the 4000 x 2500 and 2500 x 6000 cases split K), u given or a
    vector of ones, coefficients produced on the device by a one-column Gram -- the mean shift of the
    PCA operator (raleigh/interfaces/partial_svd.py:258-291)."""
    from raleigh_amd.algebra.hip import Vectors, Matrix
    from raleigh_amd.algebra.hip.matrix import coefficients_into
    from raleigh_amd.algebra.hip.memory import DeviceBuffer
    rng = np.random.default_rng(M + m)
    a = rnd((M, N), key, rng)
    nx, ny = (M, N) if transp else (N, M)
    x = rnd((m, nx), key, rng)
    w = rnd((1, nx), key, rng)
    u = rnd((1, ny), key, rng)
    A, X, W, U = Matrix(a), Vectors(x), Vectors(w), Vectors(u)
    Y = Vectors(ny, m, data_type=DT[key])
    cbuf = DeviceBuffer(m * np.dtype(DT[key]).itemsize, zero=False)
    coefficients_into(cbuf.ptr, X, W)                      # c[j] = w^H x_j, left on the device
    c = (x.astype(np.complex128) @ np.conj(w[0].astype(np.complex128)))
    ref0 = (a.conj().T if transp else a).astype(np.complex128) @ x.T.astype(np.complex128)
    tol = 3e-5 if key in 'sc' else 1e-12
    for uvec, uref in ((U, u[0].astype(np.complex128)), (None, np.ones(ny))):
        A.apply_r1(X, Y, transp, uvec, cbuf.ptr)
        ref = (ref0 - np.outer(uref, c)).T
        assert cases.rel(Y.data(), ref if key in 'cz' else ref.real) < tol
    A.apply(X, Y, transp)
    assert cases.rel(Y.data(), ref0.T if key in 'cz' else ref0.T.real) < tol


@pytest.mark.parametrize('key', ['c', 'z'])
@pytest.mark.parametrize('n,k,m', [(100003, 64, 64), (20000, 16, 16), (5001, 33, 70), (777, 128, 17)])
def test_complex_block_update_on_matrix_cores(monkeypatch, key, n, k, m):
    """The MFMA path of the complex block update (k, m >= 16): multiply, add with a scalar factor, the
    two-source and the two-source / two-output forms, against the oracle and against the VALU kernel
    (RLH_UPDATE_MFMA=0), ragged row tails and coefficient shapes that are not multiples of the tiles."""
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(n + k + m)
    x, x2 = rnd((k, n), key, rng), rnd((k + 3, n), key, rng)
    q, q2 = rnd((k, m), key, rng), rnd((k + 3, m), key, rng)
    w0 = rnd((m, n), key, rng)
    tol = 3e-5 if key == 'c' else 1e-12
    big = np.complex128
    ref_mul = (x.T.astype(big) @ q.astype(big)).T
    ref_add = w0.astype(big) + (0.5 - 0.25j) * ref_mul
    ref_two = ref_mul + (x2.T.astype(big) @ q2.astype(big)).T
    out = {}
    for mfma in ('1', '0'):
        monkeypatch.setenv('RLH_UPDATE_MFMA', mfma)
        X, X2, W = Vectors(x), Vectors(x2), Vectors(n, m, data_type=DT[key])
        X.multiply(q, W)
        r_mul = W.data()
        W.fill(w0)
        W.add(X, 0.5 - 0.25j, q)
        r_add = W.data()
        X.combine(q, X2, q2, W)
        r_two = W.data()
        ma = m // 2
        WA, WB = Vectors(n, ma, data_type=DT[key]), Vectors(n, m - ma, data_type=DT[key])
        X.combine2(q[:, :ma], q[:, ma:], X2, q2[:, :ma], q2[:, ma:], WA, WB)
        r_2x2 = np.concatenate((WA.data(), WB.data()))
        out[mfma] = (r_mul, r_add, r_two, r_2x2)
        for got, ref in ((r_mul, ref_mul), (r_add, ref_add), (r_two, ref_two), (r_2x2, ref_two)):
            assert cases.rel(got, ref) < tol
    for a, b in zip(out['1'], out['0']):
        assert cases.rel(a, b) < tol
