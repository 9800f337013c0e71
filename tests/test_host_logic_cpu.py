"""CPU tier: the host side of raleigh_amd (selection windows, padded leading
dimensions, strides, growth, error behaviour) driven through tests/fake_lib.py,
which implements the C ABI on host memory with the oracle.  Same cases as the
GPU parity tier (tests/test_hip_parity_gpu.py)."""

import numpy as np
import pytest

import _backend_cases as cases
import fake_lib


@pytest.fixture(autouse=True)
def fake():
    f = fake_lib.install()
    yield f
    fake_lib.uninstall()


@pytest.mark.parametrize('key', ['s', 'd', 'c', 'z'])
@pytest.mark.parametrize('shape', [(5, 257), (16, 192)])
def test_ops(golden_dir, key, shape):
    cases.ops_case(golden_dir, key, shape)


@pytest.mark.parametrize('key', ['s', 'd', 'c', 'z'])
def test_matrix(golden_dir, key):
    cases.matrix_case(golden_dir, key)


def test_sparse(golden_dir):
    cases.sparse_case(golden_dir)


def test_errors(fake):
    from raleigh_amd.algebra.hip import Vectors, Matrix
    with pytest.raises(ValueError):
        Vectors('nonsense')
    with pytest.raises(ValueError):
        Vectors(10, 2, data_type=np.int32)
    with pytest.raises(ValueError):
        Matrix(np.zeros((4, 6))[:, ::2])
    u = Vectors(np.zeros((3, 10)))
    w = Vectors(np.zeros((2, 10)))
    with pytest.raises(ValueError):        # 3 x 3 coefficients cannot map 3 vectors onto 2
        u._update(np.zeros((3, 3)), u, w, 1.0, 0)
    with pytest.raises(ValueError):
        u.combine(np.zeros((3, 2)), w, np.zeros((3, 2)), w.new_vectors(2))
    with pytest.raises(ValueError):
        u.fill(np.zeros((3, 9)))
    with pytest.raises(ValueError):
        u.append(Vectors(np.zeros((2, 10))), axis=1)


def test_missing_library_fails_loudly(monkeypatch):
    """No CPU fallback: without librlhip.so the backend raises."""
    from raleigh_amd import _lib
    fake_lib.uninstall()
    monkeypatch.setattr(_lib, 'LIBPATH', '/nonexistent/librlhip.so')
    from raleigh_amd.algebra.hip import Vectors
    with pytest.raises(_lib.RlhError):
        Vectors(10, 2)


def test_fill_random_device_path(fake, monkeypatch):
    """Large blocks take the library generator (rlh_fill_random) seeded from the numpy stream:
    reproducible under numpy.random.seed, uniform in [-1, 1), small blocks keep the reference's
    host draw (same numbers as numpy.random.rand)."""
    from raleigh_amd.algebra.hip import Vectors
    from oracle import ops
    v = Vectors(1000, 3)
    np.random.seed(7)
    v.fill_random()
    np.random.seed(7)
    assert np.array_equal(v.data(), 2 * np.random.rand(3, 1000) - 1)       # host path, as the reference
    assert fake.calls.get('fill_random', 0) == 0
    monkeypatch.setattr(Vectors, 'DEVICE_RANDOM_THRESHOLD', 2000)
    np.random.seed(7)
    v.fill_random()
    first = v.data()
    assert fake.calls.get('fill_random', 0) == 1
    np.random.seed(7)
    seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
    assert np.array_equal(first, ops.uniform_block(seed, 1000, 3, np.float64))
    v.fill_random()                                    # the next draw of the stream: another block
    assert not np.array_equal(v.data(), first)
    assert first.min() >= -1 and first.max() < 1 and abs(first.mean()) < 0.05
    c = Vectors(1000, 3, data_type=np.complex64)
    c.fill_random()
    assert np.all(c.data().imag == 0) and c.data().real.std() > 0.5


def test_combine2_host_logic(fake):
    """Vectors.combine2 stacks the coefficient matrices and makes ONE library call."""
    from raleigh_amd.algebra.hip import Vectors
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal((4, 300)), rng.standard_normal((3, 300))
    qxa, qxb, qya, qyb = (rng.standard_normal(s) for s in ((4, 2), (4, 5), (3, 2), (3, 5)))
    A, B = Vectors(300, 2), Vectors(300, 5)
    Vectors(x).combine2(qxa, qxb, Vectors(y), qya, qyb, A, B)
    assert fake.calls.get('block_update2x2', 0) == 1
    assert np.allclose(A.data(), qxa.T @ x + qya.T @ y) and np.allclose(B.data(), qxb.T @ x + qyb.T @ y)
    with pytest.raises(ValueError):
        Vectors(x).combine2(qxa, qxb, Vectors(y), qya, qyb, B, A)


def test_reduction_batch_rejects_foreign_blocks(fake):
    """a batch reads every block with the prototype's type and length; anything else is refused."""
    import numpy as np
    import pytest
    from raleigh_amd.algebra.hip import Vectors
    x, y = Vectors(100, 3), Vectors(100, 2)
    rb = x.reduction_batch()
    rb.gram([x], [y])
    with pytest.raises(ValueError):
        rb.gram([x], [Vectors(90, 2)])
    with pytest.raises(ValueError):
        rb.gram([Vectors(100, 2, data_type=np.float32)], [x])
    with pytest.raises(ValueError):
        rb.dots(x, y)
    with pytest.raises(ValueError):
        rb.dots(x, Vectors(100, 3, data_type=np.complex128))
