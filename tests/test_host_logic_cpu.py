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
