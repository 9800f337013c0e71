"""Pins the CPU oracle to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  fp64/complex128: 1e-13 rel; fp32/complex64: 1e-5
rel."""

import json
import os

import numpy as np
import pytest

from oracle import ops
from oracle import Vectors, Matrix, SparseSymmetricMatrix, lap3d
from oracle.sparse import lap3d_eigenvalues

KEYS = ['s', 'd', 'c', 'z']
SHAPES = [(5, 257), (16, 192)]
TOL = {'s': 1e-5, 'c': 1e-5, 'd': 1e-13, 'z': 1e-13}


def rel(a, b):
    den = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - b) / (den if den > 0 else 1.0)


@pytest.mark.parametrize('key', KEYS)
@pytest.mark.parametrize('shape', SHAPES)
def test_ops_functional(golden_dir, key, shape):
    m, n = shape
    g = np.load(os.path.join(golden_dir, 'ops_%s_%dx%d.npz' % (key, m, n)))
    u, v, q, tol = g['u'], g['v'], g['q'], TOL[key]
    assert rel(ops.dots(u, v), g['dots']) < tol
    assert rel(ops.dots_transp(u, v), g['dots_transp']) < tol
    assert rel(ops.gram(u, v), g['dot']) < tol
    assert ops.gram(u, v).shape == (m, m)
    assert rel(ops.gram(u[1:m - 1], v[2:]), g['dot_window']) < tol
    assert rel(ops.multiply(u, q), g['multiply']) < tol
    assert rel(ops.multiply(u, np.asfortranarray(q)), g['multiply_F']) < tol
    assert rel(ops.multiply(u, q[:, :3]), g['multiply_rect']) < tol
    assert rel(ops.axpy(v, u, -0.75), g['add_scalar']) < tol
    assert rel(ops.axpy_cols(v, u, g['s_vec']), g['add_vector']) < tol
    assert rel(ops.add_q(v, u, 2.0, q), g['add_q']) < tol
    assert rel(ops.add_q(v, u, -1.0, np.asfortranarray(q)), g['add_q_F']) < tol
    assert rel(ops.scale_cols(u, g['scale_s'], True), g['scale_mul']) < tol
    assert rel(ops.scale_cols(u, g['scale_s'], False), g['scale_div']) < tol
    assert np.array_equal(ops.copy_cols(u, g['ind']), g['copy_ind'])
    new, qq = ops.orthogonalize(v, u)
    assert rel(new, g['orth']) < 10 * tol
    assert rel(qq, g['orth_q']) < tol
    w, sigma, vh = ops.svd(u)
    assert rel(sigma, g['svd_sigma']) < tol
    # reconstruction u = conj(vh)... in the reference's convention u = v diag(s) w
    assert rel((vh.conj() * sigma) @ w, u) < 10 * tol


@pytest.mark.parametrize('key', KEYS)
def test_vectors_class(golden_dir, key):
    """Same calls as the reference's tests_algebra.py, through the class API."""
    m, n = 16, 192
    g = np.load(os.path.join(golden_dir, 'ops_%s_%dx%d.npz' % (key, m, n)))
    tol = TOL[key]
    u, v = Vectors(g['u'].copy()), Vectors(g['v'].copy())
    assert rel(u.dots(v), g['dots']) < tol
    assert rel(u.dot(v), g['dot']) < tol
    u.select(m - 2, 1)
    v.select(m - 2, 2)
    assert rel(u.dot(v), g['dot_window']) < tol
    u.select(m)
    v.select(m)
    w = Vectors(g['v'].copy())
    u.multiply(g['q'], w)
    assert rel(w.data(), g['multiply']) < tol
    w = Vectors(g['v'].copy())
    w.select(3, 1)
    u.multiply(g['q'][:, :3].copy(), w)
    assert rel(w.data(), g['multiply_rect']) < tol
    w.select(m)
    assert np.array_equal(w.data()[4:], g['v'][4:])
    w = Vectors(g['v'].copy())
    w.add(u, 2.0, g['q'])
    assert rel(w.data(), g['add_q']) < tol
    w = Vectors(g['v'].copy())
    w.add(u, g['s_vec'])
    assert rel(w.data(), g['add_vector']) < tol
    w = Vectors(g['v'].copy())
    w.add(u, -0.75)
    assert rel(w.data(), g['add_scalar']) < tol
    w = Vectors(g['u'].copy())
    w.scale(g['scale_s'])
    assert rel(w.data(), g['scale_div']) < tol
    w = Vectors(g['v'].copy())
    u.copy(w, g['ind'])
    assert np.array_equal(w.data(), g['copy_ind'])
    w = Vectors(g['v'].copy())
    w.select(2, 1)
    u.copy(w, g['ind'][:2])
    w.select(m)
    assert np.array_equal(w.data(), g['copy_ind_window'])
    # append / clone / new_vectors bookkeeping (dense_ndarray.py:39-47)
    e = Vectors(n, data_type=u.data_type())
    assert e.nvec() == 0 and e.dimension() == n
    u.select(3, 2)
    e.append(u)
    assert e.nvec() == 3 and np.array_equal(e.data(), g['u'][2:5])
    c = u.clone()
    assert c.nvec() == 3 and c.selected() == (0, 3)


@pytest.mark.parametrize('key', KEYS)
def test_matrix_apply(golden_dir, key):
    g = np.load(os.path.join(golden_dir, 'matrix_%s.npz' % key))
    a, x, z, tol = g['a'], g['x'], g['z'], 10 * TOL[key]
    for tag, arr in (('C', np.ascontiguousarray(a)), ('F', np.asfortranarray(a))):
        A = Matrix(arr)
        assert A.order() == tag + '_CONTIGUOUS'
        y = Vectors(np.zeros((x.shape[0], a.shape[0]), dtype=a.dtype))
        A.apply(Vectors(x.copy()), y)
        assert rel(y.data(), g['apply_' + tag]) < tol
        w = Vectors(np.zeros((x.shape[0], a.shape[1]), dtype=a.dtype))
        A.apply(Vectors(z.copy()), w, transp=True)
        assert rel(w.data(), g['apply_t_' + tag]) < tol
        A.apply(y, w, transp=True)
        assert rel(w.data(), g['ata_' + tag]) < tol


def test_sparse_apply(golden_dir):
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    A = lap3d(6, 5, 4, 1.0, 1.01, 1.02)
    assert rel(A.toarray(), g['lap_dense']) < 1e-15
    op = SparseSymmetricMatrix(A)
    y = np.zeros_like(g['lap_x'])
    op.apply(g['lap_x'], y)
    assert rel(y, g['lap_y']) < 1e-13
    y1 = np.zeros_like(g['lap_y1'])
    op.apply(g['lap_x'][:1].copy(), y1)
    assert rel(y1, g['lap_y1']) < 1e-13
    import scipy.sparse as sp
    opz = SparseSymmetricMatrix(sp.csr_matrix(g['herm_dense']))
    yz = np.zeros_like(g['herm_x'])
    opz.apply(g['herm_x'], yz)
    assert rel(yz, g['herm_y']) < 1e-13


def test_lap3d_analytic_vs_reference_solver(golden_dir):
    """The reference's own converged eigenvalues (config 1 of BASELINE.json)
    agree with the analytic spectrum used as the full-size parity property."""
    k = json.load(open(os.path.join(golden_dir, 'known_answers.json')))
    ref = np.array(k['hevp_lap30_si6']['eigenvalues'])
    ana = lap3d_eigenvalues(30, 30, 30, 1.0, 1.01, 1.02, 6)
    assert np.max(np.abs(ref - ana) / ana) < 1e-10
    ref10 = np.array(k['hevp_lap30_ilu10']['eigenvalues'])
    ana10 = lap3d_eigenvalues(30, 30, 30, 1.0, 1.01, 1.02, 10)
    assert np.max(np.abs(ref10 - ana10) / ana10) < 1e-8


def test_uniform_block_generator():
    """The counter-based generator behind rlh_fill_random: vector 0 of seed s is the splitmix64
    stream of s (published test vector for s = 1234567), rows / vectors can be generated in
    shards, values are uniform in [-1, 1)."""
    from oracle import ops
    z = [6457827717110365317, 3203168211198807973, 9817491932198370423]
    want = np.array([(v >> 11) * 2.0 ** -53 * 2 - 1 for v in z])
    assert np.array_equal(ops.uniform_block(1234567, 3, 1, np.float64)[0], want)
    full = ops.uniform_block(99, 700, 5, np.float64)
    assert np.array_equal(ops.uniform_block(99, 300, 5, np.float64, row0=250), full[:, 250:550])
    assert np.array_equal(ops.uniform_block(99, 700, 2, np.float64, col0=3), full[3:])
    assert full.min() >= -1 and full.max() < 1 and abs(full.mean()) < 0.03 and abs(full.std() - 3 ** -0.5) < 0.02
    f32 = ops.uniform_block(99, 700, 5, np.float32)
    assert f32.dtype == np.float32 and np.max(np.abs(f32 - full)) < 2e-7
    c = ops.uniform_block(99, 700, 5, np.complex128)
    assert np.array_equal(c.real, full) and np.all(c.imag == 0)
