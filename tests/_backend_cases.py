"""Backend parity cases shared by the CPU tier (host logic over tests/fake_lib.py)
and the GPU tier (the real librlhip.so).  They read like the reference's own
tests/tests_algebra.py and tests/tests_matrix.py: run every Vectors / Matrix
operation and compare with the golden vectors produced by the reference."""

import os

import numpy as np
import scipy.sparse as sp

TOL = {'s': 2e-5, 'c': 2e-5, 'd': 1e-13, 'z': 1e-13}
DT = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}


def rel(a, b):
    den = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (den if den > 0 else 1.0)


def ops_case(golden_dir, key, shape):
    from raleigh_amd.algebra.hip import Vectors
    m, n = shape
    g = np.load(os.path.join(golden_dir, 'ops_%s_%dx%d.npz' % (key, m, n)))
    tol = TOL[key]
    u, v = Vectors(g['u'].copy()), Vectors(g['v'].copy())
    assert u.dimension() == n and u.nvec() == m and u.data_type() == DT[key]
    assert u.is_complex() == (key in 'cz')
    assert np.array_equal(u.data(), g['u'])
    # dots / dot
    d = u.dots(v)
    assert d.dtype == DT[key] and d.shape == (m,)
    assert rel(d, g['dots']) < tol
    assert rel(u.dots(v, transp=True), g['dots_transp']) < tol
    p = u.dot(v)
    assert p.shape == (m, m) and p.dtype == DT[key]
    assert rel(p, g['dot']) < tol
    u.select(m - 2, 1)
    v.select(m - 2, 2)
    assert u.selected() == (1, m - 2)
    assert rel(u.dot(v), g['dot_window']) < tol
    u.select(m)
    v.select(m)
    # self-Gram path (same window read once) against the two-operand path
    uu = u.dot(u)
    u2 = u.clone()
    assert rel(uu, u.dot(u2)) < tol
    assert rel(uu, uu.conj().T) < tol
    # multiply: C- and F-ordered q, rectangular q into a window
    q = g['q']
    w = Vectors(g['v'].copy())
    u.multiply(q, w)
    assert rel(w.data(), g['multiply']) < tol
    w = Vectors(g['v'].copy())
    u.multiply(np.asfortranarray(q), w)
    assert rel(w.data(), g['multiply_F']) < tol
    w = Vectors(g['v'].copy())
    w.select(3, 1)
    u.multiply(q[:, :3].copy(), w)
    assert rel(w.data(), g['multiply_rect']) < tol
    w.select(m)
    assert np.array_equal(w.data()[4:], g['v'][4:]) and np.array_equal(w.data()[0], g['v'][0])
    # add: scalar, vector, q (C and F)
    w = Vectors(g['v'].copy())
    w.add(u, -0.75)
    assert rel(w.data(), g['add_scalar']) < tol
    w = Vectors(g['v'].copy())
    w.add(u, g['s_vec'])
    assert rel(w.data(), g['add_vector']) < tol
    w = Vectors(g['v'].copy())
    w.add(u, 2.0, q)
    assert rel(w.data(), g['add_q']) < tol
    w = Vectors(g['v'].copy())
    w.add(u, -1.0, np.asfortranarray(q))
    assert rel(w.data(), g['add_q_F']) < tol
    # the reference's own check: v - u*p = 0 after multiply (tests_algebra.py:304-328)
    w = Vectors(g['v'].copy())
    u.multiply(p, w)
    w.add(u, -1.0, p)
    assert np.linalg.norm(w.data()) < 10 * tol * np.linalg.norm(g['multiply'])
    # fused forms: combine == multiply + add, lincomb == copy + add (checked against the golden add results)
    w = Vectors(g['v'].copy())
    w2 = Vectors(g['v'].copy())
    u.multiply(q, w2)
    w2.add(v, 1.0, np.asfortranarray(q))
    u.combine(q, v, np.asfortranarray(q), w)
    assert rel(w.data(), w2.data()) < tol
    w = Vectors(g['v'].copy())
    w.lincomb(1.0, v, g['s_vec'], u)
    assert rel(w.data(), g['add_vector']) < tol
    w = Vectors(g['v'].copy())
    w.lincomb(1.0, w, -0.75, u)            # aliasing the output with an input
    assert rel(w.data(), g['add_scalar']) < tol
    # scale
    w = Vectors(g['u'].copy())
    w.scale(g['scale_s'], multiply=True)
    assert rel(w.data(), g['scale_mul']) < tol
    w = Vectors(g['u'].copy())
    w.scale(g['scale_s'])
    assert rel(w.data(), g['scale_div']) < tol
    # copy: window and cyclic-shift gather (tests_algebra.py:133-164)
    w = Vectors(g['v'].copy())
    u.copy(w)
    assert np.array_equal(w.data(), g['u'])
    w = Vectors(g['v'].copy())
    u.copy(w, g['ind'])
    assert np.array_equal(w.data(), g['copy_ind'])
    w = Vectors(g['v'].copy())
    w.select(2, 1)
    u.copy(w, g['ind'][:2])
    w.select(m)
    assert np.array_equal(w.data(), g['copy_ind_window'])
    # orthogonalize
    w = Vectors(g['v'].copy())
    qq = w.orthogonalize(u)
    assert rel(w.data(), g['orth']) < 20 * tol
    assert rel(qq.data(), g['orth_q']) < tol
    # svd: singular values + the reference's reconstruction check (tests_algebra.py:330-341)
    w = Vectors(g['u'].copy())
    sigma, qs = w.svd()
    assert rel(sigma, g['svd_sigma']) < 50 * tol
    gw = w.dot(w)
    assert np.linalg.norm(gw - np.eye(m)) < 100 * tol * m
    w.scale(sigma, multiply=True)
    t = Vectors(g['v'].copy())
    w.multiply(qs.conj().T if key in 'cz' else qs.T, t)
    assert rel(t.data(), g['u']) < 100 * tol
    # bookkeeping: clone / reference / zero / fill / append / new_vectors
    u.select(3, 2)
    c = u.clone()
    assert c.nvec() == 3 and c.selected() == (0, 3) and np.array_equal(c.data(), g['u'][2:5])
    r = u.reference()
    r.zero()
    u.select(m)
    z = u.data()
    assert np.all(z[2:5] == 0) and np.array_equal(z[:2], g['u'][:2]) and np.array_equal(z[5:], g['u'][5:])
    u.select(2, 0)
    u.fill(g['v'][:2].copy())
    u.select(m)
    assert np.array_equal(u.data()[:2], g['v'][:2])
    e = Vectors(n, data_type=DT[key])
    assert e.nvec() == 0 and e.dimension() == n
    v.select(3, 1)
    e.append(v)
    v.select(2, 0)
    e.append(v)
    assert e.nvec() == 5 and np.array_equal(e.data(), np.concatenate((g['v'][1:4], g['v'][:2])))
    for _ in range(5):      # capacity growth keeps the old vectors
        v.select(m)
        e.append(v)
    assert e.nvec() == 5 + 5 * m and np.array_equal(e.data()[:3], g['v'][1:4])
    nv = u.new_vectors(4)
    assert nv.nvec() == 4 and nv.dimension() == n and np.all(nv.data() == 0)
    assert u.new_vectors(2, 11).dimension() == 11
    # append(axis=1) (tests_algebra.py:432-466)
    a1 = Vectors(g['u'].copy())
    a1.append(Vectors(g['v'].copy()), axis=1)
    assert a1.dimension() == 2 * n and np.array_equal(a1.data(), np.concatenate((g['u'], g['v']), axis=1))
    # fill_random: U(-1,1) from the host RNG, same stream as the reference (dense_ndarray.py:34-37)
    np.random.seed(7)
    f = Vectors(n, 3, data_type=DT[key])
    f.fill_random()
    np.random.seed(7)
    expect = (2 * np.random.rand(3, n) - 1).astype(DT[key])
    assert rel(f.data(), expect) < 1e-6


def matrix_case(golden_dir, key):
    from raleigh_amd.algebra.hip import Vectors, Matrix
    g = np.load(os.path.join(golden_dir, 'matrix_%s.npz' % key))
    a, x, z = g['a'], g['x'], g['z']
    tol = 20 * TOL[key]
    k = x.shape[0]
    for tag, arr in (('C', np.ascontiguousarray(a)), ('F', np.asfortranarray(a))):
        A = Matrix(arr)
        assert A.order() == tag + '_CONTIGUOUS' and A.shape() == a.shape and A.data_type() == DT[key]
        vx, vy = Vectors(x.copy()), Vectors(a.shape[0], k, data_type=DT[key])
        A.apply(vx, vy)
        assert rel(vy.data(), g['apply_' + tag]) < tol
        assert np.array_equal(vx.data(), x)        # x restored (the reference conjugates it in place)
        vz, vw = Vectors(z.copy()), Vectors(a.shape[1], k, data_type=DT[key])
        A.apply(vz, vw, transp=True)
        assert rel(vw.data(), g['apply_t_' + tag]) < tol
        A.apply(vy, vw, transp=True)
        assert rel(vw.data(), g['ata_' + tag]) < tol
        try:
            A.apply(vx, vw)
            raise AssertionError('dimension mismatch not detected')
        except ValueError:
            pass
    # Vectors view of a C-ordered Matrix (AMatrix.as_vectors) and Matrix.dots
    A = Matrix(np.ascontiguousarray(a))
    rows = Vectors(A, shallow=True)
    assert rows.nvec() == a.shape[0] and rows.dimension() == a.shape[1]
    assert np.array_equal(rows.data(), a)
    from oracle import ops
    assert rel(A.dots(), ops.dots(a, a)) < tol


def sparse_case(golden_dir):
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    A = sp.csr_matrix(g['lap_dense'])
    op = SparseSymmetricMatrix(A)
    assert op.size() == A.shape[0]
    x = Vectors(g['lap_x'].copy())
    y = Vectors(A.shape[0], 3)
    op.apply(x, y)
    assert rel(y.data(), g['lap_y']) < 1e-13
    x1, y1 = Vectors(g['lap_x'][:1].copy()), Vectors(A.shape[0], 1)
    op.apply(x1, y1)
    assert rel(y1.data(), g['lap_y1']) < 1e-13
    # only the upper triangle defines the operator (mkl 'SUNF'): garbage below is ignored
    B = sp.csr_matrix(sp.triu(A) + 5.0 * sp.tril(A, k=-1))
    opb = SparseSymmetricMatrix(B)
    opb.apply(x, y)
    assert rel(y.data(), g['lap_y']) < 1e-13
    # ... and so is a lower triangle with an entry missing (both triangles given and CONSISTENT is the one case in which the
    # library takes the matrix as it comes instead of mirroring its upper triangle)
    C = sp.lil_matrix(A)
    C[5, 4] = 0.0
    C = sp.csr_matrix(C)
    C.eliminate_zeros()
    opc = SparseSymmetricMatrix(C)
    opc.apply(x, y)
    assert rel(y.data(), g['lap_y']) < 1e-13
    H = sp.csr_matrix(g['herm_dense'])
    Hbad = sp.lil_matrix(H)
    i, j = sp.tril(H, k=-1).nonzero()
    Hbad[i[0], j[0]] = H[i[0], j[0]].conjugate()         # a lower entry that is NOT the conjugate of its mirror image
    opzb = SparseSymmetricMatrix(sp.csr_matrix(Hbad))
    xz0 = Vectors(g['herm_x'].copy())
    yz0 = Vectors(H.shape[0], 3, data_type=np.complex128)
    opzb.apply(xz0, yz0)
    assert rel(yz0.data(), g['herm_y']) < 1e-13
    opz = SparseSymmetricMatrix(H)
    xz = Vectors(g['herm_x'].copy())
    yz = Vectors(H.shape[0], 3, data_type=np.complex128)
    opz.apply(xz, yz)
    assert rel(yz.data(), g['herm_y']) < 1e-13
    # windows: apply to vectors 1..2 into vectors 0..1
    x.select(2, 1)
    y.zero()
    y.select(2, 0)
    op.apply(x, y)
    y.select(3)
    out = y.data()
    assert rel(out[:2], g['lap_y'][1:3]) < 1e-13 and np.all(out[2] == 0)
