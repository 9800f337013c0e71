"""Randomised parity sweep of the streaming kernels against the oracle (not a pytest file: run on the GPU box as
`python tools/fuzz_parity.py [cases] [seed] [largest n]`): Gram (plain / self / stacked with shared and distinct blocks, windows at
column offsets of wider blocks), block updates (multiply, add, two sources, two results), dots -- random n (incl.
fewer rows than a tile and ragged tails), widths, real types, with the streaming paths forced."""
import os, sys
os.environ['RLH_GRAM_STREAM'] = '2'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.algebra.hip import Vectors
from oracle import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = int(sys.argv[3]) if len(sys.argv) > 3 else 60000          # largest row count drawn
def rel(a, b):
    big_t = np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64
    return float(np.linalg.norm(np.asarray(a, dtype=big_t) - np.asarray(b, dtype=big_t)) / max(np.linalg.norm(b), 1e-300))
worst = {}
def check(tag, got, want, tol, info):
    e = rel(got, want)
    worst[tag] = max(worst.get(tag, 0.0), e)
    if not e < tol:
        print('FAIL', tag, info, 'rel err %.3e' % e)
        sys.exit(1)
def window(x, dt):
    """x (m, n) on the device, either as its own block or as a window of a wider block at a column offset"""
    m, n = x.shape
    if rng.random() < 0.5:
        return Vectors(x.copy())
    off, extra = int(rng.integers(0, 5)), int(rng.integers(0, 4))
    V = Vectors(n, off + m + extra, data_type=dt)
    V.select(m, off)
    V.fill(x.copy())
    return V
for case in range(cases):
    dt = np.float64 if rng.random() < 0.6 else np.float32
    tol = 1e-12 if dt == np.float64 else 3e-4
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 3000), rng.integers(3000, big)]))
    mx, my = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    x = (2 * rng.random((mx, n)) - 1).astype(dt)
    y = (2 * rng.random((my, n)) - 1).astype(dt)
    X, Y = window(x, dt), window(y, dt)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    info = 'case %d dtype %s n %d mx %d my %d' % (case, dt.__name__, n, mx, my)
    check('gram', X.dot(Y), ops.gram(x64, y64), tol, info)
    check('self-gram', X.dot(X), ops.gram(x64, x64), tol, info)
    check('dots', X.dots(X), ops.dots(x64, x64), tol, info)
    if mx <= 32 and my + mx <= 64:
        rb = X.reduction_batch(); rb.gram([X], [Y, X]); g, = rb.run()
        check('stacked-shared', g, ops.gram(x64, np.concatenate((y64, x64))), tol, info)
    z = (2 * rng.random((int(rng.integers(1, 33)), n)) - 1).astype(dt)
    Z = window(z, dt)
    if my + z.shape[0] <= 64:
        rb = X.reduction_batch(); rb.gram([X], [Y, Z]); g, = rb.run()
        check('stacked', g, ops.gram(x64, np.concatenate((y64, z.astype(np.float64)))), tol, info)
    # updates: W = X q, W += alpha X q, [A | B] = X qa + Y qb
    m = int(rng.integers(1, 65))
    q = rng.standard_normal((mx, m)).astype(dt)
    w0 = (2 * rng.random((m, n)) - 1).astype(dt)
    W = window(w0, dt)
    X.multiply(q, W)
    check('multiply', W.data(), ops.multiply(x64, q.astype(np.float64)), tol * 20, info + ' m %d' % m)
    W.fill(w0.copy())
    W.add(X, -0.75, q)
    check('add', W.data(), ops.add_q(w0.astype(np.float64), x64, -0.75, q.astype(np.float64)), tol * 20, info + ' m %d' % m)
    ma, mb = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    qa, qb = rng.standard_normal((mx, ma + mb)).astype(dt), rng.standard_normal((my, ma + mb)).astype(dt)
    A, B = Vectors(n, ma, data_type=dt), Vectors(n, mb, data_type=dt)
    X.combine2(qa[:, :ma], qa[:, ma:], Y, qb[:, :ma], qb[:, ma:], A, B)
    ref = ops.multiply(x64, qa.astype(np.float64)) + ops.multiply(y64, qb.astype(np.float64))
    check('combine2-A', A.data(), ref[:ma], tol * 20, info + ' ma %d mb %d' % (ma, mb))
    check('combine2-B', B.data(), ref[ma:], tol * 20, info + ' ma %d mb %d' % (ma, mb))
    assert np.array_equal(X.data(), x) and np.array_equal(Y.data(), y), info
# complex blocks: quadrant panels (33 .. 64 vectors: 128 x 128 real-view panels and their symmetric self-Gram), stacked
# requests run as one call per left block, matrix-core updates incl. the two-pass two-result form
for case in range(cases // 3):
    dt = np.complex128 if rng.random() < 0.6 else np.complex64
    tol = 1e-12 if dt == np.complex128 else 3e-4
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 3000), rng.integers(3000, min(big, 200000))]))
    mx, my = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    cx = lambda shape: ((2 * rng.random(shape) - 1) + 1j * (2 * rng.random(shape) - 1)).astype(dt)
    x, y = cx((mx, n)), cx((my, n))
    X, Y = window(x, dt), window(y, dt)
    xb, yb = x.astype(np.complex128), y.astype(np.complex128)
    info = 'complex case %d dtype %s n %d mx %d my %d' % (case, dt.__name__, n, mx, my)
    check('c gram', X.dot(Y), ops.gram(xb, yb), tol, info)
    check('c self-gram', X.dot(X), ops.gram(xb, xb), tol, info)
    rb = X.reduction_batch(); rb.gram([X], [Y, X]); g, = rb.run()
    check('c stacked', g, ops.gram(xb, np.concatenate((yb, xb))), tol, info)
    m = int(rng.integers(1, 65))
    q = cx((mx, m))
    w0 = cx((m, n))
    W = window(w0, dt)
    X.multiply(q, W)
    check('c multiply', W.data(), ops.multiply(xb, q.astype(np.complex128)), tol * 20, info + ' m %d' % m)
    W.fill(w0.copy())
    W.add(X, 0.5, q)
    check('c add', W.data(), ops.add_q(w0.astype(np.complex128), xb, 0.5, q.astype(np.complex128)), tol * 20, info + ' m %d' % m)
    ma, mb = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    qa, qb = cx((mx, ma + mb)), cx((my, ma + mb))
    A, B = Vectors(n, ma, data_type=dt), Vectors(n, mb, data_type=dt)
    X.combine2(qa[:, :ma], qa[:, ma:], Y, qb[:, :ma], qb[:, ma:], A, B)
    ref = ops.multiply(xb, qa.astype(np.complex128)) + ops.multiply(yb, qb.astype(np.complex128))
    check('c combine2-A', A.data(), ref[:ma], tol * 20, info + ' ma %d mb %d' % (ma, mb))
    check('c combine2-B', B.data(), ref[ma:], tol * 20, info + ' ma %d mb %d' % (ma, mb))
print('%d cases passed; worst relative errors: %s' % (cases, ', '.join('%s %.1e' % kv for kv in sorted(worst.items()))))
