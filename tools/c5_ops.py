"""Times of the block operations block MINRES issues at BASELINE config 5's size (complex128, n = 126^3, 64 vectors)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.algebra.hip import Vectors, synchronize, SparseSymmetricMatrix
from raleigh_amd.synthetic import hermitian_lap3d_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 126
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dt = np.complex128
n = N ** 3
H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n)
A = SparseSymmetricMatrix(H)
print('layout', A.layout())
X, Y, Z, W = (Vectors(n, m, data_type=dt) for _ in range(4))
np.random.seed(1)
for v in (X, Y, Z, W):
    v.fill_random()
rng = np.random.default_rng(0)
q = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))) / m
B = n * m * 16
def bench(name, f, nbytes, reps=10):
    f(); synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    synchronize()
    ms = 1e3 * (time.perf_counter() - t) / reps
    print('%-34s %7.3f ms  %6.2f TB/s' % (name, ms, nbytes / ms / 1e9))
bench('spmm', lambda: A.apply(X, Y), 2 * B + 14 * n * 20)
bench('cheb_step', lambda: A.cheb_step(X, Y, Z, 1.0, -0.5, 0.25), 4 * B + 14 * n * 20)
bench('add scalar (axpy)', lambda: Y.add(X, -0.5), 3 * B)
bench('add q', lambda: Y.add(X, -1.0, q), 3 * B)
bench('multiply', lambda: X.multiply(q, Y), 2 * B)
bench('dot (2 blocks)', lambda: X.dot(Y), 2 * B)
bench('dot (self)', lambda: X.dot(X), B)
bench('lincomb', lambda: Y.lincomb(0.5, X, 0.0, X), 2 * B)
bench('lincomb 2 src', lambda: Y.lincomb(0.5, X, 0.25, Z), 3 * B)
bench('zero', lambda: Y.zero(), B)
bench('copy', lambda: X.copy(Y), 2 * B)
bench('combine', lambda: X.combine(q, Z, q, Y), 3 * B)
bench('combine2', lambda: X.combine2(q, q, Z, q, q, Y, W), 4 * B)
