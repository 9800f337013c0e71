"""Direct shift-invert on the device: host factorisation once (the library's L D L^H, or SuperLU with --method superlu),
solves = persistent triangular launches on the block: factorisation seconds, factor entries on the device, per-apply time
of lap3d N^3 with `--m` vectors.  usage: tools/si_bench.py [N] [--m 8] [--complex] [--method ldlt|superlu|both]"""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('N', nargs='?', type=int, default=30)
ap.add_argument('--m', type=int, default=8)
ap.add_argument('--complex', action='store_true')
ap.add_argument('--method', default='both')
a = ap.parse_args()
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
from raleigh_amd.synthetic import lap3d_rows
import scipy.sparse as sp
L = _lib.lib()
N = a.N
A = lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
dt = np.float64
if a.complex:
    n = A.shape[0]
    S = sp.diags([np.full(n - 1, 0.3)], [1])
    A = sp.csr_matrix(A.astype(np.complex128) + 1j * S - 1j * S.T)
    dt = np.complex128
n, m = A.shape[0], a.m
for method in (('ldlt', 'superlu') if a.method == 'both' else (a.method,)):
    t0 = time.time()
    solver = SparseSymmetricSolver(dtype=dt, method=method)
    solver.analyse(A, sigma=40.0)
    solver.factorize()
    t1 = time.time()
    chain = solver._device_chain()
    _lib.check(L.rlh_sync())
    t2 = time.time()
    extra = ''
    if method == 'ldlt':
        extra = ' %s' % {k: v for k, v in solver.factors().info.items() if k in ('two_by_two', 'delayed', 'max_front', 'supernodes')}
    try:
        inertia = solver.inertia()
    except RuntimeError as e:
        inertia = 'refused'
    print('lap3d %d^3 %s %s: factorisation %.2f s, device operators %.2f s, factor entries %s, levels %s, inertia %s%s'
          % (N, np.dtype(dt).name, method, t1 - t0, t2 - t1, chain.nnz, chain.levels, inertia, extra))
    B, X = Vectors(n, m, data_type=dt), Vectors(n, m, data_type=dt)
    B.fill_random()
    ms = ctypes.c_float()
    solver.solve(B, X)
    _lib.check(L.rlh_sync())
    ts = []
    for _ in range(5):
        _lib.check(L.rlh_timer_start())
        solver.solve(B, X)
        _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
        ts.append(ms.value)
    t = float(np.median(ts))
    es = np.dtype(dt).itemsize
    nb = sum(chain.nnz) * (es + 4) + 2 * n * m * es
    print('  apply m=%d: %.3f ms  %.1f GB/s algorithmic' % (m, t, nb / t / 1e6))
    r = (A - 40.0 * sp.identity(n)) @ X.data().T - B.data().T
    print('  residual %.2e' % (np.linalg.norm(r) / np.linalg.norm(B.data())))
