"""Device ILU apply (host ILUT + level-scheduled triangular solves): per-apply time against the
algorithmic bytes (factor entries x (value + 4-byte index) + read B + write X), and where the time of
the config-3 solve goes.  usage: tools/ilu_bench.py [fe|lapN] [--m 16] [--profile]"""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('matrix', nargs='?', default='fe')
ap.add_argument('--m', type=int, default=16)
ap.add_argument('--profile', action='store_true')
ap.add_argument('--solve', action='store_true')
a = ap.parse_args()
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
from raleigh_amd.algebra.hip.precond import IncompleteLU
from raleigh_amd.synthetic import fe_surrogate, lap3d_rows
L = _lib.lib()
if a.matrix == 'fe':
    A = fe_surrogate()
else:
    N = int(a.matrix[3:])
    A = lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
n, m = A.shape[0], a.m
t0 = time.time()
T = IncompleteLU(A)
T.factorize()
print('%s n=%d nnz=%d: ILUT %.2f s, fill %.2f, levels %s' % (a.matrix, n, A.nnz, time.time() - t0, T.fill, T.levels))
B, X = Vectors(n, m), Vectors(n, m)
B.fill_random()
ms = ctypes.c_float()
T.apply(B, X)
_lib.check(L.rlh_sync())
ts = []
for _ in range(5):
    _lib.check(L.rlh_timer_start())
    T.apply(B, X)
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    ts.append(ms.value)
t = float(np.median(ts))
nb = T.chain().algorithmic_bytes(m)
print('ilu apply m=%d: %.3f ms  %.1f GB/s algorithmic (%.2f%% of 8 TB/s)' % (m, t, nb / t / 1e6, nb / t / 1e6 / 80))
t0 = time.perf_counter()
for _ in range(5):
    T.apply(B, X)
_lib.check(L.rlh_sync())
print('host wall per apply: %.3f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
if a.solve or a.profile:
    from raleigh_amd.interfaces import partial_hevp
    np.random.seed(1)
    if a.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    t0 = time.time()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    print('solve: %.2f s, status %d, %d iterations' % (time.time() - t0, status, partial_hevp.last['iterations']))
    if a.profile:
        pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(18)
