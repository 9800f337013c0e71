"""The sparse product on the config-3 surrogate (FE-like, n = 179 860, 54.9 entries per row, fp64, 16 vectors): HIP-event median
of 30 calls and 20 calls back to back.  RLH_WIDE_PAIR=0: one row per thread (the round-2/3 kernel)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, CsrOperator
from raleigh_amd.synthetic import fe_surrogate
import scipy.sparse as sp
L = _lib.lib()
dt = {'d': np.float64, 's': np.float32}[sys.argv[1] if len(sys.argv) > 1 else 'd']
A = fe_surrogate().astype(dt)
n, m = A.shape[0], 16
op = CsrOperator(A)
x = np.random.default_rng(0).standard_normal((m, n)).astype(dt)
X, Y = Vectors(x), Vectors(n, m, data_type=dt)
f = lambda: op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())
f(); _lib.check(L.rlh_sync())
ref = (A.astype(np.float64) @ x.T.astype(np.float64)).T
err = np.linalg.norm(Y.data() - ref) / np.linalg.norm(ref)
ms = ctypes.c_float(); ts = []
for _ in range(30):
    _lib.check(L.rlh_timer_start()); f(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
_lib.check(L.rlh_timer_start())
for _ in range(20): f()
_lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
nb = A.nnz * (A.dtype.itemsize + 4) + (n + 1) * 4 + 2 * n * m * A.dtype.itemsize
t = float(np.median(ts))
print('PAIR=%s %s layout %s: %.4f ms (%.1f%% of 8 TB/s), back to back %.4f ms (%.1f%%), error %.1e' %
      (os.environ.get('RLH_WIDE_PAIR', '1'), dt.__name__, op.layout()[0], t, nb / t / 1e6 / 80, ms.value / 20, nb / (ms.value / 20) / 1e6 / 80, err))
