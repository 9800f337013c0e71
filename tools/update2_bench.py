"""The driver's two-source / two-result block update (rlh_block_update2x2: [A | B] = X qx + Y qy, every source read
once by the algorithm) at the roofline point: algorithmic bytes = 2 sources + 2 results = 4 blocks."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
L = _lib.lib()
n, m = 9938375, 32
rng = np.random.default_rng(1)
X, Y, A, B = (Vectors(n, m) for _ in range(4))
X.fill_random(); Y.fill_random()
q = [rng.standard_normal((m, m)) for _ in range(4)]
ms = ctypes.c_float()
def timed(fn, reps=10):
    fn(); _lib.check(L.rlh_sync())
    ts = []
    for _ in range(reps):
        _lib.check(L.rlh_timer_start()); fn(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
    return float(np.median(ts))
Bk = n * m * 8
t = timed(lambda: X.combine2(q[0], q[1], Y, q[2], q[3], A, B))
print('combine2 [A|B] = X qx + Y qy (k = 32 + 32, m = 32 + 32): %.3f ms  %.1f GB/s algorithmic (4 blocks)' % (t, 4 * Bk / t / 1e6))
W = Vectors(n, 2 * m)
X2 = Vectors(n, 2 * m); X2.fill_random()
q64 = rng.standard_normal((2 * m, 2 * m))
t = timed(lambda: X2.multiply(q64, W))
print('multiply m = k = 64: %.3f ms  %.1f GB/s algorithmic (2 blocks of 5.09 GB)' % (t, 4 * Bk / t / 1e6))
