"""Time of the complex128 64 x 64 two-operand Gram at config 5's size: python tools/zgram_bench.py [N] (RLH_GRAM_ZDMA, RLH_GRAM_ZDBG)"""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, synchronize
N = int(sys.argv[1]) if len(sys.argv) > 1 else 126
n, m = N ** 3, 64
L = _lib.lib()
X, Y = Vectors(n, m, data_type=np.complex128), Vectors(n, m, data_type=np.complex128)
np.random.seed(1); X.fill_random(); Y.fill_random()
code = _lib.dtype_code(np.complex128)
res = ctypes.c_void_p(); _lib.check(L.rlh_malloc(ctypes.byref(res), m * m * 16))
f = lambda: _lib.check(L.rlh_gram(code, n, m, X.data_ptr(), X.ld(), m, Y.data_ptr(), Y.ld(), res, None))
ms = ctypes.c_float()
for _ in range(3): f()
_lib.check(L.rlh_sync())
ts = []
for _ in range(20):
    _lib.check(L.rlh_timer_start()); f(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
t = float(np.median(ts))
print('ZDMA=%s ZDBG=%s: %.3f ms  %.1f TF  %.2f TB/s' % (os.environ.get('RLH_GRAM_ZDMA', '1'), os.environ.get('RLH_GRAM_ZDBG', '0'), t, 8.0 * n * m * m / t / 1e9, 2 * n * m * 16 / t / 1e9))
