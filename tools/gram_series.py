"""Per-call HIP-event times of rlh_gram / rlh_dots at the roofline block size (are slow calls systematic?).
usage: tools/gram_series.py [m] [reps]"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
L = _lib.lib()
n = 9938375
m = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
X, Y = Vectors(n, m), Vectors(n, m)
X.fill_random(); Y.fill_random()
res = ctypes.c_void_p(); _lib.check(L.rlh_malloc(ctypes.byref(res), m * m * 8))
ms = ctypes.c_float()
def run(fn):
    fn(); _lib.check(L.rlh_sync())
    ts = []
    for _ in range(reps):
        _lib.check(L.rlh_timer_start()); fn(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
    return ts
g = lambda: L.rlh_gram(1, n, m, X.data_ptr(), X.ld(), m, Y.data_ptr(), Y.ld(), res, None)
gs = lambda: L.rlh_gram(1, n, m, X.data_ptr(), X.ld(), m, X.data_ptr(), X.ld(), res, None)
d = lambda: L.rlh_dots(1, n, m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), res, None)
for name, fn in (('gram X.dot(Y)', g), ('gram X.dot(X)', gs), ('dots X.dots(Y)', d)):
    ts = run(fn)
    print('%-16s m=%d each: %s' % (name, m, ' '.join('%.3f' % t for t in ts)))
    _lib.check(L.rlh_timer_start())
    for _ in range(20): fn()
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    print('%-16s 20 back to back: %.3f ms each' % (name, ms.value / 20))
