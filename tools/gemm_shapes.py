"""fp32 dense apply pair (A x, A^T y, m = 128) at several shapes, matrices filled on the device: TFLOP/s per
product, a short and a long timed loop (clock ramp)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, Matrix
L = _lib.lib()
m = 128
ms = ctypes.c_float()
shapes = [(20000, 20000), (40000, 20000), (20000, 40000), (40000, 40000), (62500, 40000), (20096, 20000), (24576, 20000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (M, N) in shapes:
    rows = Vectors(N, M, data_type=np.float32)
    rows.fill_random()
    A = Matrix(rows)
    x = Vectors(N, m, data_type=np.float32); x.fill_random()
    y = Vectors(M, m, data_type=np.float32); w = Vectors(N, m, data_type=np.float32)
    out = []
    for transp, (src, dst) in ((False, (x, y)), (True, (y, w))):
        A.apply(src, dst, transp)
        for reps in (5, 40):
            _lib.check(L.rlh_sync())
            _lib.check(L.rlh_timer_start())
            for _ in range(reps):
                A.apply(src, dst, transp)
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            t = ms.value / reps
            out.append('%s x%-2d %.3f ms %5.1f TF' % ('A^T' if transp else 'A  ', reps, t, 2.0 * M * N * m / t / 1e9))
    print('%6d x %6d: ' % (M, N) + ' | '.join(out), flush=True)
    del rows, A, x, y, w
