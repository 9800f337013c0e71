"""Dense apply pair (A x, A^H y, m = 128) at several shapes, matrices filled on the device: TFLOP/s per product (real
flop: complex counts 4 multiply-adds per element pair), a short and a long timed loop (clock ramp).
usage: tools/gemm_shapes.py [--dtype s|d|c|z] [MxN ...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, Matrix
L = _lib.lib()
m = 128
ms = ctypes.c_float()
shapes = [(20000, 20000), (40000, 20000), (20000, 40000), (40000, 40000), (62500, 40000), (20096, 20000), (24576, 20000)]
argv = sys.argv[1:]
key = 's'
if argv and argv[0] == '--dtype':
    key, argv = argv[1], argv[2:]
dt = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}[key]
mult = 4.0 if key in 'cz' else 1.0
if argv:
    shapes = [tuple(int(v) for v in a.split('x')) for a in argv]
for (M, N) in shapes:
    rows = Vectors(N, M, data_type=dt)
    rows.fill_random()
    A = Matrix(rows)
    x = Vectors(N, m, data_type=dt); x.fill_random()
    y = Vectors(M, m, data_type=dt); w = Vectors(N, m, data_type=dt)
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
            out.append('%s x%-2d %.3f ms %5.1f TF' % ('A^T' if transp else 'A  ', reps, t, mult * 2.0 * M * N * m / t / 1e9))
    print('%s %6d x %6d: ' % (key, M, N) + ' | '.join(out), flush=True)
    del rows, A, x, y, w
