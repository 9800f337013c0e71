"""Is the two-operand Gram's time a property of the PAIR of blocks (addresses) or of what ran before it?
Seven blocks as in bench.py; every ordered pair timed 6 times back to back, then the headline's own sequence."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors
L = _lib.lib()
n, m = 9938375, 32
names = ['X', 'AX', 'Y', 'AY', 'Z', 'AZ', 'W']
blocks = []
for nm in names:
    v = Vectors(n, m); v.fill_random(); blocks.append(v)
print('block addresses:', ' '.join('%s=%#x' % (nm, b.data_ptr()) for nm, b in zip(names, blocks)))
res = ctypes.c_void_p(); _lib.check(L.rlh_malloc(ctypes.byref(res), m * m * 8))
ms = ctypes.c_float()
def t_gram(a, b):
    _lib.check(L.rlh_timer_start())
    _lib.check(L.rlh_gram(1, n, m, a.data_ptr(), a.ld(), m, b.data_ptr(), b.ld(), res, None))
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    return ms.value
def t_dots(a):
    _lib.check(L.rlh_timer_start())
    _lib.check(L.rlh_dots(1, n, m, a.data_ptr(), a.ld(), a.data_ptr(), a.ld(), res, None))
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    return ms.value
if os.environ.get('PAIRS'):
  for i in range(7):
    row = []
    for j in range(7):
        if i == j:
            row.append('  -  ')
            continue
        ts = [t_gram(blocks[i], blocks[j]) for _ in range(6)]
        row.append('%.3f' % float(np.median(ts[1:])))
    print('%-3s' % names[i], ' '.join(row))
X, AX, Y, AY, Z, AZ, W = blocks
print('sequences (ms):')
for rep in range(4):
    seq = [('AX.X', t_gram(X, AX)), ('W.W', t_dots(W)), ('Y.AZ', t_gram(AZ, Y)), ('Y.Z', t_gram(Z, Y)), ('dY', t_dots(Y)), ('dZ', t_dots(Z)),
           ('Y.X', t_gram(X, Y)), ('dY', t_dots(Y)), ('Y.X', t_gram(X, Y)), ('AY.X', t_gram(X, AY)), ('AY.Y', t_gram(Y, AY))]
    print('  ' + '  '.join('%s %.3f' % s for s in seq))

def t_copy(a, b):
    _lib.check(L.rlh_timer_start())
    _lib.check(L.rlh_copy(1, n, m, a.data_ptr(), a.ld(), b.data_ptr(), b.ld()))
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    return ms.value
print('gram after copy / after gram / after dots:')
for rep in range(3):
    print('  copy %.3f -> gram %.3f -> gram %.3f -> dots %.3f -> gram %.3f -> gram %.3f' % (
        t_copy(X, W), t_gram(Y, Z), t_gram(AY, AZ), t_dots(W), t_gram(Y, Z), t_gram(AY, AZ)))
