"""Times the bfloat16 Chebyshev step on lap3d side^3 with m vectors (HIP events, variants taking turns):
python tools/bf16_time.py 215 16 [--dbg]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
side, m = int(sys.argv[1]), int(sys.argv[2])
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
from raleigh_amd.algebra.hip.sparse import Bf16Block
from oracle.sparse import lap3d
L = _lib.lib()
A = lap3d(side, side, side, 1.0, 1.01, 1.02).astype(np.float32)
n = A.shape[0]
op = SparseSymmetricMatrix(A)
print('n', n, 'layout', op.layout())
rng = np.random.default_rng(1)
col = rng.standard_normal((1, n)).astype(np.float32)
blocks = []
for s in range(3):
    blk = Bf16Block(n, m)
    V = Vectors(n, m, data_type=np.float32)
    for j in range(m):
        V.select(1, j)
        V.fill(np.roll(col, 13 * j + 5 * s + 1, axis=1))
    V.select(m)
    blk.pack(V, 1.0)
    blocks.append(blk)
y, p, b = blocks
variants = [('1024-row blocks', {'RLH_SPMM_STACK_BF16': '0', 'RLH_SPMM_BF16_DBG': '0'}), ('stacks', {'RLH_SPMM_STACK_BF16': '1', 'RLH_SPMM_BF16_DBG': '0'})]
if '--dbg' in sys.argv:
    what = {1: 'no y DMAs', 2: 'no row products', 3: 'no y DMAs, no row products', 4: 'no operands / update', 5: 'row products only', 6: 'y DMAs only', 7: 'entries, waits, barriers only'}
    variants += [('stacks, ' + what[d], {'RLH_SPMM_STACK_BF16': '1', 'RLH_SPMM_BF16_DBG': str(d)}) for d in range(1, 8)]
ms = ctypes.c_float()
times = {k: [] for k, _ in variants}
for rep in range(12):
    for name, env in variants:
        os.environ.update(env)
        _lib.check(L.rlh_timer_start())
        op.cheb_step_bf16(m, y, p, b, 1.0, 0.0, 1e-9)
        _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
        if rep >= 2:
            times[name].append(ms.value)
nbytes = 8.0 * n * m + 48.0 * n
for name, _ in variants:
    med = float(np.median(times[name]))
    print('%-44s %8.1f us  %7.1f GB/s of the 1024-row layout\'s algorithmic bytes' % (name, med * 1e3, nbytes / med / 1e6))
