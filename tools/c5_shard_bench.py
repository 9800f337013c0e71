"""One of eight row shards of BASELINE config 5's operator (126^3 complex128, 64 vectors): product and fused Chebyshev step on
the layouts the library may pick for it -- stacks of one block on the LDS-DMA ring (default) against the interleaved layout
(RLH_SPMM_STACK_SINGLE=0).  usage: tools/c5_shard_bench.py [shards=8] [which=1]"""
import ctypes, os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, CsrOperator
from raleigh_amd.synthetic import hermitian_lap3d_rows
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, m = 126, 64
n = N ** 3
r0, r1 = which * (n // shards), (which + 1) * (n // shards)
loc = sp.csr_matrix(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, r0, r1))
used = np.unique(loc.indices)
halo = used[(used < r0) | (used >= r1)]
nown = r1 - r0
n_own_pad = -(-nown // 8) * 8
newcol = np.full(n, -1, dtype=np.int64)
newcol[r0:r1] = np.arange(nown)
newcol[halo] = n_own_pad + np.arange(len(halo))
nh = -(-len(halo) // 8) * 8
Lm = sp.csr_matrix((loc.data, newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
Lm.sort_indices()
L = _lib.lib()
y, p, b, w = (Vectors(n_own_pad, m, data_type=np.complex128) for _ in range(4))
h = Vectors(nh, m, data_type=np.complex128)
for v in (y, p, b, h):
    v.fill_random()
ms = ctypes.c_float()
def timed(f, reps=20):
    f(); _lib.check(L.rlh_sync())
    ts = []
    for _ in range(reps):
        _lib.check(L.rlh_timer_start()); f(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
    return float(np.median(ts))
B = nown * m * 16
for single in ('1', '0'):
    os.environ['RLH_SPMM_STACK_SINGLE'] = single
    op = CsrOperator(Lm, n_own=n_own_pad)
    t1 = timed(lambda: op.apply_ptr(m, y.data_ptr(), y.ld(), w.data_ptr(), w.ld(), h.data_ptr(), h.ld()))
    t2 = timed(lambda: op.cheb_step_ptr(m, y, p, b, 0.9, -0.2, 0.011, h.data_ptr(), h.ld()))
    print('shard %d of %d (%d rows, %d halo rows), RLH_SPMM_STACK_SINGLE=%s: layout %s stacks %d | product %.4f ms (%.2f TB/s) | fused step %.4f ms (%.2f TB/s)'
          % (which, shards, nown, len(halo), single, op.layout()[0], op.stacks()[0], t1, (2 * B + Lm.nnz * 20) / t1 / 1e9,
             t2, (4 * B + Lm.nnz * 20) / t2 / 1e9))
