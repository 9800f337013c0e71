"""One row shard of the headline operator (lap3d 215^3) built the way ShardedSparseMatrix builds it, on one GPU: which layout
the library picks for it (fp64 product; fp32 operator of the bfloat16 Chebyshev step) and what the kernels take.
usage: tools/lap_shard_bench.py [shards=8] [which=1] [side=215]"""
import ctypes, os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, CsrOperator
from raleigh_amd.algebra.hip.sparse import Bf16Block
from raleigh_amd.synthetic import lap3d_rows
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 215
m = 32
n = N ** 3
per = -(-(n // shards) // 64) * 64
r0, r1 = which * per, min(n, (which + 1) * per)
loc = sp.csr_matrix(lap3d_rows(N, N, N, 1.0, 1.01, 1.02, r0, r1))
used = np.unique(loc.indices)
halo = used[(used < r0) | (used >= r1)]
nown = r1 - r0
n_own_pad = -(-nown // 8) * 8
newcol = np.full(n, -1, dtype=np.int64)
newcol[r0:r1] = np.arange(nown)
newcol[halo] = n_own_pad + np.arange(len(halo))
nh = -(-len(halo) // 8) * 8
L = _lib.lib()
ms = ctypes.c_float()
def timed(f, reps=20):
    f(); _lib.check(L.rlh_sync())
    ts = []
    for _ in range(reps):
        _lib.check(L.rlh_timer_start()); f(); _lib.check(L.rlh_timer_stop(ctypes.byref(ms))); ts.append(ms.value)
    return float(np.median(ts))
for dt in (np.float64, np.float32):
    Lm = sp.csr_matrix((loc.data.astype(dt), newcol[loc.indices].astype(np.int32), loc.indptr), shape=(nown, n_own_pad + nh))
    Lm.sort_indices()
    op = CsrOperator(Lm, n_own=n_own_pad)
    y, w = Vectors(n_own_pad, m, data_type=dt), Vectors(n_own_pad, m, data_type=dt)
    h = Vectors(nh, m, data_type=dt)
    y.fill_random(); h.fill_random()
    t = timed(lambda: op.apply_ptr(m, y.data_ptr(), y.ld(), w.data_ptr(), w.ld(), h.data_ptr(), h.ld()))
    es = np.dtype(dt).itemsize
    nb = Lm.nnz * (es + 4) + 2 * nown * m * es
    line = 'shard %d of %d of lap3d %d^3 (%d rows, %d halo rows) %s: layout %s, stacks %d, product %.4f ms (%.2f TB/s)' % (
        which, shards, N, nown, len(halo), np.dtype(dt).name, op.layout()[0], op.stacks()[0], t, nb / t / 1e9)
    if dt == np.float32:
        ok = op.bf16_ready(nh)
        line += ', bfloat16 step ready: %s' % ok
        if ok:
            yb, pb, bb, hb = Bf16Block(n_own_pad, 16), Bf16Block(n_own_pad, 16), Bf16Block(n_own_pad, 16), Bf16Block(nh, 16)
            t2 = timed(lambda: op.cheb_step_bf16(16, yb, pb, bb, 0.9, -0.2, 0.01, hb.ptr(), hb.ld))
            line += ', bfloat16 step of 16 vectors %.4f ms (%.2f TB/s)' % (t2, (8.0 * nown * 16 + 48.0 * nown) / t2 / 1e9)
    print(line)
