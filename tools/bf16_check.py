"""Debug: the bfloat16 Chebyshev step on the stacked layout against the unstacked kernel (same handle), large sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
side, m = int(sys.argv[1]), int(sys.argv[2])
from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
from raleigh_amd.algebra.hip.sparse import Bf16Block
from oracle.sparse import lap3d
A = lap3d(side, side, side, 1.0, 1.01, 1.02).astype(np.float32)
n = A.shape[0]
op = SparseSymmetricMatrix(A)
print('n', n, 'layout', op.layout())
rng = np.random.default_rng(1)
col = rng.standard_normal((1, n)).astype(np.float32)


def run():
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
    for it in range(3):
        op.cheb_step_bf16(m, y, p, b, 1.3, -0.3, 1e-5)
        y, p = p, y
    out = Vectors(n, m, data_type=np.float32)
    y.unpack(out)
    return out.data()
os.environ['RLH_SPMM_STACK_BF16'] = '1'
a = run()
os.environ['RLH_SPMM_STACK_BF16'] = '0'
b = run()
bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
print('finite', np.isfinite(a).all(), np.isfinite(b).all(), 'mismatches', len(bad))
if len(bad):
    print('vectors', np.unique(bad[:, 0])[:20], 'rows min/max', bad[:, 1].min(), bad[:, 1].max())
    rows = np.unique(bad[:, 1])
    print('row blocks', np.unique(rows // 1024)[:40], '... count', len(np.unique(rows // 1024)))
    print('within-block offsets', np.unique(rows % 1024)[:40])
    print(bad[:10], a[tuple(bad[:5].T)], b[tuple(bad[:5].T)])
