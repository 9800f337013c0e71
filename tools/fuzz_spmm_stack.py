"""Randomised parity sweep of the stacked SpMM paths against SciPy (GPU box; not a pytest file):
`python tools/fuzz_spmm_stack.py [cases] [seed]`.  Stacks forced (RLH_SPMM_STACK=2) on random matrices with at most 8
entries per row and random column locality -- stencils on random grids, random bands, blocks of far couplings, empty
rows, ragged tails, a number of row blocks that leaves a stack of one -- all four types, random block sizes, LDS-DMA and
register-staged kernels, and (float32) the bfloat16 Chebyshev step against the unstacked kernel."""
import os, sys
os.environ['RLH_SPMM_STACK'] = '2'
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
from raleigh_amd.algebra.hip.sparse import Bf16Block
from oracle import ops
from oracle.sparse import lap3d
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DT = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}


def random_matrix(key):
    kind = rng.integers(0, 4)
    if kind == 0:                                   # stencil on a random grid
        nx, ny, nz = (int(rng.integers(3, 90)) for _ in range(3))
        A = lap3d(nx, ny, nz, 1.0, 1.01, 1.02)
    elif kind == 1:                                 # random symmetric band with holes
        n = int(rng.integers(1500, 60000))
        offs = sorted(set(int(o) for o in rng.integers(1, min(4000, n - 1), size=3)))
        diags = [rng.standard_normal(n)] + [rng.standard_normal(n - o) * (rng.random(n - o) < 0.7) for o in offs]
        U = sp.diags(diags, [0] + offs, format='csr')
        A = U + sp.triu(U, 1).T
    elif kind == 2:                                 # 2-D stencil
        nx, ny = int(rng.integers(5, 400)), int(rng.integers(5, 400))
        A = lap3d(nx, ny, 1, 1.0, 1.01, 1.02)
    else:                                           # tridiagonal + one far symmetric coupling per row, some empty rows
        n = int(rng.integers(2000, 50000))
        far = min(int(rng.integers(1100, n // 2 + 1101)), n - 1)
        d = rng.standard_normal(n)
        d[rng.random(n) < 0.01] = 0.0
        U = sp.diags([d, rng.standard_normal(n - 1), rng.standard_normal(n - far)], [0, 1, far], shape=(n, n), format='csr')
        A = U + sp.triu(U, 1).T
    A = sp.csr_matrix(A)
    if key in 'cz':
        S = sp.triu(A, k=1)
        A = A + 0.5j * S - 0.5j * S.T
    A = sp.csr_matrix(A.astype(DT[key]))
    A.eliminate_zeros()
    A.sort_indices()
    return A


worst = {}
used = {'stacks': 0, 'none': 0}
for case in range(cases):
    key = 'sdcz'[int(rng.integers(0, 4))]
    A = random_matrix(key)
    n = A.shape[0]
    if np.diff(A.indptr).max() > 8:
        continue
    m = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 32, 40]))
    os.environ['RLH_SPMM_STACK_DMA'] = str(rng.choice([0, 1, 2]))
    os.environ['RLH_SPMM_STACK_PAT'] = str(rng.choice([0, 1]))       # the value dictionary (read at create and at launch)
    os.environ['RLH_SPMM_STACK_DPAT'] = str(rng.choice([0, 1]))      # ... and the position patterns
    op = SparseSymmetricMatrix(A)
    lay = op.layout()
    used['stacks' if lay[3] > 0 else 'none'] += 1
    x = (2 * rng.random((m, n)) - 1).astype(DT[key])
    if key in 'cz':
        x = x + 1j * (2 * rng.random((m, n)) - 1).astype(DT[key])
    X, Y = Vectors(x), Vectors(n, m, data_type=DT[key])
    Y.fill(np.full((m, n), np.nan, dtype=DT[key]))
    op.apply(X, Y)
    y = Y.data()
    ref = (A.astype(np.complex128 if key in 'cz' else np.float64) @ x.T).T
    err = float(np.linalg.norm(y - ref) / max(np.linalg.norm(ref), 1e-300))
    tol = 3e-6 if key in 'sc' else 1e-13
    worst[key] = max(worst.get(key, 0.0), err)
    if not err < tol or not np.all(np.isfinite(y)):
        print('FAIL apply', key, 'n', n, 'm', m, 'layout', lay, 'dma', os.environ['RLH_SPMM_STACK_DMA'], 'err %.3e' % err)
        sys.exit(1)
    if key == 's' and op.supports_bf16() and lay[3] > 0:
        y0, p0, b0 = (ops.bf16_round((2 * rng.random((m, n)) - 1).astype(np.float32)) for _ in range(3))

        def step():
            blocks = []
            for a in (y0, p0, b0):
                blk = Bf16Block(n, m)
                blk.pack(Vectors(a), 1.0)
                blocks.append(blk)
            yb, pb, bb = blocks
            op.cheb_step_bf16(m, yb, pb, bb, 1.25, -0.25, 0.05)
            out = Vectors(n, m, data_type=np.float32)
            pb.unpack(out)
            return out.data()
        os.environ['RLH_SPMM_STACK_BF16'] = '1'
        a = step()
        os.environ['RLH_SPMM_STACK_BF16'] = '0'
        b = step()
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            print('FAIL bf16 step', 'n', n, 'm', m, 'layout', lay, 'mismatches', len(bad), bad[:5])
            sys.exit(1)
        worst['bf16 steps'] = worst.get('bf16 steps', 0) + 1
print('ok: %d cases, %s; worst relative errors %s' % (cases, used, {k: ('%.2e' % v if isinstance(v, float) else v) for k, v in worst.items()}))
