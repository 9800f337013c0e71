"""Headline benchmark: one block-JCG INNER ITERATION of the abstract-vectors hot
path (the Gram / dots / SpMM calls the solver issues per iteration,
raleigh/core/solver.py:854-861,968-974,1321-1339,1360,1376-1381,1444-1447) on a
synthetic n x m block, n = 215^3 = 9 938 375 (the "n = 10^7" roofline point of
BASELINE.json), m = 32, fp64, A = 7-point 3-D Laplacian.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL): the rows
are sharded over the ranks, every Gram / dots carries one all-reduce, the SpMM
one halo exchange.  Weak scaling: every GPU holds a 215^3-row shard (the grid
grows along z: 215 x 215 x 215 N), so per-GPU work is fixed and `value` is the
aggregate over the N GPUs.

Prints ONE JSON line (rank 0).  `value` = algorithmic GB/s of the whole job:
bytes of SURVEY 8(d) (9 Gram calls = 16 blocks, 4 self-dots = 4 blocks, one
SpMM = nnz*12 + (n+1)*4 + 2 blocks) divided by the max-over-ranks step time,
operands resident in HBM, every dot/dots returning its result to the host as the
solver requires.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)


def lap3d_rows(nx, ny, nz, ax, ay, az, r0, r1):
    """Rows [r0, r1) of the 7-point Laplacian of raleigh/examples/laplace.py:23-27 as a full
    (both triangles) CSR block with GLOBAL column indices, built without the global matrix."""
    import scipy.sparse as sp
    n = nx * ny * nz
    r = np.arange(r0, r1, dtype=np.int64)
    ix, iy, iz = r % nx, (r // nx) % ny, r // (nx * ny)
    cx, cy, cz = ((nx + 1.0) / ax) ** 2, ((ny + 1.0) / ay) ** 2, ((nz + 1.0) / az) ** 2
    rows, cols, vals = [r], [r], [np.full(r.shape, 2 * (cx + cy + cz))]
    for cond, shift, c in ((ix > 0, -1, cx), (ix < nx - 1, 1, cx), (iy > 0, -nx, cy), (iy < ny - 1, nx, cy),
                           (iz > 0, -nx * ny, cz), (iz < nz - 1, nx * ny, cz)):
        rows.append(r[cond]); cols.append(r[cond] + shift); vals.append(np.full(int(cond.sum()), -c))
    rows = np.concatenate(rows) - r0
    blk = sp.csr_matrix((np.concatenate(vals), (rows, np.concatenate(cols))), shape=(r1 - r0, n))
    blk.sort_indices()
    return blk



class InnerIteration:
    """The Gram / dots / SpMM sequence of one steady-state iteration (standard problem,
    identity preconditioner, no deflation) on blocks X, AX, Y, AY, Z, AZ, W."""

    def __init__(self, blocks, op):
        self.b, self.op = blocks, op

    def headline(self):
        X, AX, Y, AY, Z, AZ, W = self.b
        out = []
        out.append(AX.dot(X))       # XAX   solver.py:857
        out.append(X.dot(X))        # XBX   solver.py:859 (B = I: self-Gram)
        out.append(W.dots(W))       # residual norms, solver.py:968-974
        out.append(Y.dot(AZ))       # ZAY   solver.py:1326
        out.append(Y.dot(Z))        # ZBY   solver.py:1328
        out.append(Y.dots(Y))       # solver.py:1337
        out.append(Z.dots(Z))       # solver.py:1338
        out.append(Y.dot(X))        # Q = Y.dot(BX), solver.py:1360
        out.append(Y.dots(Y))       # normalisation, solver.py:1376
        out.append(Y.dot(X))        # XBY   solver.py:1379
        out.append(Y.dot(Y))        # YBY   solver.py:1380 (self-Gram)
        self.op.apply(Y, AY)        # AY = A Y, solver.py:1444
        out.append(AY.dot(X))       # XAY   solver.py:1446
        out.append(AY.dot(Y))       # YAY   solver.py:1447
        return out

    def all_ops(self, q):
        """The whole per-iteration op mix as the reference's driver issues it (about 65
        blocks of traffic + the SpMM): the headline plus 4 multiply, 7 add-with-matrix, 6 copies and
        1 scale on the same blocks (solver.py:1609-1656, 1321-1381)."""
        X, AX, Y, AY, Z, AZ, W = self.b
        self.headline()
        for src, dst in ((X, W), (AX, W), (Y, W), (AY, W)):
            src.multiply(q, dst)                # new X / AX / Z / AZ from the Ritz coefficients
        for dst, src in ((W, Y), (W, AY), (Y, Z), (AY, AZ), (Y, X), (AY, AX), (W, X)):
            dst.add(src, 1.0, q)                # the second halves of the updates, orthogonalisations
        for src, dst in ((W, Z), (W, AZ), (Y, W), (AY, W), (X, W), (AX, W)):
            src.copy(dst)
        Y.scale(np.full(Y.nvec(), 1.0))

    @staticmethod
    def all_ops_bytes(n, m, es, nnz):
        B = n * m * es
        return InnerIteration.headline_bytes(n, m, es, nnz)[0] + 4 * 2 * B + 7 * 3 * B + 6 * 2 * B + 2 * B

    @staticmethod
    def headline_bytes(n, m, es, nnz):
        B = n * m * es
        gram = 7 * 2 * B + 2 * B          # 7 two-operand + 2 self Grams = 16 blocks
        dots = 4 * B
        spmm = nnz * (es + 4) + (n + 1) * 4 + 2 * B
        return gram + dots + spmm, {'gram': gram, 'dots': dots, 'spmm': spmm}


def cpu_baseline(m, reps_target_s=12.0):
    """The CPU oracle (oracle/, NumPy + the node's BLAS, SciPy SpMM) timed on a bounded sample
    of the same workload: lap3d 100^3 (n = 10^6), m = 32, fp64."""
    from oracle import Vectors as OV, SparseSymmetricMatrix as OS
    from oracle.sparse import lap3d
    N = 100
    A = lap3d(N, N, N, 1.0, 1.01, 1.02)
    n = A.shape[0]
    rng = np.random.default_rng(1)
    blocks = [OV(2 * rng.random((m, n)) - 1) for _ in range(7)]
    it = InnerIteration(blocks, OS(A))
    it.headline()
    t0 = time.perf_counter()
    reps = 0
    while True:
        it.headline()
        reps += 1
        el = time.perf_counter() - t0
        if el > reps_target_s or reps >= 20:
            break
    nbytes, _ = InnerIteration.headline_bytes(n, m, 8, A.nnz)
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {'value': round(nbytes * reps / el / 1e9, 3), 'unit': 'GB/s', 'cores': int(threads),
            'kind': 'port',
            'sample': 'oracle (NumPy/BLAS + SciPy CSR) inner iteration, lap3d 100^3 (n=1e6), m=%d fp64, %d reps in %.1f s'
                      % (m, reps, el)}


def solve_ten(side, comm):
    """Seconds to 10 eigenpairs (second half of BASELINE.json's metric): the repository's
    block-JCG driver on lap3d(side^3), 10 smallest eigenvalues, eigenvector tolerance 1e-6, a
    device-resident polynomial preconditioner (all blocks stay in HBM), rows sharded over the
    ranks; checked against the analytic spectrum."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from oracle.sparse import lap3d_eigenvalues
    n = side ** 3
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 5000
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner
    if comm is None:
        from raleigh_amd.algebra.hip import CsrOperator

        class Op:
            def __init__(self, dtype):
                rows = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, n)
                self.csr = CsrOperator(rows.astype(dtype))
                self.dtype = dtype

            def size(self):
                return n

            def data_type(self):
                return self.dtype

            def apply(self, x, y):
                self.csr.apply_ptr(x.nvec(), x.data_ptr(), x.ld(), y.data_ptr(), y.ld())

            def cheb_step(self, y, p, b, cy, cp, cb):
                self.csr.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb)

            def supports_bf16(self):
                return self.dtype == np.float32 and self.csr.layout()[0] == 'well'

            def cheb_step_bf16(self, m, y, p, b, cy, cp, cb):
                self.csr.cheb_step_bf16(m, y, p, b, cy, cp, cb)
        op, op32, vectors = Op(np.float64), Op(np.float32), None
    else:
        from raleigh_amd.algebra.hip.dist import ShardedVectors, ShardedSparseMatrix, partition
        off = partition(n, comm.size)
        r0, r1 = int(off[comm.rank]), int(off[comm.rank + 1])
        rows = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, r0, r1)
        op = ShardedSparseMatrix.from_local_rows(rows, r0, n, comm, off)
        op32 = ShardedSparseMatrix.from_local_rows(rows.astype(np.float32), r0, n, comm, off)
        vectors = lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm, offsets=off)
    # device-resident Chebyshev polynomial preconditioner (degree 32 on [hi/7000, hi], hi = the
    # Gershgorin bound 4 (cx + cy + cz) of the stencil), evaluated in float32: every block stays in HBM
    hi = 4.0 * sum(((side + 1.0) / a) ** 2 for a in (1.0, 1.01, 1.02))
    # (work blocks in bfloat16, float32 arithmetic; row shards exchange 2-byte halo rows)
    T = ChebyshevPreconditioner(None, hi, ratio=7000.0, degree=32, low_precision_op=op32, storage='bf16')
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(None, T=T, which=10, tol=1e-6, verb=-1, opt=opt, vectors=vectors,
                                  operator=op)
    seconds = time.perf_counter() - t0
    ana = lap3d_eigenvalues(side, side, side, 1.0, 1.01, 1.02, 10)
    err = float(np.max(np.abs(lmd[:10] - ana) / ana)) if status == 0 and len(lmd) >= 10 else None
    return {'problem': 'lap3d %d^3 (n=%d), 10 smallest eigenpairs, eigenvector tol 1e-6, device Chebyshev '
                       'preconditioner (degree 32, float32 arithmetic, bfloat16 work blocks), rows sharded over the ranks' % (side, n),
            'seconds': round(seconds, 3), 'status': int(status), 'iterations': int(partial_hevp.last['iterations']),
            'max_rel_eigenvalue_error': err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--side', type=int, default=215, help='lap3d side (n = side^3)')
    ap.add_argument('--m', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='use the sharded (torch.distributed / RCCL) code path even with one rank')
    ap.add_argument('--solve-side', type=int, default=215,
                    help='lap3d side of the end-to-end "seconds to 10 eigenpairs" run (0: skip)')
    args = ap.parse_args()

    # Everything any library prints (RCCL prints a version banner on stdout) goes to stderr;
    # the real stdout carries exactly one JSON line, written at the very end by rank 0.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('bench.py --gpus %d must be launched by torch.distributed.run with %d ranks'
                         % (args.gpus, args.gpus))

    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    comm = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        from raleigh_amd.algebra.hip.dist import Comm, ShardedVectors, ShardedSparseMatrix, partition
        comm = Comm()
    L = _lib.lib(local_rank)

    side, m, es = args.side, args.m, 8
    nzg = side * world                           # weak scaling: the grid grows along z with the GPUs
    n = side * side * nzg
    nnz = 7 * n - 2 * (side * nzg * 2 + side * side)      # 7-point stencil minus the six faces
    if comm is None:
        r0, r1 = 0, n
    else:
        off = partition(n, world)
        r0, r1 = int(off[rank]), int(off[rank + 1])
    nloc = r1 - r0

    # ---- synthetic inputs, resident in HBM before the timed region
    rng = np.random.default_rng(1 + rank)
    if comm is None:
        mk = lambda: Vectors(nloc, m, data_type=np.float64)
    else:
        mk = lambda: ShardedVectors(n, m, np.float64, comm=comm, offsets=off)
    blocks = [mk() for _ in range(7)]
    for j in range(m):                            # U(-1, 1), one vector at a time (bounded host memory)
        blocks[0].select(1, j)
        Vectors.fill(blocks[0], 2 * rng.random((1, nloc)) - 1)
    blocks[0].select(m)
    for i, b in enumerate(blocks[1:], 1):         # distinct random-looking blocks from device-side ops
        blocks[0].copy(b, np.roll(np.arange(m), i))
        b.add(blocks[0], 0.37 * i)
    rows = lap3d_rows(side, side, nzg, 1.0, 1.01, 1.02 * world, r0, r1)
    assert world > 1 or rows.nnz == nnz
    if comm is None:
        csr = CsrOperator(rows)

        class Op:
            def apply(self, x, y):
                csr.apply_ptr(x.nvec(), x.data_ptr(), x.ld(), y.data_ptr(), y.ld())
        op = Op()
    else:
        op = ShardedSparseMatrix.from_local_rows(rows, r0, n, comm, off)
    del rows
    it = InnerIteration(blocks, op)

    def sync_all():
        _lib.check(L.rlh_sync())
        if comm is not None:
            import torch
            torch.cuda.synchronize()
            comm.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        it.headline()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it.headline()
    sync_all()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    nbytes, parts = InnerIteration.headline_bytes(n, m, es, nnz)
    value = nbytes / (ms_per_step * 1e-3) / 1e9

    # ---- the all-ops figure beside the headline (reported, not the metric): a few steps of the full mix
    q = np.eye(m) + 1e-3 * np.random.default_rng(7).standard_normal((m, m))
    it.all_ops(q)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(5):
        it.all_ops(q)
    sync_all()
    all_ms = (time.perf_counter() - t0) / 5 * 1e3
    all_bytes = InnerIteration.all_ops_bytes(n, m, es, nnz)

    # ---- roofline of the dominant kernel (two-operand Gram), HIP events on the kernels' stream
    X, AX = blocks[0], blocks[1]
    ms = ctypes.c_float()
    res = ctypes.c_void_p()
    _lib.check(L.rlh_malloc(ctypes.byref(res), m * m * es))
    code = _lib.dtype_code(np.float64)
    reps = 20
    gram = lambda: L.rlh_gram(code, nloc, m, X.data_ptr(), X.ld(), m, AX.data_ptr(), AX.ld(), res, None)
    _lib.check(gram())
    _lib.check(L.rlh_timer_start())
    for _ in range(reps):
        _lib.check(gram())
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    gram_ms = ms.value / reps
    gram_bytes = 2 * nloc * m * es
    achieved = gram_bytes / (gram_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'gram_traffic.json')
    if os.path.exists(tpath) and world == 1:
        try:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    roofline = {'bound': 'hbm', 'kernel': 'gram_kernel<fp64, 2x2 tiles> (X.dot(Y), m=k=%d)' % m,
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'algorithmic_bytes_per_launch': gram_bytes, 'avg_launch_ms': round(gram_ms, 4),
                'inner_iteration_frac': round(value / world / HBM_PEAK_GBS, 4)}

    out = {'metric': 'inner-iter GB/s vs HBM roofline (Gram+dots+SpMM of one block-JCG iteration)',
           'value': round(value, 1), 'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'block-JCG inner iteration: 9 Gram + 4 dots + 1 SpMM, n=%dx%dx%d=%d rows '
                                  '(%d^3 per GPU), m=%d, 7-pt Laplacian nnz=%d, rows sharded over %d GPU(s)'
                                  % (side, side, nzg, n, side, m, nnz, world),
                      'n': n, 'm': m, 'nnz': nnz, 'algorithmic_bytes_per_step': nbytes,
                      'bytes_breakdown': parts},
           'roofline': roofline,
           'all_ops': {'ms_per_step': round(all_ms, 3), 'algorithmic_bytes_per_step': all_bytes,
                       'gbs': round(all_bytes / (all_ms * 1e-3) / 1e9, 1),
                       'what': 'headline + 4 multiply + 7 add(q) + 6 copy + 1 scale (the reference driver\'s per-iteration mix)'}}
    if args.solve_side > 0:
        try:
            out['solve'] = solve_ten(args.solve_side, comm)
        except Exception as e:      # the headline above is already measured: report, do not lose the line
            out['solve'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(m)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == '__main__':
    main()
