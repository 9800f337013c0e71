"""Headline benchmark: one block-JCG INNER ITERATION of the abstract-vectors hot
path (the Gram / dots / SpMM calls the solver issues per iteration,
raleigh/core/solver.py:854-861,968-974,1321-1339,1360,1376-1381,1444-1447) on a
synthetic n x m block, n = 215^3 = 9 938 375 (the "n = 10^7" roofline point of
BASELINE.json), m = 32, fp64, A = 7-point 3-D Laplacian.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 runs one rank per GPU under torch.distributed.run (RCCL) -- started by the driver, or, when WORLD_SIZE is not
set, by this script itself as a child process (`python bench.py --gpus 8` works as typed): the rows
are sharded over the ranks, every Gram / dots carries one all-reduce, the SpMM
one halo exchange.  `--scaling strong` (default): the 215^3 rows are split over the N
ranks -- the north-star's ">= 6x at 8 GPUs at n = 10M" question; `value` is the whole
job's rate, so value(N) / value(1) is the speed-up.  `--scaling weak`: every GPU holds a
215^3-row shard (the grid grows along z: 215 x 215 x 215 N), per-GPU work is fixed and
`value` is the aggregate over the N GPUs.  At N = 1 the two are the same workload; with
N > 1 the run also reports the other mode under "also", the strong figure of the batched
(5 round trips) iteration under "fused", the measured latency of one reduction round trip
under "collectives", and BASELINE config 4 (pca of a row-sharded 62 500 N x 40 000 matrix)
under "config4".

Prints ONE JSON line (rank 0).  `value` = algorithmic GB/s of the whole job:
bytes of SURVEY 8(d) (9 Gram calls = 16 blocks, 4 self-dots = 4 blocks, one
SpMM = nnz*12 + (n+1)*4 + 2 blocks) divided by the max-over-ranks step time,
operands resident in HBM, every dot/dots returning its result to the host as the
solver requires.  The step time is the wall-clock mean over the K steps between two
barriers (the driver's contract); the median of per-step HIP-event times is reported
beside it.
"""

import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)


class InnerIteration:
    """The Gram / dots / SpMM sequence of one steady-state iteration (standard problem,
    identity preconditioner, no deflation) on blocks X, AX, Y, AY, Z, AZ, W."""

    def __init__(self, blocks, op):
        self.b, self.op = blocks, op

    def headline(self):
        """One blocking call per reduction, in the reference driver's order (13 host round trips)."""
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

    def fused(self):
        """The same quantities as this repository's driver requests them (raleigh_amd/core/solver.py):
        stacked Grams that read every block once and five batches = five host round trips."""
        X, AX, Y, AY, Z, AZ, W = self.b
        out = []
        rb = X.reduction_batch()
        rb.gram([X], [AX, X])       # XAX, XBX
        rb.dots(W, W)               # residual norms
        out.append(rb.run())
        rb = Y.reduction_batch()
        rb.gram([Y], [AZ, Z])       # ZAY, ZBY
        rb.dots(Y, Y)
        rb.dots(Z, Z)
        out.append(rb.run())
        out.append(Y.dot(X))        # Q
        rb = Y.reduction_batch()
        rb.gram([Y], [X, Y])        # XBY, YBY (+ the norms of Y on its diagonal)
        out.append(rb.run())
        self.op.apply(Y, AY)
        rb = AY.reduction_batch()
        rb.gram([AY], [X, Y])       # XAY, YAY
        out.append(rb.run())
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

    @staticmethod
    def fused_bytes(n, m, es, nnz):
        B = n * m * es                    # stacked Grams: 3 + 3 + 2 + 2 + 3 blocks, dots 3 blocks
        return 13 * B + 3 * B + nnz * (es + 4) + (n + 1) * 4 + 2 * B


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def host_blas():
    """Which BLAS the CPU baseline's NumPy calls land in (vendor, version, threading layer)."""
    try:
        from threadpoolctl import threadpool_info
        return '; '.join('%s %s (%s, %s threads)' % (p.get('internal_api'), p.get('version'), p.get('threading_layer', p.get('user_api')),
                                                    p.get('num_threads')) for p in threadpool_info() if p.get('user_api') == 'blas') or 'unknown'
    except Exception:
        return 'unknown'


def cpu_baseline(side, m, budget_s=20.0):
    """The CPU oracle (oracle/: NumPy + the node's BLAS, SciPy CSR SpMM) on the SAME workload as the
    GPU headline -- lap3d side^3, m vectors, fp64, the same 14 calls -- timed on the host cores for a
    bounded number of repetitions (at 215^3 one repetition moves 57 GB: 2-3 repetitions)."""
    from oracle import Vectors as OV, SparseSymmetricMatrix as OS
    from raleigh_amd.synthetic import lap3d_rows
    n = side ** 3
    A = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, n)
    rng = np.random.default_rng(1)
    first = 2 * rng.random((m, n)) - 1
    blocks = [OV(first)]
    for i in range(1, 7):
        blocks.append(OV(np.roll(first, i, axis=0) * (1.0 + 0.37 * i)))
    it = InnerIteration(blocks, OS(A))
    t0 = time.perf_counter()
    it.headline()                                        # (first repetition: also warms the BLAS threads up)
    first_s = time.perf_counter() - t0
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or (time.perf_counter() - t0 + first_s * 1.2 < budget_s and reps < 5):
        it.headline()
        reps += 1
    el = time.perf_counter() - t0
    nbytes, _ = InnerIteration.headline_bytes(n, m, 8, A.nnz)
    return {'value': round(nbytes * reps / el / 1e9, 3), 'unit': 'GB/s', 'cores': host_threads(), 'kind': 'port', 'blas': host_blas(),
            'sample': 'oracle (NumPy/BLAS + SciPy CSR) on the same input as the GPU headline: lap3d %d^3 (n=%d), m=%d '
                      'fp64, %d repetition(s) of the 14-call inner iteration in %.1f s' % (side, n, m, reps, el)}


_ROWS = {}


def lap_rows(side, nz, r0, r1):
    """Rows [r0, r1) of lap3d side x side x nz as float64 CSR; the last request is kept (the float64 and the float32
    operator of one solve are built from the same rows: generated once, 1 s at 215^3)."""
    from raleigh_amd.synthetic import lap3d_rows
    key = (side, nz, r0, r1)
    if key not in _ROWS:
        _ROWS.clear()
        _ROWS[key] = lap3d_rows(side, side, nz, 1.0, 1.01, 1.02 * nz / side, r0, r1)
    return _ROWS[key]


def lap_operator(side, dtype, comm, off, nz=None):
    """(operator with apply / cheb_step / size / data_type, vectors factory) for lap3d side x side x nz."""
    nz = side if nz is None else nz
    n = side * side * nz
    if comm is None:
        from raleigh_amd.algebra.hip import CsrOperator

        class Op:
            def __init__(self):
                self.csr = CsrOperator(lap_rows(side, nz, 0, n).astype(dtype, copy=False))

            def size(self):
                return n

            def data_type(self):
                return dtype

            def apply(self, x, y):
                self.csr.apply_ptr(x.nvec(), x.data_ptr(), x.ld(), y.data_ptr(), y.ld())

            def cheb_step(self, y, p, b, cy, cp, cb):
                self.csr.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb)

            def supports_bf16(self):
                return self.csr.bf16_ready()

            def cheb_step_bf16(self, m, y, p, b, cy, cp, cb):
                self.csr.cheb_step_bf16(m, y, p, b, cy, cp, cb)
        return Op()
    from raleigh_amd.algebra.hip.dist import ShardedSparseMatrix
    r0, r1 = int(off[comm.rank]), int(off[comm.rank + 1])
    rows = lap_rows(side, nz, r0, r1).astype(dtype, copy=False)
    return ShardedSparseMatrix.from_local_rows(rows, r0, n, comm, off)


def solve_ten(side, comm):
    """Seconds to 10 eigenpairs (second half of BASELINE.json's metric): the repository's
    block-JCG driver on lap3d(side^3), 10 smallest eigenvalues, eigenvector tolerance 1e-6, a
    device-resident polynomial preconditioner (all blocks stay in HBM), rows sharded over the
    ranks; checked against the analytic spectrum."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner
    from oracle.sparse import lap3d_eigenvalues
    n = side ** 3
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 5000
    vectors, off = None, None
    if comm is not None:
        from raleigh_amd.algebra.hip.dist import ShardedVectors, partition
        off = partition(n, comm.size)
        vectors = lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm, offsets=off)
    t_setup = time.perf_counter()
    op, op32 = lap_operator(side, np.float64, comm, off), lap_operator(side, np.float32, comm, off)
    # device-resident Chebyshev polynomial preconditioner (degree 32 on [hi/7000, hi], hi = the
    # Gershgorin bound 4 (cx + cy + cz) of the stencil), evaluated in float32: every block stays in HBM
    # (work blocks in bfloat16, float32 arithmetic; row shards exchange 2-byte halo rows)
    hi = 4.0 * sum(((side + 1.0) / a) ** 2 for a in (1.0, 1.01, 1.02))
    T = ChebyshevPreconditioner(None, hi, ratio=7000.0, degree=32, low_precision_op=op32, storage='bf16')
    _ROWS.clear()
    _lib_sync()
    t_setup = time.perf_counter() - t_setup
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(None, T=T, which=10, tol=1e-6, verb=-1, opt=opt, vectors=vectors, operator=op)
    seconds = time.perf_counter() - t0
    ana = lap3d_eigenvalues(side, side, side, 1.0, 1.01, 1.02, 10)
    err = float(np.max(np.abs(lmd[:10] - ana) / ana)) if status == 0 and len(lmd) >= 10 else None
    return {'problem': 'lap3d %d^3 (n=%d), 10 smallest eigenpairs, eigenvector tol 1e-6, device Chebyshev '
                       'preconditioner (degree 32, float32 arithmetic, bfloat16 work blocks; not one of the '
                       'reference\'s preconditioners: see solve_ilu for its ILU), rows sharded over the ranks' % (side, n),
            'seconds': round(seconds, 3), 'setup_seconds': round(t_setup, 3),
            'setup_what': 'this rank\'s rows of the matrix as SciPy CSR (host) + the float64 and float32 device operators (layout build)',
            'status': int(status), 'iterations': int(partial_hevp.last['iterations']),
            'max_rel_eigenvalue_error': err}


def _lib_sync():
    from raleigh_amd import _lib
    _lib.check(_lib.lib().rlh_sync())


def solve_ilu_pair(side):
    """The reference's own configuration -- partial_hevp(A, T=IncompleteLU(A), which=10), ILUT(1e-6, fill 1)
    as sparse_mkl.py:122-140 -- on lap3d side^3: on the GPU (host ILUT once, level-scheduled triangular
    solves on the device) and, beside it, the same driver on the CPU oracle backend with the same factors
    (SciPy triangular solves), both timed from the factorised preconditioner to ten eigenpairs."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU, ilut
    from raleigh_amd.core.solver import Options, Problem, Solver, DefaultConvergenceCriteria
    from raleigh_amd.synthetic import lap3d_rows
    from oracle import Vectors as OV, SparseSymmetricMatrix as OS
    from oracle.sparse import lap3d_eigenvalues
    n = side ** 3
    A = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, n)
    ana = lap3d_eigenvalues(side, side, side, 1.0, 1.01, 1.02, 10)
    out = {'problem': 'lap3d %d^3 (n=%d), 10 smallest eigenpairs, eigenvector tol 1e-6, ILUT(1e-6, fill 1) '
                      'preconditioner as the reference\'s IncompleteLU' % (side, n)}
    t0 = time.perf_counter()
    T = IncompleteLU(A)
    T.factorize()
    out['ilut_host_seconds'] = round(time.perf_counter() - t0, 3)
    out['levels'] = list(T.levels)
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 2000
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1, opt=opt)
    out['gpu'] = {'seconds': round(time.perf_counter() - t0, 3), 'setup_seconds': out['ilut_host_seconds'],
                  'setup_what': 'host ILUT + device triangular-solve set-up (the operator itself is built inside the timed solve)',
                  'status': int(status),
                  'iterations': int(partial_hevp.last['iterations']),
                  'max_rel_eigenvalue_error': float(np.max(np.abs(lmd[:10] - ana) / ana)) if status == 0 else None}
    # CPU: the same driver on the oracle's Vectors, the same ILUT factors
    lo, up = ilut(A, 1e-6, min(n - 1, A.nnz // n))
    lo1 = sp.csr_matrix(lo + sp.identity(n))

    class HostILU:
        def apply(self, x, y):
            w = sla.spsolve_triangular(lo1, x.data().T, lower=True)
            y.fill(np.ascontiguousarray(sla.spsolve_triangular(up, w, lower=False).T))
    np.random.seed(1)
    ev = OV(n, 0, data_type=np.float64)
    solver = Solver(Problem(ev, OS(A)))
    solver.set_preconditioner(HostILU())
    opt = Options()
    opt.max_iter = 2000
    opt.convergence_criteria = DefaultConvergenceCriteria()
    opt.convergence_criteria.set_error_tolerance('k eigenvector error', 1e-6)
    t0 = time.perf_counter()
    status = solver.solve(ev, opt, which=(10, 0))
    lc = np.sort(solver.eigenvalues)
    out['cpu'] = {'seconds': round(time.perf_counter() - t0, 3), 'status': int(status), 'iterations': int(solver.iteration),
                  'cores': host_threads(), 'kind': 'port',
                  'max_rel_eigenvalue_error': float(np.max(np.abs(lc[:10] - ana) / ana)) if len(lc) >= 10 else None}
    return out


def solve_ilu_large(side):
    """The reference's configuration -- partial_hevp(A, T=IncompleteLU(A), which=10), ILUT(1e-6, fill 1) -- at a size where
    only the GPU leg is run (the CPU port needs minutes there): set-up (host ILUT + device triangular-solve set-up), one
    application of the preconditioner, seconds to ten eigenpairs, eigenvalue error against the analytic spectrum."""
    from raleigh_amd import _lib
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    from raleigh_amd.core.solver import Options
    from raleigh_amd.synthetic import lap3d_rows
    from oracle.sparse import lap3d_eigenvalues
    n = side ** 3
    A = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, n)
    ana = lap3d_eigenvalues(side, side, side, 1.0, 1.01, 1.02, 10)
    t0 = time.perf_counter()
    T = IncompleteLU(A)
    T.factorize()
    _lib_sync()
    t_setup = time.perf_counter() - t0
    m = 16
    B_, Z_ = Vectors(n, m), Vectors(n, m)
    B_.fill_random()
    ta = timed_calls(_lib.lib(), lambda: T.apply(B_, Z_), 5)
    del B_, Z_
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 3000
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1, opt=opt)
    seconds = time.perf_counter() - t0
    return {'problem': 'lap3d %d^3 (n=%d), 10 smallest eigenpairs, eigenvector tol 1e-6, ILUT(1e-6, fill 1) as the reference\'s '
                       'IncompleteLU, GPU only' % (side, n),
            'seconds': round(seconds, 3), 'setup_seconds': round(t_setup, 3),
            'setup_what': 'host ILUT + device triangular-solve set-up (the operator itself is built inside the timed solve)',
            'status': int(status), 'iterations': int(partial_hevp.last['iterations']), 'levels': list(T.levels),
            'ilu_apply_ms_16_vectors': round(ta, 3), 'ilu_apply_gbs': round(T.chain().algorithmic_bytes(m) / ta / 1e6, 1),
            'max_rel_eigenvalue_error': float(np.max(np.abs(lmd[:10] - ana) / ana)) if status == 0 and len(lmd) >= 10 else None}


def timed_calls(L, fn, reps):
    """Median HIP-event time (ms) of `reps` back-to-back calls of fn on the library stream."""
    from raleigh_amd import _lib
    ms = ctypes.c_float()
    fn()
    _lib.check(L.rlh_sync())
    ts = []
    for _ in range(reps):
        _lib.check(L.rlh_timer_start())
        fn()
        _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
        ts.append(ms.value)
    return float(np.median(ts))


def config_legs(L):
    """Per-configuration figures of BASELINE.json's other configs on one GPU (reported beside the
    headline, not part of `value`): config 3 (FE-like shipsec5 surrogate: SpMM roofline, ILU solve),
    config 5 (complex128, m = 64: Gram / SpMM / update at n = 126^3), config 2 (pca 20000^2, npc = 200)."""
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, CsrOperator
    from raleigh_amd.synthetic import fe_surrogate, hermitian_lap3d_rows
    out = {}
    # ---- config 3
    A = fe_surrogate()
    n, m = A.shape[0], 16
    op = CsrOperator(A)
    X, Y = Vectors(n, m), Vectors(n, m)
    X.fill_random()
    t = timed_calls(L, lambda: op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld()), 30)
    nbytes = A.nnz * 12 + (n + 1) * 4 + 2 * n * m * 8
    c3 = {'workload': 'FE-like shipsec5 surrogate n=%d nnz=%d (%.1f per row), fp64, m=%d' % (n, A.nnz, A.nnz / n, m),
          'spmm_ms': round(t, 4), 'spmm_gbs': round(nbytes / t / 1e6, 1), 'spmm_frac_of_hbm_peak': round(nbytes / t / 1e6 / HBM_PEAK_GBS, 4),
          'layout': op.layout()[0]}
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.precond import IncompleteLU
    np.random.seed(1)
    T = IncompleteLU(A)
    t0 = time.perf_counter()
    T.factorize()
    c3['ilut_host_seconds'] = round(time.perf_counter() - t0, 2)
    B_, Z_ = Vectors(n, m), Vectors(n, m)
    B_.fill_random()
    ta = timed_calls(L, lambda: T.apply(B_, Z_), 5)
    c3['ilu_apply_ms'] = round(ta, 3)
    c3['ilu_apply_gbs'] = round(T.chain().algorithmic_bytes(m) / ta / 1e6, 1)
    c3['ilu_levels'] = list(T.levels)
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(A, T=T, which=10, tol=1e-6, verb=-1)
    c3['solve'] = {'seconds': round(time.perf_counter() - t0, 3), 'setup_seconds': c3['ilut_host_seconds'],
                   'setup_what': 'host ILUT(1e-6, fill 1) on the host threads + device triangular-solve set-up',
                   'status': int(status),
                   'iterations': int(partial_hevp.last['iterations']), 'smallest': [float(v) for v in lmd[:3]]}
    out['config3'] = c3
    del op, X, Y, T, B_, Z_
    # ---- config 1 (the reference's CPU-runnable case: lap3d, six eigenvalues nearest 0 by DIRECT shift-invert)
    from raleigh_amd.algebra.hip.host_ops import SparseSymmetricSolver
    from raleigh_amd.synthetic import lap3d_rows
    from oracle.sparse import lap3d_eigenvalues                      # (the checker: closed-form spectrum)
    N1 = 32
    A1 = lap3d_rows(N1, N1, N1, 1.0, 1.01, 1.02, 0, N1 ** 3)
    c1 = {'workload': 'lap3d %d^3 = %d rows fp64, 6 eigenvalues nearest 0, direct shift-invert: L D L^H on the host '
                      '(multifrontal, 1x1 / 2x2 pivots), L^-1 / D^-1 / L^-H on the device' % (N1, N1 ** 3)}
    t0 = time.perf_counter()
    sol = SparseSymmetricSolver()
    sol.analyse(A1, 0.0)
    sol.factorize()
    c1['factorize_host_seconds'] = round(time.perf_counter() - t0, 3)
    t1 = time.perf_counter()
    chain = sol._device_chain()
    _lib.check(L.rlh_sync())
    c1['device_operators_seconds'] = round(time.perf_counter() - t1, 3)
    info = sol.factors().info
    c1['factor_entries'] = int(info['nnz_l'])
    c1['largest_front'] = int(info['max_front'])
    c1['inertia'] = list(sol.inertia())
    B1, X1 = Vectors(N1 ** 3, 8), Vectors(N1 ** 3, 8)
    B1.fill_random()
    c1['apply_ms_8_vectors'] = round(timed_calls(L, lambda: sol.solve(B1, X1), 5), 3)
    c1['levels'] = [int(v) for v in chain.levels]
    np.random.seed(1)
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(sol, which=6, tol=1e-6, verb=-1)
    exact = lap3d_eigenvalues(N1, N1, N1, 1.0, 1.01, 1.02, 6)
    c1['solve'] = {'seconds': round(time.perf_counter() - t0, 3),
                   'setup_seconds': round(c1['factorize_host_seconds'] + c1['device_operators_seconds'], 3),
                   'setup_what': 'host L D L^H factorisation + device triangular-solve set-up', 'status': int(status),
                   'iterations': int(partial_hevp.last['iterations']),
                   'max_rel_error_vs_closed_form': float(np.max(np.abs(np.sort(lmd)[:6] - exact) / exact))}
    out['config1'] = c1
    del sol, chain, B1, X1
    # ---- config 5 (one GPU: n = 126^3, complex128, m = 64)
    N, m = 126, 64
    n = N ** 3
    H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n)
    op = CsrOperator(H)
    X, Y, W = (Vectors(n, m, data_type=np.complex128) for _ in range(3))
    X.fill_random()
    X.copy(Y)
    code = _lib.dtype_code(np.complex128)
    res = ctypes.c_void_p()
    _lib.check(L.rlh_malloc(ctypes.byref(res), m * m * 16))
    B = n * m * 16
    q = (np.random.default_rng(2).standard_normal((m, m)) / m).astype(np.complex128)
    a1 = np.array([1.0, 0.0])
    legs = {
        'gram': (2 * B, 8.0 * n * m * m, lambda: _lib.check(L.rlh_gram(code, n, m, X.data_ptr(), X.ld(), m, Y.data_ptr(), Y.ld(), res, None))),
        'self_gram': (B, 4.0 * n * m * m, lambda: _lib.check(L.rlh_gram(code, n, m, X.data_ptr(), X.ld(), m, X.data_ptr(), X.ld(), res, None))),
        'multiply': (2 * B, 8.0 * n * m * m, lambda: _lib.check(L.rlh_block_update(code, n, m, X.data_ptr(), X.ld(), m, W.data_ptr(), W.ld(),
                                                                                   _lib.host_ptr(q), m, 1, _lib.host_ptr(a1), 0))),
        'spmm': (H.nnz * 20 + (n + 1) * 4 + 2 * B, 8.0 * H.nnz * m, lambda: op.apply_ptr(m, X.data_ptr(), X.ld(), W.data_ptr(), W.ld())),
    }
    c5 = {'workload': 'Hermitian lap3d + i skew, n=126^3=%d, complex128, m=%d (blocks of %.2f GB)' % (n, m, B / 1e9)}
    for name, (nb, flops, fn) in legs.items():
        t = timed_calls(L, fn, 10)
        c5[name] = {'ms': round(t, 3), 'gbs': round(nb / t / 1e6, 1), 'frac_of_hbm_peak': round(nb / t / 1e6 / HBM_PEAK_GBS, 4),
                    'tflops': round(flops / t / 1e9, 1), 'frac_of_fp64_peak_78.6': round(flops / t / 1e9 / 78.6, 4)}
    _lib.check(L.rlh_free(res))
    del op, X, Y, W
    # one full iteration of the driver on this operator (a direct factorisation of the 126^3 complex operator -- about
    # 10^9 factor entries -- is out of reach for a test box: the end-to-end shift-invert run is covered at 24^3, here
    # the ITERATION is timed at full size): block of 64, no preconditioner, two runs of different length
    from raleigh_amd.core.solver import Options
    ts = {}
    for its in (2, 4, 12, 4, 12):                  # (a first short run takes the allocations; best of two for each length)
        np.random.seed(1)
        opt = Options()
        opt.max_iter, opt.block_size = its, 64
        lmd, x, status = partial_hevp(H, T=True, which=20, tol=1e-6, verb=-1, opt=opt)
        _lib.check(L.rlh_sync())
        t, n_it = float(partial_hevp.last['solve_time']), int(partial_hevp.last['iterations'])
        if its not in ts or t < ts[its][0]:
            ts[its] = (t, n_it)
        del lmd, x
    di = ts[12][1] - ts[4][1]
    c5['iteration_ms'] = round((ts[12][0] - ts[4][0]) / max(di, 1) * 1e3, 2)
    c5['iteration_runs'] = {str(k): {'seconds': round(v[0], 4), 'iterations': v[1]} for k, v in ts.items() if k != 2}
    c5['iteration_what'] = ('block-JCG driver (raleigh_amd/core/solver.py), block of 64 complex128 vectors, n = 126^3, no preconditioner: '
                            'wall time per iteration from runs of %d and %d iterations' % (ts[4][1], ts[12][1]))
    del H
    c5['solve'] = config5_solve(None)
    out['config5'] = c5
    # ---- config 2
    from raleigh_amd.interfaces import pca
    M = Nn = 20000
    r, npc = 400, 200
    rng = np.random.default_rng(1)
    U = rng.standard_normal((M, r)).astype(np.float32)
    U[:, 0] = 1.0
    V = rng.standard_normal((Nn, r)).astype(np.float32)
    U, _ = np.linalg.qr(U)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(min(M, Nn)).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    Am = np.ascontiguousarray((U * s) @ V.T, dtype=np.float32)
    np.random.seed(1)
    t0 = time.perf_counter()
    mean, trans, comps = pca(Am, npc=npc)
    el = time.perf_counter() - t0
    sv = np.linalg.norm(trans, axis=0)
    out['config2'] = {'workload': 'pca of a dense 20000 x 20000 fp32 matrix, 200 components', 'seconds': round(el, 3),
                      'iterations': int(pca.last['iterations']), 'operator_seconds': round(float(pca.last['operator_time']), 3),
                      'max_sigma_error_over_sigma_max': float(np.max(np.abs(sv - s[1:npc + 1])) / s[1])}
    del Am, mean, trans, comps
    # ---- config 4: one GPU's row shard (62 500 of the 500 000 rows) at full size, built on the device from factors
    from raleigh_amd.algebra.hip import Matrix
    from raleigh_amd.algebra.dense_matrix import AMatrix
    M, Nn, r, npc, m = 62500, 40000, 1280, 1000, 128
    rng = np.random.default_rng(4)
    U = rng.standard_normal((M, r)).astype(np.float32)
    U[:, 0] = 1.0
    V = rng.standard_normal((Nn, r)).astype(np.float32)
    U, _ = np.linalg.qr(U)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(Nn).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    rows = Vectors(Nn, M, data_type=np.float32)
    Matrix(np.ascontiguousarray(V)).apply(Vectors(np.ascontiguousarray(U * s)), rows)      # rows = (U s) V^T
    del U, V
    A4 = AMatrix(rows)
    op4 = A4.as_operator()
    x, w = Vectors(Nn, m, data_type=np.float32), Vectors(Nn, m, data_type=np.float32)
    y = Vectors(M, m, data_type=np.float32)
    x.fill_random()

    def pair():
        op4.apply(x, y)
        op4.apply(y, w, transp=True)
    t = timed_calls(L, pair, 5)
    np.random.seed(1)
    t0 = time.perf_counter()
    mean, trans, comps = pca(A4, npc=npc)
    el = time.perf_counter() - t0
    sv = np.linalg.norm(trans, axis=0)
    out['config4_shard'] = {'workload': 'one of the 8 row shards of config 4 at full size: pca of 62500 x 40000 fp32 rows (10 GB, '
                                        'built on the device), %d components' % npc,
                            'dense_pair_ms': round(t, 3), 'dense_pair_tflops': round(4.0 * M * Nn * m / t / 1e9, 1),
                            'dense_pair_frac_of_fp32_mfma_peak_157.3': round(4.0 * M * Nn * m / t / 1e9 / 157.3, 4),
                            'matrix_read_gbs': round(2.0 * M * Nn * 4 / t / 1e6, 1),
                            'seconds': round(el, 3), 'iterations': int(pca.last['iterations']),
                            'operator_seconds': round(float(pca.last['operator_time']), 3),
                            'max_sigma_error_over_sigma_max': float(np.max(np.abs(sv - s[1:npc + 1])) / s[1])}
    return out


def config5_solve(comm, N=126, below=40, block=64, want=20, degree=16, ratio=250.0):
    """BASELINE config 5 end to end: Hermitian complex128 operator (lap3d + i skew, n = N^3), block of 64 vectors, the 20
    eigenpairs nearest a shift with `below` eigenvalues under it, by INEXACT shift-invert (block MINRES with a Chebyshev
    polynomial preconditioner on the device blocks: raleigh_amd/algebra/hip/shift_invert.py), rows sharded over the ranks
    when there are several; eigenvalues against the closed-form spectrum.  (A shift in the lower part of the spectrum: for
    one in the middle of it no polynomial preconditioner exists and no factor fits -- DESIGN.md section 6.)"""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues, lap3d_coefficients
    n, skew = N ** 3, 0.3
    exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02, skew=skew)
    sigma = 0.5 * (exact[below - 1] + exact[below])
    hi = 4.0 * sum(lap3d_coefficients(N, N, N, 1.0, 1.01, 1.02)) + 2 * skew          # Gershgorin
    t0 = time.perf_counter()
    vectors = None
    if comm is None:
        from raleigh_amd.algebra.hip import SparseSymmetricMatrix
        op = SparseSymmetricMatrix(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n, skew=skew))
    else:
        from raleigh_amd.algebra.hip.dist import ShardedSparseMatrix, ShardedVectors, partition
        off = partition(n, comm.size)
        r0, r1 = int(off[comm.rank]), int(off[comm.rank + 1])
        op = ShardedSparseMatrix.from_local_rows(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, r0, r1, skew=skew), r0, n, comm, off)
        vectors = lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm, offsets=off)
    sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=degree, ratio=ratio, hi=hi)
    sol.analyse(op, sigma)
    sol.factorize()
    t_setup = time.perf_counter() - t0
    np.random.seed(1)
    opt = Options()
    opt.block_size, opt.max_iter = block, 100
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(sol, which=want, tol=1e-6, verb=-1, opt=opt, vectors=vectors)
    seconds = time.perf_counter() - t0
    # (the full Lanczos count, outside the timed solve: partial_hevp itself only asks for the signs)
    mk = None if vectors is None else (lambda n_, nv, data_type: vectors(n_, data_type=data_type).new_vectors(nv))
    negative = int(sol.inertia(vectors=mk)[0])
    near = exact[np.argsort(np.abs(exact - sigma))[:want]]
    err = float(max(np.min(np.abs(lmd - e)) / abs(e) for e in near)) if status == 0 and lmd is not None and len(lmd) >= want else None
    last = partial_hevp.last
    return {'problem': 'Hermitian lap3d + i skew, complex128, n = %d^3 = %d, block of %d, the %d eigenpairs nearest sigma = %.4f '
                       '(%d eigenvalues below it), inexact shift-invert: block MINRES + Chebyshev(degree %d on [hi/%g, hi]), '
                       'inner tolerance %.0e, eigenvector tolerance 1e-6, Rayleigh-Ritz with A on the converged vectors'
                       % (N, n, block, want, sigma, below, degree, ratio, sol.tol),
            'seconds': round(seconds, 3), 'setup_seconds': round(t_setup, 3), 'status': int(status),
            'negative_eigenvalues_counted': negative,
            'outer_iterations': int(last['iterations']), 'inner_solves': int(last['inner_solves']),
            'inner_iterations': int(last['inner_iterations']), 'operator_applications_in_vectors': int(last['inner_columns_applied']),
            'max_rel_eigenvalue_error': err}


def config4_sharded(L, comm, rows_per_gpu=62500, Nn=40000, r=1280, npc=1000, m=128):
    """BASELINE config 4 on the sharded path: pca of a (rows_per_gpu x world) x 40 000 fp32 matrix, 1000 components, rows
    sharded over the ranks (at 8 GPUs: the 500 000 x 40 000 of BASELINE.json), every shard built in HBM from factors with
    known singular values; the transposed product's N x k all-reduce runs in column chunks behind the next chunk's GEMM."""
    from raleigh_amd.algebra.hip import Vectors, Matrix
    from raleigh_amd.algebra.hip.dist import ShardedAMatrix
    from raleigh_amd.interfaces import pca
    world, rank, M = comm.size, comm.rank, rows_per_gpu
    rng = np.random.default_rng(4)                           # the right factor and the spectrum: the same on every rank
    V = rng.standard_normal((Nn, r)).astype(np.float32)
    V, _ = np.linalg.qr(V)
    s = np.sort(rng.random(Nn).astype(np.float32)) ** (-0.75)
    s = (s / s[0])[:r]
    rng = np.random.default_rng(100 + rank)                  # this rank's rows of the left factor: orthonormal columns, the
    U = rng.standard_normal((M, r)).astype(np.float32)       # first one constant => stacked: sqrt(world) x orthonormal
    U[:, 0] = 1.0
    U, _ = np.linalg.qr(U)
    rows = Vectors(Nn, M, data_type=np.float32)
    Matrix(np.ascontiguousarray(V)).apply(Vectors(np.ascontiguousarray(U * s)), rows)      # rows = (U s) V^T
    del U, V
    A4 = ShardedAMatrix(rows, comm)
    op4 = A4.as_operator()
    x, w = op4.new_vectors(Nn, m), op4.new_vectors(Nn, m)
    y = op4.new_vectors(M * world, m)
    x.fill_random()

    def pair():
        op4.apply(x, y)
        op4.apply(y, w, transp=True)
    t = max(timed_calls(L, pair, 5), 1e-6)              # (the rehearsal's stand-in timer reads 0)
    np.random.seed(1)
    t0 = time.perf_counter()
    mean, trans, comps = pca(A4, npc=npc)
    el = time.perf_counter() - t0
    sv = np.linalg.norm(trans, axis=0)
    exact = math.sqrt(world) * s[1:npc + 1]
    return {'workload': 'pca of a %d x %d fp32 matrix, %d components, rows sharded over %d GPU(s) (%d rows = %.1f GB per GPU, '
                        'built on the device)' % (M * world, Nn, npc, world, M, M * Nn * 4 / 1e9),
            'dense_pair_ms': round(t, 3), 'dense_pair_tflops_per_gpu': round(4.0 * M * Nn * m / t / 1e9, 1),
            'dense_pair_frac_of_fp32_mfma_peak_157.3': round(4.0 * M * Nn * m / t / 1e9 / 157.3, 4),
            'allreduce_bytes_per_product': Nn * m * 4, 'allreduce_chunks': int(op4.reduce_chunks),
            'seconds': round(el, 3), 'iterations': int(pca.last['iterations']),
            'operator_seconds': round(float(pca.last['operator_time']), 3),
            'max_sigma_error_over_sigma_max': float(np.max(np.abs(sv - exact)) / exact[0])}


def launch_ranks(n):
    """Runs this script under torch.distributed.run with n ranks on this node (one per GPU, rendezvous on 127.0.0.1 at a
    free port) and prints the one JSON line its rank 0 wrote; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    # (the launcher would otherwise pin every rank's host BLAS to ONE thread: the ranks build their operators on the host)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for l in r.stdout.decode(errors='replace').splitlines():
        if l.startswith('{') and l.rstrip().endswith('}'):
            line = l
    if line is not None:
        sys.stdout.write(line + '\n')
        sys.stdout.flush()
    else:
        sys.stderr.write(r.stdout.decode(errors='replace')[-2000:])
    return r.returncode if (r.returncode != 0 or line is not None) else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--side', type=int, default=215, help='lap3d side (n = side^3)')
    ap.add_argument('--m', type=int, default=32)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the config legs (2 / 3 / 5 / one config-4 shard at N = 1, the '
                                                            'row-sharded config 4 on the sharded path)')
    ap.add_argument('--force-dist', action='store_true',
                    help='use the sharded (torch.distributed / RCCL) code path even with one rank, all-reduce and '
                         'halo exchange included (a rank then exchanges with itself)')
    ap.add_argument('--solve-side', type=int, default=215,
                    help='lap3d side of the end-to-end "seconds to 10 eigenpairs" run (0: skip)')
    ap.add_argument('--ilu-large-side', type=int, default=160,
                    help='lap3d side of the GPU-only solve with the reference\'s ILU preconditioner (0: skip)')
    ap.add_argument('--ilu-side', type=int, default=64,
                    help='lap3d side of the GPU-vs-CPU solve with the reference\'s ILU preconditioner (0: skip)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` as the driver types it for N = 1: start the N ranks ourselves -- as a CHILD process,
        # before anything here has touched the GPU (this process never does) -- and relay rank 0's JSON line and the exit code
        sys.exit(launch_ranks(args.gpus))

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
    from raleigh_amd.algebra.hip import Vectors
    # rehearsal of the multi-rank control flow on a machine without GPUs (tests/test_bench_rehearsal.py): the
    # C-ABI stand-in of the test tier on host memory and the gloo backend; never set on a GPU box
    rehearsal = os.environ.get('RLH_BENCH_REHEARSAL') == '1'
    if rehearsal:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import fake_lib
        fake_lib.install()
    comm = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        from raleigh_amd.algebra.hip.dist import Comm, ShardedVectors, partition
        comm = Comm(force_collectives=True if args.force_dist else None)
        # (forced at one rank: the rows a real pair of neighbours would trade -- two planes of the grid -- not half the shard)
        comm.forced_halo_rows = 2 * args.side * args.side
    L = _lib.lib(local_rank)
    side, m, es = args.side, args.m, 8

    def sync_all():
        _lib.check(L.rlh_sync())
        if comm is not None:
            import torch
            if not rehearsal:
                torch.cuda.synchronize()
            comm.barrier()
            if not rehearsal:
                torch.cuda.synchronize()

    def max_over_ranks(x):
        if comm is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=comm.device)
        comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MAX)
        return float(t.item())

    def run_mode(mode, steps, warmup, extras):
        """Times `steps` headline iterations on lap3d side x side x nzg (weak: nzg = side * world)."""
        nzg = side * world if mode == 'weak' else side
        n = side * side * nzg
        nnz = 7 * n - 2 * (side * nzg * 2 + side * side)      # 7-point stencil minus the six faces
        off = None
        if comm is None:
            nloc = n
            mk = lambda: Vectors(nloc, m, data_type=np.float64)
        else:
            off = partition(n, world)
            nloc = int(off[rank + 1] - off[rank])
            mk = lambda: ShardedVectors(n, m, np.float64, comm=comm, offsets=off)
        # ---- synthetic inputs, resident in HBM before the timed region
        rng = np.random.default_rng(1 + rank)
        blocks = [mk() for _ in range(7)]
        for j in range(m):                            # U(-1, 1), one vector at a time (bounded host memory)
            blocks[0].select(1, j)
            Vectors.fill(blocks[0], 2 * rng.random((1, nloc)) - 1)
        blocks[0].select(m)
        for i, b in enumerate(blocks[1:], 1):         # distinct random-looking blocks from device-side ops
            blocks[0].copy(b, np.roll(np.arange(m), i))
            b.add(blocks[0], 0.37 * i)
        op = lap_operator(side, np.float64, comm, off, nz=nzg)
        it = InnerIteration(blocks, op)
        for _ in range(warmup):
            it.headline()
        sync_all()
        ms = ctypes.c_float()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            _lib.check(L.rlh_timer_start())
            it.headline()
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            per_step.append(ms.value)
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ms_per_step = elapsed / steps * 1e3
        nbytes, parts = InnerIteration.headline_bytes(n, m, es, nnz)
        res = {'mode': mode, 'n': n, 'nnz': nnz, 'nzg': nzg, 'nloc': nloc, 'ms_per_step': ms_per_step,
               'median_ms': max_over_ranks(float(np.median(per_step))), 'nbytes': nbytes, 'parts': parts,
               'value': nbytes / (ms_per_step * 1e-3) / 1e9}
        if extras:
            # the same quantities as the repository's own driver asks for them: stacked Grams, 5 round trips
            it.fused()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                it.fused()
            sync_all()
            f_ms = max_over_ranks(time.perf_counter() - t0) / steps * 1e3
            fb = InnerIteration.fused_bytes(n, m, es, nnz)
            res['fused'] = {'ms_per_step': round(f_ms, 4), 'host_round_trips_per_step': 5,
                            'algorithmic_bytes_per_step': fb, 'gbs': round(fb / (f_ms * 1e-3) / 1e9, 1),
                            'what': 'the reductions of one iteration as raleigh_amd/core/solver.py issues them: stacked '
                                    'Grams [AX|X]^H X, [AZ|Z]^H Y, [X|Y]^H Y, [X|Y]^H AY (every block read once) in 5 batches'}
            # the all-ops figure beside the headline (reported, not the metric): a few steps of the full mix
            q = np.eye(m) + 1e-3 * np.random.default_rng(7).standard_normal((m, m))
            it.all_ops(q)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(5):
                it.all_ops(q)
            sync_all()
            all_ms = max_over_ranks(time.perf_counter() - t0) / 5 * 1e3
            all_bytes = InnerIteration.all_ops_bytes(n, m, es, nnz)
            res['all_ops'] = {'ms_per_step': round(all_ms, 3), 'algorithmic_bytes_per_step': all_bytes,
                              'gbs': round(all_bytes / (all_ms * 1e-3) / 1e9, 1),
                              'what': 'headline + 4 multiply + 7 add(q) + 6 copy + 1 scale (the reference driver\'s per-iteration mix)'}
            # roofline of the dominant kernel (two-operand Gram), HIP events on the kernels' stream
            X, AX = blocks[0], blocks[1]
            rbuf = ctypes.c_void_p()
            _lib.check(L.rlh_malloc(ctypes.byref(rbuf), m * m * es))
            code = _lib.dtype_code(np.float64)
            gram = lambda: L.rlh_gram(code, nloc, m, X.data_ptr(), X.ld(), m, AX.data_ptr(), AX.ld(), rbuf, None)
            reps = 20
            _lib.check(gram())
            _lib.check(L.rlh_timer_start())
            for _ in range(reps):
                _lib.check(gram())
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            _lib.check(L.rlh_free(rbuf))
            res['gram_ms'] = ms.value / reps
            res['gram_bytes'] = 2 * nloc * m * es
            # device-copy ceiling of this box ("first measure a device-copy ceiling on the box and
            # report both"): one block copied onto another, read + write bytes, same event timing
            W = blocks[6]
            cp = lambda: L.rlh_copy(code, nloc, m, X.data_ptr(), X.ld(), W.data_ptr(), W.ld())
            _lib.check(cp())
            _lib.check(L.rlh_timer_start())
            for _ in range(reps):
                _lib.check(cp())
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            res['copy_gbs'] = 2 * nloc * m * es / (max(ms.value, 1e-9) / reps * 1e-3) / 1e9
        return res

    main_res = run_mode(args.scaling, args.steps, args.warmup, True)
    also = None
    if world > 1:
        other = 'strong' if args.scaling == 'weak' else 'weak'
        r2 = run_mode(other, max(3, args.steps // 2), 2, False)
        also = {'scaling': other, 'value': round(r2['value'], 1), 'unit': 'GB/s', 'ms_per_step': round(r2['ms_per_step'], 4),
                'n': r2['n'], 'rows_per_gpu': r2['nloc']}

    achieved = main_res['gram_bytes'] / (max(main_res['gram_ms'], 1e-9) * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'gram_traffic.json')
    if os.path.exists(tpath) and world == 1:
        try:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    roofline = {'bound': 'hbm', 'kernel': 'gram_stream_kernel<double, 256, 32> (X.dot(Y), m=k=%d)' % m,
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'traffic_from_profile': traffic,
                'traffic_note': 'PMC counters cannot be read inside this process: hbm_bytes_per_launch of the same kernel on the same '
                                'shape from profiles/gram_traffic.json (rocprofv3 --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE)',
                'algorithmic_bytes_per_launch': main_res['gram_bytes'], 'avg_launch_ms': round(main_res['gram_ms'], 4),
                'inner_iteration_frac': round(main_res['value'] / world / HBM_PEAK_GBS, 4),
                'device_copy_ceiling_gbs': round(main_res.get('copy_gbs', 0.0), 1),
                'frac_of_copy_ceiling': round(achieved / max(main_res.get('copy_gbs', 0.0), 1e-9), 4)}
    n, nnz, nzg = main_res['n'], main_res['nnz'], main_res['nzg']
    out = {'metric': 'inner-iter GB/s vs HBM roofline (Gram+dots+SpMM of one block-JCG iteration)',
           'value': round(main_res['value'], 1), 'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(main_res['ms_per_step'], 4), 'higher_is_better': True,
           'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': 'block-JCG inner iteration: 9 Gram + 4 dots + 1 SpMM, n=%dx%dx%d=%d rows '
                                  '(%d per GPU), m=%d, 7-pt Laplacian nnz=%d, rows sharded over %d GPU(s), %s scaling'
                                  % (side, side, nzg, n, main_res['nloc'], m, nnz, world, args.scaling),
                      'n': n, 'm': m, 'nnz': nnz, 'algorithmic_bytes_per_step': main_res['nbytes'],
                      'bytes_breakdown': main_res['parts']},
           'ms_per_step_median_hip_events': round(main_res['median_ms'], 4),
           'roofline': roofline, 'fused': main_res['fused'], 'all_ops': main_res['all_ops']}
    if also is not None:
        out['also'] = also
    if comm is not None:
        # one reduction round trip as every dot / dots makes it: all-reduce of an m x m block on the kernels' stream + the
        # fetch of the result to the host (no kernel in front of it)
        buf = comm.reduction_buffer(m * m * es)
        comm.allreduce_from_device(buf, np.float64, m * m)
        trips = []
        for _ in range(20):
            t0 = time.perf_counter()
            comm.allreduce_from_device(buf, np.float64, m * m)
            trips.append(time.perf_counter() - t0)
        out['collectives'] = {'backend': 'gloo (rehearsal)' if rehearsal else 'nccl (RCCL)', 'forced_at_one_rank': bool(args.force_dist),
                              'round_trips_per_step': {'headline': 13, 'fused': 5},
                              'round_trip_us': round(max_over_ranks(float(np.median(trips))) * 1e6, 1),
                              'small_reductions': ('shared memory of the node (rlh_shm_allreduce: partial fetched, slots summed in rank order)'
                                                   if getattr(comm, '_shm', None) is not None else 'the backend all-reduce on the stream'),
                              'what': 'median wall time of one reduction of an m x m fp64 block over the ranks + the result on the host, max over ranks'}

    def guarded(key, fn):
        """A failure on ANY rank is agreed on by all of them (the others would otherwise wait in the next
        collective); the line reports it and the process exits non-zero after printing."""
        err = None
        try:
            val = fn()
        except Exception as e:
            val, err = None, '%s: %s' % (type(e).__name__, e)
        failed = max_over_ranks(1.0 if err else 0.0)
        if failed:
            out[key] = {'error': err or 'failed on another rank'}
            return False
        out[key] = val
        return True

    ok = True
    if args.solve_side > 0:
        ok = guarded('solve', lambda: solve_ten(args.solve_side, comm)) and ok
    if comm is not None and not args.no_configs:
        if rehearsal:        # (the control flow of the leg at toy size)
            ok = guarded('config4', lambda: config4_sharded(L, comm, rows_per_gpu=192, Nn=96, r=24, npc=8, m=8)) and ok
        else:
            ok = guarded('config4', lambda: config4_sharded(L, comm)) and ok
        if rehearsal:
            ok = guarded('config5', lambda: config5_solve(comm, N=8, below=6, block=16, want=6, degree=6, ratio=20.0)) and ok
        else:
            ok = guarded('config5', lambda: config5_solve(comm)) and ok
    if world == 1 and comm is None:
        if args.ilu_side > 0:
            ok = guarded('solve_ilu', lambda: solve_ilu_pair(args.ilu_side)) and ok
        if args.ilu_large_side > 0:
            ok = guarded('solve_ilu_large', lambda: solve_ilu_large(args.ilu_large_side)) and ok
        if not args.no_configs:
            ok = guarded('configs', lambda: config_legs(L)) and ok
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(side, m)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if comm is not None:
        comm.dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == '__main__':
    main()
