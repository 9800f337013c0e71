"""A/B of the windowed SpMM's layouts on one matrix (GPU box): the 1024-row blocks, the stacked blocks with
register staging and the stacked blocks with LDS-DMA staging -- same handle, same vectors, the variant chosen per call
through the environment (RLH_SPMM_STACK / RLH_SPMM_STACK_DMA are read at launch time).  Checks that the three produce
the same block and prints HIP-event times.
    python tools/stack_bench.py --lap 215 [--dtype d] [--m 32] [--check]      (--lap2d N, --band K --n N as microbench.py)"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lap', type=int, default=0)
    ap.add_argument('--lap2d', type=int, default=0)
    ap.add_argument('--herm', type=int, default=0, help='Hermitian lap3d + i skew of side N (BASELINE config 5; --dtype z or c)')
    ap.add_argument('--band', type=int, default=-1)
    ap.add_argument('--n', type=int, default=9_938_375)
    ap.add_argument('--m', type=int, default=32)
    ap.add_argument('--dtype', default='d')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--dbg', action='store_true', help='time the timing-only builds of the LDS-DMA kernel (fp64, lap3d 215)')
    ap.add_argument('--check', action='store_true', help='compare with scipy on the host (small sizes)')
    args = ap.parse_args()
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    from oracle.sparse import lap3d
    import scipy.sparse as sp
    L = _lib.lib()
    dt = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}[args.dtype]
    es = np.dtype(dt).itemsize
    t0 = time.time()
    if args.herm:
        from raleigh_amd.synthetic import hermitian_lap3d_rows
        A = hermitian_lap3d_rows(args.herm, args.herm, args.herm, 1.0, 1.01, 1.02, 0, args.herm ** 3)
    elif args.lap:
        A = lap3d(args.lap, args.lap, args.lap, 1.0, 1.01, 1.02)
    elif args.lap2d:
        A = lap3d(args.lap2d, args.lap2d, 1, 1.0, 1.01, 1.02)
    else:
        k = args.band
        A = sp.diags([np.full(args.n - abs(d), 1.0 / (1 + abs(d))) for d in range(-k, k + 1)], list(range(-k, k + 1)), format='csr')
    A = sp.csr_matrix(A.astype(dt))
    n, m = A.shape[0], args.m
    os.environ['RLH_SPMM_STACK'] = os.environ.get('RLH_SPMM_STACK', '1')
    op = SparseSymmetricMatrix(A)
    lay, _, _, stacks, s0, s1 = op.layout()
    print('n = %d, nnz = %d, setup %.1f s: layout %s, %d stacks, staged per row %.3f -> %.3f' %
          (n, A.nnz, time.time() - t0, lay, stacks, s0, s1), flush=True)
    rng = np.random.default_rng(1)
    X, Y = Vectors(n, m, data_type=dt), Vectors(n, m, data_type=dt)
    if n * m <= 40_000_000:
        x = rng.standard_normal((m, n)).astype(dt)
        if args.dtype in 'cz':
            x = x + 1j * rng.standard_normal((m, n)).astype(dt)
        X.fill(x)
    else:
        col = (2 * rng.random((1, n)) - 1).astype(dt)
        if args.dtype in 'cz':
            col = col + 1j * (2 * rng.random((1, n)) - 1).astype(dt)
        for j in range(m):
            X.select(1, j)
            X.fill(np.roll(col, 7 * j + 1, axis=1) * (1 + 0.01 * j))
        X.select(m)
        x = None
    ms = ctypes.c_float()
    nbytes = A.nnz * (es + 4) + (n + 1) * 4 + 2 * n * m * es

    def timed(reps=args.reps):
        op.apply(X, Y)
        _lib.check(L.rlh_sync())
        ts = []
        for _ in range(reps):
            _lib.check(L.rlh_timer_start())
            op.apply(X, Y)
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            ts.append(ms.value)
        return float(np.median(ts)), float(np.min(ts))

    results = {}
    variants = [('blocks', {'RLH_SPMM_STACK': '0'})]
    if stacks and args.dtype in 'cz':
        variants += [('stacks, LDS-DMA', {'RLH_SPMM_STACK': '1'})]
    elif stacks:
        variants += [('stacks, register staging', {'RLH_SPMM_STACK': '1', 'RLH_SPMM_STACK_DMA': '0'}),
                     ('stacks, LDS-DMA', {'RLH_SPMM_STACK': '1', 'RLH_SPMM_STACK_DMA': '2', 'RLH_SPMM_STACK_DBG': '0'})]
        if args.dbg:
            variants += [('stacks, LDS-DMA, plain stores', {'RLH_SPMM_STACK': '1', 'RLH_SPMM_STACK_DMA': '2', 'RLH_SPMM_STACK_DBG': '8'}),
                         ('stacks, LDS-DMA, stores counted', {'RLH_SPMM_STACK': '1', 'RLH_SPMM_STACK_DMA': '2', 'RLH_SPMM_STACK_DBG': '16'})]
    # the variants take turns (one call each per round): the first calls of a process run up to 8 % slower than the
    # later ones, which a variant-after-variant comparison books to whoever goes first
    times = {name: [] for name, _ in variants}
    for rep in range(args.reps + 2):
        for name, env in variants:
            os.environ.update(env)
            _lib.check(L.rlh_timer_start())
            op.apply(X, Y)
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            if rep >= 2:
                times[name].append(ms.value)
            if rep == 0:
                results[name] = Y.data().copy() if n * m <= 40_000_000 else Y.data()[:, ::max(1, n // 200_000)].copy()
    os.environ['RLH_SPMM_STACK_DBG'] = '0'
    for name, _ in variants:
        med, mn = float(np.median(times[name])), float(np.min(times[name]))
        print('%-28s %8.4f ms (min %8.4f)  %8.1f GB/s  %5.1f%% of 8 TB/s' % (name, med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80), flush=True)
    if args.dbg and stacks:
        what = {1: 'no DMA', 2: 'no LDS reads', 3: 'no DMA, no LDS reads', 4: 'no stores', 5: 'no DMA, no stores',
                6: 'no LDS reads, no stores', 7: 'nothing but the entries, waits and barriers'}
        for d in range(1, 8):
            os.environ['RLH_SPMM_STACK_DBG'] = str(d)
            med, mn = timed()
            print('  timing-only build %d (%s): %8.4f ms (min %8.4f)' % (d, what[d], med, mn), flush=True)
        os.environ['RLH_SPMM_STACK_DBG'] = '0'
    base = results['blocks']
    for name, y in results.items():
        if name != 'blocks' and y.shape == base.shape:
            print('%-28s max |difference to blocks| = %.3e (max |y| = %.3e)' % (name, float(np.max(np.abs(y - base))), float(np.max(np.abs(base)))))
    if args.check and x is not None:
        ref = (A @ x.T.astype(np.complex128 if args.dtype in 'cz' else np.float64)).T
        for name, y in results.items():
            print('%-28s relative error vs scipy = %.3e' % (name, float(np.linalg.norm(y - ref) / np.linalg.norm(ref))))


if __name__ == '__main__':
    main()
