"""Per-kernel microbenchmark on the GPU box: algorithmic GB/s of each hot-path op
(SURVEY 8d byte counts), HIP-event timed through rlh_timer_start/stop."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=10_000_000)
    ap.add_argument('--m', type=int, default=32)
    ap.add_argument('--dtype', default='d')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--lap', type=int, default=0, help='lap3d side N (n = N^3) for the SpMM line')
    ap.add_argument('--only', default='')
    ap.add_argument('--lap2d', type=int, default=0, help='2-D Laplacian side (n = N^2), SpMM line only')
    ap.add_argument('--band', type=int, default=-1, help='banded test matrix with diagonals -band..band (needs --n)')
    ap.add_argument('--fe', action='store_true', help='SpMM on the FE-like shipsec5 surrogate (BASELINE config 3; n = 179860)')
    ap.add_argument('--herm', type=int, default=0, help='SpMM on the Hermitian lap3d + i skew operator of side N (config 5; use --dtype z)')
    args = ap.parse_args()
    from raleigh_amd import _lib
    from raleigh_amd.algebra.hip import Vectors, SparseSymmetricMatrix
    L = _lib.lib()
    dt = {'s': np.float32, 'd': np.float64, 'c': np.complex64, 'z': np.complex128}[args.dtype]
    es = np.dtype(dt).itemsize
    n, m = args.n, args.m
    if args.lap:
        n = args.lap ** 3
    if args.lap2d:
        n = args.lap2d ** 2
    if args.fe:
        n = 179860
    if args.herm:
        n = args.herm ** 3
    code = _lib.dtype_code(dt)
    X, Y, W = Vectors(n, m, data_type=dt), Vectors(n, m, data_type=dt), Vectors(n, m, data_type=dt)
    # device-side fill: upload one random column block and replicate (host RNG for 10^7 x 32 is slow)
    rng = np.random.default_rng(1)
    col = (2 * rng.random((1, n)) - 1).astype(dt)
    for V in (X, Y, W):
        for j in range(m):
            V.select(1, j)
            V.fill(np.roll(col, j + 1, axis=1) if j < 4 else col * (1 + 0.01 * j))
        V.select(m)
    q = rng.standard_normal((m, m)).astype(dt)
    B = n * m * es
    ms = ctypes.c_float()

    def timed(fn, reps=args.reps):
        fn()
        _lib.check(L.rlh_sync())
        ts = []
        for _ in range(reps):
            _lib.check(L.rlh_timer_start())
            fn()
            _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
            ts.append(ms.value)
        return float(np.median(ts)), float(np.min(ts))

    g = np.zeros((m, m), dtype=dt)
    d = np.zeros((m,), dtype=dt)
    res_d = ctypes.c_void_p()
    _lib.check(L.rlh_malloc(ctypes.byref(res_d), m * m * es))
    a1 = np.array([1.0, 0.0])
    hq = _lib.host_ptr(q)
    ops = [
        ('gram X.dot(Y)', 2 * B, lambda: L.rlh_gram(code, n, m, X.data_ptr(), X.ld(), m, Y.data_ptr(), Y.ld(), res_d, None)),
        ('gram X.dot(X)', B, lambda: L.rlh_gram(code, n, m, X.data_ptr(), X.ld(), m, X.data_ptr(), X.ld(), res_d, None)),
        ('dots X.dots(Y)', 2 * B, lambda: L.rlh_dots(code, n, m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), res_d, None)),
        ('dots X.dots(X)', B, lambda: L.rlh_dots(code, n, m, X.data_ptr(), X.ld(), X.data_ptr(), X.ld(), res_d, None)),
        ('multiply W=X*Q', 2 * B, lambda: L.rlh_block_update(code, n, m, X.data_ptr(), X.ld(), m, W.data_ptr(), W.ld(), hq, m, 1, _lib.host_ptr(a1), 0)),
        ('add W+=X*Q', 3 * B, lambda: L.rlh_block_update(code, n, m, X.data_ptr(), X.ld(), m, W.data_ptr(), W.ld(), hq, m, 1, _lib.host_ptr(a1), 1)),
        ('axpy W+=a*X', 3 * B, lambda: L.rlh_axpy(code, n, m, _lib.host_ptr(a1), X.data_ptr(), X.ld(), W.data_ptr(), W.ld())),
        ('copy X->W', 2 * B, lambda: L.rlh_copy(code, n, m, X.data_ptr(), X.ld(), W.data_ptr(), W.ld())),
        ('scale W', 2 * B, lambda: L.rlh_scale_cols(code, n, m, _lib.host_ptr(np.full(2 * m, 1.0)), 1, W.data_ptr(), W.ld())),
    ]
    print('n=%d m=%d dtype=%s block=%.3f GB' % (n, m, args.dtype, B / 1e9))
    for name, nbytes, fn in ops:
        if args.only and args.only not in name:
            continue
        med, mn = timed(fn)
        print('%-18s %8.3f ms (min %8.3f)  %8.1f GB/s  %5.1f%% of 8 TB/s' % (name, med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80))
    if args.band >= 0:
        import scipy.sparse as sp
        offs = list(range(-args.band, args.band + 1))
        A = sp.diags([np.full(n - abs(o), 1.0 + 0.1 * o) for o in offs], offs, shape=(n, n), format='csr').astype(dt)
        op = SparseSymmetricMatrix(A)
        nbytes = A.nnz * (es + 4) + (n + 1) * 4 + 2 * B
        med, mn = timed(lambda: op.apply(X, W))
        print('%-18s %8.3f ms (min %8.3f)  %8.1f GB/s  %5.1f%% of 8 TB/s' % ('spmm band %d' % args.band, med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80))
    if args.fe or args.herm:
        from raleigh_amd.synthetic import fe_surrogate, hermitian_lap3d_rows
        from raleigh_amd.algebra.hip import CsrOperator
        t0 = time.time()
        if args.fe:
            A = fe_surrogate().astype(dt)
        else:
            A = hermitian_lap3d_rows(args.herm, args.herm, args.herm, 1.0, 1.01, 1.02, 0, n).astype(dt)
        op = CsrOperator(A)
        print('%s setup %.1f s, nnz=%d (%.1f per row), layout %s' % ('fe surrogate' if args.fe else 'hermitian lap3d',
              time.time() - t0, A.nnz, A.nnz / n, op.layout()))
        nbytes = A.nnz * (es + 4) + (n + 1) * 4 + 2 * B
        fn = lambda: op.apply_ptr(m, X.data_ptr(), X.ld(), W.data_ptr(), W.ld())
        med, mn = timed(fn, reps=max(args.reps, 30))
        print('%-18s %8.4f ms (min %8.4f)  %8.1f GB/s  %5.1f%% of 8 TB/s' % ('spmm ' + ('fe' if args.fe else 'herm'), med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80))
        # back-to-back launches (one event pair around 20 applications): what a solver sees
        fn(); _lib.check(L.rlh_sync())
        _lib.check(L.rlh_timer_start())
        for _ in range(20):
            fn()
        _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
        print('%-18s %8.4f ms per application, 20 back to back  %8.1f GB/s' % ('', ms.value / 20, nbytes / (ms.value / 20) / 1e6))
    if args.lap or args.lap2d:
        from oracle.sparse import lap3d
        t0 = time.time()
        A = lap3d(args.lap, args.lap, args.lap, 1.0, 1.01, 1.02) if args.lap else lap3d(args.lap2d, args.lap2d, 1, 1.0, 1.01, 1.02)
        op = SparseSymmetricMatrix(A.astype(dt))
        print('lap3d setup %.1f s, nnz=%d' % (time.time() - t0, A.nnz))
        nbytes = A.nnz * (es + 4) + (n + 1) * 4 + 2 * B
        med, mn = timed(lambda: op.apply(X, W))
        print('%-18s %8.3f ms (min %8.3f)  %8.1f GB/s  %5.1f%% of 8 TB/s' % ('spmm lap3d', med, mn, nbytes / med / 1e6, nbytes / med / 1e6 / 80))
    if args.only:
        return
    # host-visible latency of a synchronising Gram
    t0 = time.perf_counter()
    for _ in range(20):
        X.dot(Y)
    print('X.dot(Y) host wall per call: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))


if __name__ == '__main__':
    main()
