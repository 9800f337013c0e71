"""BASELINE config 2: pca() on a synthetic dense M x N fp32 matrix, npc components, 1 MI355X.
Reports wall time, iterations, the dense-apply (A x / A^T x) TFLOP/s against the fp32 MFMA
peak (157.3 TF) and the singular-value error against the generator's exact values."""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=20000)
ap.add_argument('--N', type=int, default=20000)
ap.add_argument('--rank', type=int, default=400)
ap.add_argument('--npc', type=int, default=200)
ap.add_argument('--gemm-only', action='store_true')
ap.add_argument('--profile', action='store_true')
ap.add_argument('--no-check', action='store_true')
a = ap.parse_args()
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, Matrix
L = _lib.lib()
M, N, r = a.M, a.N, a.rank
rng = np.random.default_rng(1)
t0 = time.time()
# low-rank-plus-structure data generated from factors (QR of the full generator is too slow at 20k)
U = rng.standard_normal((M, r)).astype(np.float32); U[:, 0] = 1.0
V = rng.standard_normal((N, r)).astype(np.float32)
U, _ = np.linalg.qr(U); V, _ = np.linalg.qr(V)
s = np.sort(rng.random(min(M, N)).astype(np.float32)) ** (-0.75); s = (s / s[0])[:r]
A = np.ascontiguousarray((U * s) @ V.T, dtype=np.float32)
print('generated %dx%d rank %d in %.1fs' % (M, N, r, time.time() - t0))
m = 128
Am = Matrix(A)
x = Vectors(rng.standard_normal((m, N)).astype(np.float32)); y = Vectors(M, m, data_type=np.float32); w = Vectors(N, m, data_type=np.float32)
ms = ctypes.c_float()
for transp, (src, dst) in ((False, (x, y)), (True, (y, w))):
    Am.apply(src, dst, transp)
    _lib.check(L.rlh_sync())
    for _ in range(10):             # (the first calls after an idle period run at the idle clock: see tools/gemm_shapes.py)
        Am.apply(src, dst, transp)
    _lib.check(L.rlh_timer_start())
    for _ in range(20):
        Am.apply(src, dst, transp)
    _lib.check(L.rlh_timer_stop(ctypes.byref(ms)))
    t = ms.value / 20
    print('dense apply transp=%d m=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)' % (transp, m, t, 2.0 * M * N * m / t / 1e9, 2.0 * M * N * m / t / 1e9 / 157.3 * 100))
ref = (x.data().astype(np.float64) @ A.astype(np.float64).T)[:4, :200]
Am.apply(x, y)
print('apply rel err vs fp64 host: %.2e' % (np.linalg.norm(y.data()[:4, :200] - ref) / np.linalg.norm(ref)))
if not a.gemm_only:
    from raleigh_amd.interfaces import pca
    np.random.seed(1)
    if a.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    t0 = time.time()
    mean, trans, comps = pca(A, npc=a.npc)
    el = time.time() - t0
    if a.profile:
        pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(22)
    sv = np.linalg.norm(trans, axis=0)
    # exact singular values of the shifted matrix: A - e*mean removes the constant left vector
    print('pca npc=%d: %.2f s, iterations %d, operator time %.2f s' % (a.npc, el, pca.last['iterations'], pca.last['operator_time']))
    if a.no_check:
        sys.exit(0)
    As = A - A.mean(axis=0, keepdims=True)
    G = (As.T @ As).astype(np.float64) if N <= M else (As @ As.T).astype(np.float64)
    ex = np.sqrt(np.abs(np.linalg.eigvalsh(G)[::-1][:a.npc]))
    print('max |sigma - exact| / sigma_max = %.2e' % (np.max(np.abs(sv[:a.npc] - ex)) / ex[0]))
