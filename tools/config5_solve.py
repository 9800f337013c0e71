"""BASELINE config 5 end to end on one GPU: Hermitian lap3d + i skew, complex128, n = N^3 (default 126^3 = 2 000 376), block of
64 vectors, the 20 eigenpairs nearest a shift with K eigenvalues below it, by INEXACT shift-invert (block MINRES with a
Chebyshev polynomial preconditioner: raleigh_amd/algebra/hip/shift_invert.py).  Eigenvalues against the closed-form spectrum.

    python tools/config5_solve.py [N] [K] [degree] [ratio] [inner tol] [outer tol]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.interfaces import partial_hevp
from raleigh_amd.core.solver import Options
from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
from raleigh_amd.algebra.hip import synchronize
from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues
arg = lambda i, d, t=float: t(sys.argv[i]) if len(sys.argv) > i else d
N, K, degree, ratio = arg(1, 126, int), arg(2, 40, int), arg(3, 16, int), arg(4, 250.0)
itol, otol = arg(5, 1e-10), arg(6, 1e-6)
low = arg(7, 0, int)
t0 = time.time()
H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
sigma = 0.5 * (exact[K - 1] + exact[K])
print('matrix and spectrum: %.1f s; sigma = %.6f between eigenvalues %d and %d (gap %.3f)' % (time.time() - t0, sigma, K, K + 1, exact[K] - exact[K - 1]), flush=True)
np.random.seed(1)
opt = Options(); opt.block_size = 64; opt.max_iter = 60
sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, tol=itol, degree=degree, ratio=ratio)
t0 = time.time()
lmd, x, status = partial_hevp(H, sigma=sigma, which=20, tol=otol, verb=0, opt=opt, solver=sol)
synchronize()
el = time.time() - t0
print('status %d, %.2f s in all, solve %.2f s, %d outer iterations, %d inner solves, %d block MINRES steps, %d vectors through the operator'
      % (status, el, partial_hevp.last['solve_time'], partial_hevp.last['iterations'], sol.solves, sol.iterations, sol.columns_applied))
near = exact[np.argsort(np.abs(exact - sigma))[:20]]
err = [np.min(np.abs(lmd - e)) / abs(e) for e in near]
print('%d eigenvalues returned; the 20 nearest the shift: max relative error %.2e' % (len(lmd), max(err)))
r = H @ x - x * lmd
print('max residual / max |lambda|: %.2e' % (np.max(np.linalg.norm(r, axis=0)) / np.max(np.abs(exact))))
