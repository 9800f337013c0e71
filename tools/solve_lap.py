"""Seconds to k eigenpairs of the 3-D Laplacian on the GPU (no preconditioner: every block
stays in HBM), checked against the analytic spectrum."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--side', type=int, default=60)
ap.add_argument('--k', type=int, default=10)
ap.add_argument('--tol', type=float, default=1e-6)
ap.add_argument('--block', type=int, default=-1)
ap.add_argument('--maxit', type=int, default=3000)
ap.add_argument('--verb', type=int, default=-1)
ap.add_argument('--profile', action='store_true')
ap.add_argument('--cheb', type=int, default=0, help='degree of the device Chebyshev preconditioner (0: none)')
ap.add_argument('--ratio', type=float, default=100.0)
ap.add_argument('--low', action='store_true', help='evaluate the preconditioner in float32')
ap.add_argument('--bf16', action='store_true', help='with --low: keep the preconditioner work blocks in bfloat16')
a = ap.parse_args()
from raleigh_amd.interfaces import partial_hevp
from raleigh_amd.core.solver import Options
from oracle.sparse import lap3d, lap3d_eigenvalues
N = a.side
t0 = time.time()
A = lap3d(N, N, N, 1.0, 1.01, 1.02)
print('n=%d nnz=%d matrix build %.1fs' % (A.shape[0], A.nnz, time.time() - t0))
np.random.seed(1)
opt = Options(); opt.max_iter = a.maxit; opt.block_size = a.block
t0 = time.time()
if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
T = True
if a.cheb > 0:
    from raleigh_amd.algebra.hip import SparseSymmetricMatrix
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    low = SparseSymmetricMatrix(A.astype(np.float32)) if a.low else None
    T = ChebyshevPreconditioner(None if a.low else SparseSymmetricMatrix(A), gershgorin_upper_bound(A), ratio=a.ratio, degree=a.cheb,
                                low_precision_op=low, storage='bf16' if a.bf16 else None)
lmd, x, status = partial_hevp(A, T=T, which=a.k, tol=a.tol, verb=a.verb, opt=opt)
if a.profile:
    pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(35)
el = time.time() - t0
ana = lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02, a.k)
print('status %d, %d eigenvalues, iterations %d, total %.2fs (solve %.2fs)' % (status, len(lmd), partial_hevp.last['iterations'], el, partial_hevp.last['solve_time']))
print('max rel eigenvalue error vs analytic: %.2e' % np.max(np.abs(lmd[:a.k] - ana) / ana))
print('convergence status:', partial_hevp.last['convergence_status'])
print('residual norms:', np.array2string(np.asarray(partial_hevp.last['residual_norms']), precision=2))
print('eigvec err (kinematic):', np.array2string(partial_hevp.last['eigenvector_errors'][0], precision=2))
print('eigvec err (residual):', np.array2string(partial_hevp.last['eigenvector_errors'][1], precision=2))
print('rel eigenvalue errors:', np.array2string(np.abs(lmd[:a.k] - ana) / ana, precision=2))
