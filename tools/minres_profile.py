"""Where one block MINRES solve of BASELINE config 5 spends its time (n = N^3 complex128, block of 64): every Vectors operation,
operator and preconditioner application wrapped with a synchronisation and a wall-clock timer; what is left is host algebra.

    python tools/minres_profile.py [N] [K] [degree] [ratio] [tol] [sync 0/1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.algebra.hip import Vectors, synchronize
from raleigh_amd.algebra.hip import shift_invert as S
from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner
from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues
arg = lambda i, d, t=float: t(sys.argv[i]) if len(sys.argv) > i else d
N, K, degree, ratio, tol, do_sync, low = arg(1, 126, int), arg(2, 40, int), arg(3, 16, int), arg(4, 250.0), arg(5, 1e-10), arg(6, 1, int), arg(7, 0, int)
H = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
sigma = 0.5 * (exact[K - 1] + exact[K])
sol = S.IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, tol=tol, degree=degree, ratio=ratio)
sol.analyse(H, sigma)
sol.factorize()
n, m = N ** 3, 64
b, x = Vectors(n, m, data_type=np.complex128), Vectors(n, m, data_type=np.complex128)
np.random.seed(1)
b.fill_random()
sol.solve(b, x)           # warm-up: allocations, first launches
acc = {}
def timed(name, f):
    def g(*a, **k):
        if do_sync:
            synchronize()
        t = time.perf_counter()
        r = f(*a, **k)
        if do_sync:
            synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        acc[name + '#'] = acc.get(name + '#', 0) + 1
        return r
    return g
for name in ('dot', 'multiply', 'add', 'copy', 'zero'):
    setattr(Vectors, name, timed(name, getattr(Vectors, name)))
sol._op.apply = timed('operator', sol._op.apply)
sol._pre.apply = timed('preconditioner', sol._pre.apply)
for name in ('eigh', 'qr', 'solve_triangular'):
    setattr(S.sla, name, timed('host ' + name, getattr(S.sla, name)))
synchronize()
t0 = time.perf_counter()
sol.solve(b, x)
synchronize()
total = time.perf_counter() - t0
its = sol.last.iterations
print('n = %d, %d steps, %.3f s = %.1f ms per step (sync %d)' % (n, its, total, 1e3 * total / its, do_sync))
rest = total
for k in sorted(k for k in acc if not k.endswith('#')):
    print('  %-22s %7.1f ms per step  (%d calls)' % (k, 1e3 * acc[k] / its, acc[k + '#']))
    rest -= acc[k]
print('  %-22s %7.1f ms per step' % ('other host work', 1e3 * rest / its))
r = Vectors(n, m, data_type=np.complex128)
sol._op.apply(x, r)
r.add(b, -1.0)
print('true residual (2-norm, worst column): %.2e; estimated (preconditioner norm): %.2e' %
      (np.max(np.sqrt(np.abs(r.dots(r)) / np.abs(b.dots(b)))), np.max(sol.last.residuals)))
