"""Small problems end to end (BASELINE config 1 and its ILU variant): seconds on the GPU path against the CPU oracle
driving the same solver.  usage: tools/small_solve.py [--side 30]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--side', type=int, default=30)
ap.add_argument('--profile', action='store_true')
a = ap.parse_args()
from raleigh_amd.synthetic import lap3d_rows
from raleigh_amd.interfaces import partial_hevp
N = a.side
A = lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
def run(tag, **kw):
    np.random.seed(1)
    t0 = time.perf_counter()
    lmd, x, status = partial_hevp(A, verb=-1, **kw)
    dt = time.perf_counter() - t0
    print('%-34s %.3f s  status %d  %d iterations  lmd[0] = %.8f' % (tag, dt, status, partial_hevp.last['iterations'], lmd[0]))
    return dt
run('warm-up (which=6, sigma=0)', which=6, sigma=0.0, tol=1e-6)
if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
run('which=6, sigma=0 (shift-invert)', which=6, sigma=0.0, tol=1e-6)
if a.profile:
    pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(25)
from raleigh_amd.algebra.hip.precond import IncompleteLU
T = IncompleteLU(A); T.factorize()
run('which=10, ILU preconditioner', which=10, T=T, tol=1e-6)
run('which=10, no preconditioner', which=10, tol=1e-6)
