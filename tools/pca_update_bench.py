"""Incremental PCA / PCA update on one MI355X: wall time of pca(A, batch_size=...) and pca(A1, have=...) beside
the one-shot pca(A) on the same synthetic fp32 data (rows x cols, given rank), with the errors of each."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=12000)
ap.add_argument('--N', type=int, default=4000)
ap.add_argument('--rank', type=int, default=2000)
ap.add_argument('--batch', type=int, default=4000)
ap.add_argument('--tol', type=float, default=0.05)
ap.add_argument('--device-eigh', action='store_true', help='import PyTorch first: the k x k eigenproblems then go to the GPU')
a = ap.parse_args()
if a.device_eigh:
    import torch  # noqa: F401
from raleigh_amd.interfaces import pca, pca_error
from raleigh_amd import _lib
from oracle.pca_data import generate          # the reference's test-data generator (examples/pca/generate_matrix.py), restated
np.random.seed(1)
M, N, r = a.M, a.N, a.rank
A, sigma, u, v = generate(M, N, r, pca=True)
pca(A[:2000], npc=10)       # warm the library up


def timed(name, f):
    _lib.synchronize(); t0 = time.time()
    mean, trans, comps = f()
    _lib.synchronize(); el = time.time() - t0
    em, ef = pca_error(A, mean, trans, comps)
    print('%-28s %6.2f s  %4d components  iterations %3d  operator %5.2f s  errors %.1e %.1e'
          % (name, el, comps.shape[0], pca.last['iterations'], pca.last['operator_time'], em, ef))
    return mean, trans, comps


timed('one shot', lambda: pca(A, tol=a.tol))
timed('incremental, batch %d' % a.batch, lambda: pca(A, batch_size=a.batch, tol=a.tol))
cut = M - a.batch
m0 = pca(np.ascontiguousarray(A[:cut]), tol=a.tol)
timed('update with last %d rows' % a.batch, lambda: pca(np.ascontiguousarray(A[cut:]), have=m0))
if os.environ.get('PCA_UPDATE_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    pca(np.ascontiguousarray(A[cut:]), have=m0)
    _lib.synchronize()
    pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(25)
