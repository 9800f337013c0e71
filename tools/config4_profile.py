"""cProfile of BASELINE config 4's row shard (62500 x 40000 fp32, 1000 components) on one MI355X: where the time of
pca() goes beside the dense products."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import Vectors, Matrix
from raleigh_amd.algebra.dense_matrix import AMatrix
from raleigh_amd.interfaces import pca
L = _lib.lib()
M, Nn, r, npc = 62500, 40000, 1280, 1000
rng = np.random.default_rng(4)
U = rng.standard_normal((M, r)).astype(np.float32); U[:, 0] = 1.0
V = rng.standard_normal((Nn, r)).astype(np.float32)
U, _ = np.linalg.qr(U); V, _ = np.linalg.qr(V)
s = np.sort(rng.random(Nn).astype(np.float32)) ** (-0.75); s = (s / s[0])[:r]
rows = Vectors(Nn, M, data_type=np.float32)
Matrix(np.ascontiguousarray(V)).apply(Vectors(np.ascontiguousarray(U * s)), rows)
del U, V
A4 = AMatrix(rows)
np.random.seed(1)
pca(A4, npc=50)
np.random.seed(1)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
mean, trans, comps = pca(A4, npc=npc)
el = time.perf_counter() - t0
pr.disable()
print('pca %.3f s, iterations %d, operator %.3f s' % (el, pca.last['iterations'], pca.last['operator_time']))
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
