"""Host set-up times next to the solves they serve: the SpMM layout build of lap3d side^3 (float64 and
float32 operators, from the upper triangle as the solver hands it over) and the ILUT factorisation + triangular-solve set-up
of the config-3 surrogate.  RLH_SPMM_VERBOSE=1 adds the phases of the layout build on stderr.

    python tools/setup_bench.py [side] [threads ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
side = int(sys.argv[1]) if len(sys.argv) > 1 else 215
threads = [int(t) for t in sys.argv[2:]] or [0]
from raleigh_amd import _lib
from raleigh_amd.algebra.hip import CsrOperator, synchronize
from raleigh_amd.algebra.hip.precond import IncompleteLU
from raleigh_amd.synthetic import lap3d_rows, fe_surrogate
_lib.lib()
t0 = time.perf_counter()
A = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, side ** 3)
print('lap3d %d^3 as SciPy CSR: %.2f s (nnz %d)' % (side, time.perf_counter() - t0, A.nnz), flush=True)
A32 = A.astype(np.float32)
F = fe_surrogate()
for th in threads:
    if th:
        os.environ['RLH_HOST_THREADS'] = str(th)
    for name, mat in (('float64', A), ('float32', A32)):
        for upper in (True, False):
            t0 = time.perf_counter()
            op = CsrOperator(mat, upper=upper)
            synchronize()
            print('threads %s: %s operator %s: %.3f s  layout %s stacks %d' % (th or 'default', name, 'from its upper triangle' if upper else 'from both triangles',
                  time.perf_counter() - t0, op.layout()[0], op.stacks()[0]), flush=True)
            del op
    t0 = time.perf_counter()
    T = IncompleteLU(F)
    t1 = time.perf_counter()
    T.factorize()
    synchronize()
    print('threads %s: config-3 surrogate: IncompleteLU() %.3f s, factorize() %.3f s (levels %s, fill %.2f)' %
          (th or 'default', t1 - t0, time.perf_counter() - t1, T.levels, T.fill), flush=True)
