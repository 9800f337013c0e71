"""Time of the host ILUT(1e-6, fill 1) of the config-3 surrogate by thread count: python tools/ilut_bench.py [threads ...]"""
import os, sys, time, ctypes
import numpy as np, scipy.sparse as scs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd import _lib
from raleigh_amd.synthetic import fe_surrogate
a = scs.csr_matrix(fe_surrogate()); a.sort_indices()
n = a.shape[0]
indptr = np.ascontiguousarray(a.indptr, dtype=np.int64); indices = np.ascontiguousarray(a.indices, dtype=np.int32)
values = np.ascontiguousarray(a.data, dtype=np.float64)
L = _lib.library()
for th in [int(t) for t in sys.argv[1:]] or [1, 4, 16]:
    os.environ['RLH_HOST_THREADS'] = str(th)
    best = 1e9
    for rep in range(2):
        f = ctypes.c_void_p()
        t = time.perf_counter()
        _lib.check(L.rlh_ilut_factor(ctypes.byref(f), _lib.DTYPE_CODE[np.float64], n, _lib.host_ptr(indptr), _lib.host_ptr(indices),
                                     _lib.host_ptr(values), 1e-6, int(min(n - 1, a.nnz // n))))
        best = min(best, time.perf_counter() - t)
        L.rlh_factors_destroy(f)
    print('threads %3d: %.3f s' % (th, best), flush=True)
