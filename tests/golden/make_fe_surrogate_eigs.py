"""Generates tests/golden/fe_surrogate_eigs.json: the 12 smallest eigenvalues of the FE-like
shipsec5 surrogate (raleigh_amd/synthetic.py fe_surrogate, BASELINE config 3) by SciPy's
shift-invert Lanczos (eigsh, sigma = 0, SuperLU factorisation: ~10 minutes and 3.4 GB of fill in
the build container), the independent reference VERDICT r01 asks the GPU run to be checked against.
Run from the repository root:  PYTHONPATH=. python tests/golden/make_fe_surrogate_eigs.py"""
import json
import os
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from raleigh_amd.synthetic import fe_surrogate

A = fe_surrogate()
t = time.time()
lu = sla.splu(sp.csc_matrix(A), permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
print('splu %.1fs fill L nnz %d' % (time.time() - t, lu.L.nnz), flush=True)
t = time.time()
op = sla.LinearOperator(A.shape, matvec=lu.solve, dtype=np.float64)
w, v = sla.eigsh(A, k=12, sigma=0.0, OPinv=op, which='LM', tol=1e-14)
print('eigsh %.1fs' % (time.time() - t))
w = np.sort(w)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fe_surrogate_eigs.json')
json.dump({'eigenvalues': list(map(float, w))}, open(out, 'w'))
