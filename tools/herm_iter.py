"""A few block-JCG iterations on the config-5 operator at full size (Hermitian lap3d + i skew, n = 126^3, complex128, block of 64),
no preconditioner and no factorisation: what the driver's iteration costs on complex blocks of 2 GB.  For rocprofv3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raleigh_amd.interfaces import partial_hevp
from raleigh_amd.core.solver import Options
from raleigh_amd.synthetic import hermitian_lap3d_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 126
its = int(sys.argv[2]) if len(sys.argv) > 2 else 12
A = hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, N ** 3)
np.random.seed(1)
opt = Options(); opt.max_iter = its; opt.block_size = 64
t0 = time.time()
lmd, x, status = partial_hevp(A, T=True, which=20, tol=1e-6, verb=-1, opt=opt)   # (T=True: no factorisation, no preconditioner)
print('n=%d: %d iterations in %.2f s (solve %.2f s), status %d' % (N ** 3, partial_hevp.last['iterations'], time.time() - t0, partial_hevp.last['solve_time'], status))
