"""Fixed cost of one halo exchange of the row-sharded sparse operator, on ONE GPU with the collectives forced (the rank
exchanges two grid planes with itself): lap3d side^3, m vectors, float64 product and the bfloat16 Chebyshev step.
Stages timed with a synchronisation after each (what every stage costs alone) and the whole application unsynchronised.

    python tools/halo_bench.py [side] [m]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
side = int(sys.argv[1]) if len(sys.argv) > 1 else 215
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29571')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
from raleigh_amd import _lib
L = _lib.lib()
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from raleigh_amd.algebra.hip.dist import Comm, ShardedVectors, ShardedSparseMatrix, partition
from raleigh_amd.algebra.hip import Vectors, CsrOperator
from raleigh_amd.synthetic import lap3d_rows
comm = Comm(force_collectives=True)
comm.forced_halo_rows = 2 * side * side
n = side ** 3
off = partition(n, 1)
rows = lap3d_rows(side, side, side, 1.0, 1.01, 1.02, 0, n)
op = ShardedSparseMatrix.from_local_rows(rows, 0, n, comm, off)
plain = CsrOperator(rows)
X = ShardedVectors(n, m, np.float64, comm=comm, offsets=off); Y = ShardedVectors(n, m, np.float64, comm=comm, offsets=off)
X.fill_random()
def sync():
    _lib.check(L.rlh_sync()); torch.cuda.synchronize()
def timeit(f, reps=20):
    f(); sync()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    sync()
    return (time.perf_counter() - t) / reps * 1e3
print('lap3d %d^3, %d vectors, halo rows %d (%.1f MB per exchange)' % (side, m, op.halo_rows(), op.halo_rows() * m * 8 / 1e6))
print('layouts: plain', plain.layout(), plain.stacks(), ' sharded', op._op.layout(), op._op.stacks())
print('plain product          %.3f ms' % timeit(lambda: plain.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld())))
print('sharded product        %.3f ms' % timeit(lambda: op.apply(X, Y)))
hp0, ldh0 = op._exchange_halo(X)
print('sharded operator, all rows in one call (part 0) %.3f ms' % timeit(lambda: op._op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), hp0, ldh0, part=0)))
# stages, each followed by a synchronisation
acc = {}
def stage(name, f):
    sync(); t = time.perf_counter(); r = f(); sync(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
reps = 20
for _ in range(reps):
    pending = stage('pack + post', lambda: op._start_exchange(X))
    hp, ldh = op._halo_slot(X)
    stage('interior rows', lambda: op._op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), hp, ldh, part=1))
    halo_ptr, ldh = stage('wait + unpack', lambda: op._finish_exchange(pending))
    stage('boundary rows', lambda: op._op.apply_ptr(m, X.data_ptr(), X.ld(), Y.data_ptr(), Y.ld(), halo_ptr, ldh, part=2))
for k, v in acc.items():
    print('  %-20s %.3f ms' % (k, v / reps * 1e3))
# host-side time of posting alone (no sync)
t = time.perf_counter()
for _ in range(reps):
    p = op._start_exchange(X)
    host = time.perf_counter()
    op._finish_exchange(p)
sync()
print('exchange alone (post + wait + unpack, unsynchronised loop) %.3f ms' % ((time.perf_counter() - t) / reps * 1e3))
# ---- the bfloat16 Chebyshev step (float32 operator, 16 vectors): plain against sharded
from raleigh_amd.algebra.hip.sparse import Bf16Block
rows32 = rows.astype(np.float32)
op32 = ShardedSparseMatrix.from_local_rows(rows32, 0, n, comm, off)
plain32 = CsrOperator(rows32)
mb = 16
nloc = n
yb, pb, bb = (Bf16Block(nloc, mb) for _ in range(3))
X32 = Vectors(n, mb, data_type=np.float32); X32.fill_random()
for blk in (yb, pb, bb):
    blk.pack(X32, 1.0)
print('bf16 ready: plain %s, sharded %s' % (plain32.bf16_ready(), op32.supports_bf16()))
print('bf16 Chebyshev step, plain    %.3f ms' % timeit(lambda: plain32.cheb_step_bf16(mb, yb, pb, bb, 1.0, -0.5, 0.25), 50))
print('bf16 Chebyshev step, sharded  %.3f ms' % timeit(lambda: op32.cheb_step_bf16(mb, yb, pb, bb, 1.0, -0.5, 0.25), 50))
import cProfile, pstats
pr = cProfile.Profile(); sync(); pr.enable()
for _ in range(300):
    op32.cheb_step_bf16(mb, yb, pb, bb, 1.0, -0.5, 0.25)
pr.disable(); sync()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
t = time.perf_counter()
for _ in range(300):
    op32.cheb_step_bf16(mb, yb, pb, bb, 1.0, -0.5, 0.25)
host = time.perf_counter() - t
sync()
print('300 sharded steps: host issue time %.3f ms per step, with the device %.3f ms per step' % (host / 300 * 1e3, (time.perf_counter() - t) / 300 * 1e3))
dist.destroy_process_group()
