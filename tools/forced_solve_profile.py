"""cProfile of the 215^3 ten-eigenpair solve on the sharded path forced at one rank (where does the host time go?)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29572')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import numpy as np, torch, torch.distributed as dist
from raleigh_amd import _lib
_lib.lib()
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from raleigh_amd.algebra.hip.dist import Comm
import bench
side = int(sys.argv[1]) if len(sys.argv) > 1 else 215
forced = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
comm = Comm(force_collectives=True) if forced else None
if comm is not None:
    comm.forced_halo_rows = 2 * side * side
r = bench.solve_ten(side, comm)       # warm-up (allocations, layouts)
pr = cProfile.Profile()
pr.enable()
r = bench.solve_ten(side, comm)
pr.disable()
print({k: r[k] for k in ('seconds', 'setup_seconds', 'iterations', 'max_rel_eigenvalue_error')})
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
dist.destroy_process_group()
