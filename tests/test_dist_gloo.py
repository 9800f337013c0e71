"""Multi-process CPU tier: the row-sharded path (partial Gram + all-reduce, halo
exchange for the sparse operator) with the gloo backend, world sizes 2, 3 and 4."""

import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('world,host_reduce', [(2, '1'), (3, '1'), (4, '1'), (2, '0')])
def test_sharded_vectors_and_operator(world, host_reduce):
    """host_reduce '1': the small reductions through the node's shared-memory segment (the default of a multi-rank run on one
    node); '0': everything through the backend's all-reduce."""
    env = dict(os.environ)
    env.pop('RANK', None)
    env['RLH_HOST_REDUCE'] = host_reduce
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(29500 + world + (os.getpid() % 200)),
           os.path.join(HERE, '_dist_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_OK world=%d' % world in r.stdout
