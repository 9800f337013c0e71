"""bench.py's multi-rank control flow (rank bookkeeping, weak and strong modes, the sharded solve, the
agreed failure flag, ONE JSON line from rank 0) rehearsed with two ranks on the CPU: RLH_BENCH_REHEARSAL=1
puts the C-ABI stand-in of the test tier and the gloo backend under it.  The driver's N = 2, 4, 8 runs are
the only place the real thing executes, so a crash there would lose the scaling record."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    """Exactly `python bench.py --gpus 2 ...` with no rank environment set: the script starts its own two ranks under
    torch.distributed.run (as a child process) and relays rank 0's single JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(RLH_BENCH_REHEARSAL='1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--side', '12', '--solve-side', '12']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:].decode()
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1                                 # only rank 0's line reaches stdout
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in d
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 2
    assert d['config']['n'] == 12 ** 3                     # strong (the default): the same rows split over the ranks
    assert d['also']['scaling'] == 'weak' and d['also']['n'] == 12 * 12 * 24     # weak: the grid grows along z
    assert d['collectives']['round_trips_per_step'] == {'headline': 13, 'fused': 5} and d['collectives']['round_trip_us'] > 0
    c4 = d['config4']                                      # the row-sharded PCA leg (toy size under the rehearsal)
    assert 'error' not in c4 and c4['max_sigma_error_over_sigma_max'] < 5e-3 and c4['iterations'] > 0
    assert d['solve']['status'] == 0 and d['solve']['max_rel_eigenvalue_error'] < 1e-9
    c5 = d['config5']                                      # the row-sharded inexact shift-invert leg (toy size)
    assert 'error' not in c5 and c5['status'] == 0 and c5['max_rel_eigenvalue_error'] < 1e-10
    assert c5['negative_eigenvalues_counted'] == 6 and c5['inner_iterations'] > c5['inner_solves'] > 2
    assert 'cpu_baseline' not in d and 'configs' not in d  # N = 1 only
