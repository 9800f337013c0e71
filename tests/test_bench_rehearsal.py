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
    env = dict(os.environ, RLH_BENCH_REHEARSAL='1', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29583',
               OMP_NUM_THREADS='2')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--side', '12', '--solve-side', '12']
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, cwd=ROOT) for r in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:].decode() + outs[1][1][-2000:].decode()
    assert outs[1][0].strip() == b''                       # only rank 0 prints
    lines = [l for l in outs[0][0].decode().splitlines() if l.strip()]
    assert len(lines) == 1
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
    assert 'cpu_baseline' not in d and 'configs' not in d  # N = 1 only
