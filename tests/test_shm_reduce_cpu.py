"""The shared-memory sum of small host arrays over the processes of a node (rlh_shm_*: host-only entry points of the real
library): 2, 4 and 7 processes make thousands of reductions of changing sizes and types back to back; every process must
see exactly the sum in rank order (the same bits everywhere), nothing may be left behind in /dev/shm, and the argument
checks must hold."""

import ctypes
import multiprocessing as mp
import os
import uuid

import numpy as np
import pytest

from raleigh_amd import _lib


def _worker(name, rank, nranks, rounds, seed, q):
    try:
        L = _lib.library()
        h = ctypes.c_void_p()
        _lib.check(L.rlh_shm_create(ctypes.byref(h), name.encode(), rank, nranks, 1 << 16))
        rng = np.random.default_rng(seed)                       # the SAME stream on every rank: sizes and data agree
        bad = 0
        for it in range(rounds):
            count = int(rng.integers(0, 2049))
            dt = np.float64 if rng.integers(0, 2) else np.float32
            data = rng.standard_normal((nranks, count)).astype(dt)
            mine = data[rank].copy()
            _lib.check(L.rlh_shm_allreduce(h, _lib.DTYPE_CODE[dt], count, _lib.host_ptr(mine)))
            want = data[0].copy()
            for r in range(1, nranks):
                want += data[r]                                  # rank order, the element type's own arithmetic
            if not np.array_equal(mine, want):
                bad += 1
        L.rlh_shm_destroy(h)
        q.put((rank, bad))
    except Exception as e:                                       # pragma: no cover
        q.put((rank, 'error: %r' % (e,)))


@pytest.mark.parametrize('nranks,rounds', [(2, 3000), (4, 2000), (7, 800)])
def test_shared_memory_sum_over_processes(nranks, rounds):
    ctx = mp.get_context('spawn')
    name = '/rlh_test_%s' % uuid.uuid4().hex[:12]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(name, r, nranks, rounds, 1234, q)) for r in range(nranks)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    _lib.check(_lib.library().rlh_shm_unlink(name.encode()))
    assert sorted(got) == [(r, 0) for r in range(nranks)]
    assert not os.path.exists('/dev/shm' + name)


def test_shared_memory_sum_argument_checks():
    L = _lib.library()
    h = ctypes.c_void_p()
    name = '/rlh_test_%s' % uuid.uuid4().hex[:12]
    with pytest.raises(_lib.RlhError):
        _lib.check(L.rlh_shm_create(ctypes.byref(h), b'no_slash', 0, 1, 1024))
    with pytest.raises(_lib.RlhError):
        _lib.check(L.rlh_shm_create(ctypes.byref(h), name.encode(), 2, 2, 1024))
    _lib.check(L.rlh_shm_create(ctypes.byref(h), name.encode(), 0, 1, 1024))
    try:
        with pytest.raises(_lib.RlhError):                       # the name is taken
            h2 = ctypes.c_void_p()
            _lib.check(L.rlh_shm_create(ctypes.byref(h2), name.encode(), 0, 1, 1024))
        x = np.arange(8, dtype=np.float64)
        _lib.check(L.rlh_shm_allreduce(h, _lib.DTYPE_CODE[np.float64], 8, _lib.host_ptr(x)))      # one rank: unchanged
        assert np.array_equal(x, np.arange(8))
        big = np.zeros(1024, dtype=np.float64)
        with pytest.raises(_lib.RlhError):                       # 8 KB into a 1 KB slot
            _lib.check(L.rlh_shm_allreduce(h, _lib.DTYPE_CODE[np.float64], 1024, _lib.host_ptr(big)))
        with pytest.raises(_lib.RlhError):                       # complex types go as pairs of reals
            _lib.check(L.rlh_shm_allreduce(h, _lib.DTYPE_CODE[np.complex128], 4, _lib.host_ptr(big)))
    finally:
        L.rlh_shm_destroy(h)
        L.rlh_shm_unlink(name.encode())
    assert not os.path.exists('/dev/shm' + name)
