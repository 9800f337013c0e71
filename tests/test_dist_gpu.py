"""The row-sharded (torch.distributed / RCCL) code path on ONE GPU with the collectives forced: the
all-reduce of every reduction, the halo exchange of the sparse operator (pack -> send to self -> receive
-> halo block -> boundary rows) and their ordering with the kernels' stream run on hardware, which a
single rank otherwise skips ("no RCCL collective has ever executed")."""

import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ops
from oracle.sparse import lap3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def comm():
    import torch
    import torch.distributed as dist
    from raleigh_amd import _lib
    _lib.set_library(None)
    L = _lib.lib()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29561')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    from raleigh_amd.algebra.hip.dist import Comm
    c = Comm(force_collectives=True)
    yield c
    _lib.check(L.rlh_sync())
    torch.cuda.synchronize()
    dist.destroy_process_group()
    _lib.check(L.rlh_set_stream(None))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize('dt', [np.float64, np.complex128, np.float32])
def test_reductions_through_rccl(comm, dt):
    from raleigh_amd.algebra.hip.dist import ShardedVectors
    rng = np.random.default_rng(5)
    n, m, k = 50021, 12, 7
    x = rng.standard_normal((m, n)).astype(dt)
    y = rng.standard_normal((k, n)).astype(dt)
    if dt == np.complex128:
        x = x + 1j * rng.standard_normal((m, n))
        y = y - 1j * rng.standard_normal((k, n))
    tol = 2e-5 if dt == np.float32 else 1e-13
    X, Y = ShardedVectors(x, comm=comm), ShardedVectors(y, comm=comm)
    assert rel(X.dot(Y), ops.gram(x, y)) < tol
    assert rel(X.dots(X), ops.dots(x, x)) < tol
    rb = X.reduction_batch()
    rb.gram([X], [Y, X])
    rb.dots(Y, Y)
    g, d = rb.run()
    assert rel(g[:k], ops.gram(x, y)) < tol and rel(g[k:], ops.gram(x, x)) < tol and rel(d, ops.dots(y, y)) < tol
    assert np.array_equal(X.data(), x)                    # all-gather of the shards


@pytest.mark.parametrize('dt', [np.float64, np.complex128, np.float32])
def test_reductions_through_shared_memory(comm, monkeypatch, dt):
    """The same reductions with the small results summed in the node's shared-memory segment (what a multi-rank run on one
    node does by default; asked for at one rank by RLH_HOST_REDUCE=2): the partial comes off the device through the
    library's fetch, the sum happens on the host -- no collective is launched for it."""
    from raleigh_amd.algebra.hip.dist import Comm, ShardedVectors
    monkeypatch.setenv('RLH_HOST_REDUCE', '2')
    c2 = Comm(force_collectives=True)
    assert c2._shm is not None
    calls = []
    real = c2.dist.all_reduce
    monkeypatch.setattr(c2.dist, 'all_reduce', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    rng = np.random.default_rng(6)
    n, m, k = 50021, 12, 7
    x = rng.standard_normal((m, n)).astype(dt)
    y = rng.standard_normal((k, n)).astype(dt)
    if dt == np.complex128:
        x = x + 1j * rng.standard_normal((m, n))
        y = y - 1j * rng.standard_normal((k, n))
    tol = 2e-5 if dt == np.float32 else 1e-13
    X, Y = ShardedVectors(x, comm=c2), ShardedVectors(y, comm=c2)
    assert rel(X.dot(Y), ops.gram(x, y)) < tol
    assert rel(X.dots(X), ops.dots(x, x)) < tol
    rb = X.reduction_batch()
    rb.gram([X], [Y, X])
    rb.dots(Y, Y)
    g, d = rb.run()
    assert rel(g[:k], ops.gram(x, y)) < tol and rel(g[k:], ops.gram(x, x)) < tol and rel(d, ops.dots(y, y)) < tol
    assert not calls
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('rlh_')]


@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_halo_exchange_with_itself(comm, dt):
    """The forced one-rank run cuts the shard into two virtual ranks (RLH_FORCE_COLLECTIVES semantics of
    ShardedSparseMatrix with one rank): the grid plane on either side of the cut is fetched through the halo path --
    gather_rows, batch_isend_irecv to the own rank, the strided copy into the halo block and the interior / boundary
    split of the windowed kernel."""
    from raleigh_amd.algebra.hip.dist import ShardedVectors, ShardedSparseMatrix
    A = lap3d(40, 40, 40, 1.0, 1.01, 1.02).astype(dt)
    n = A.shape[0]
    rng = np.random.default_rng(8)
    x = rng.standard_normal((9, n)).astype(dt)
    op = ShardedSparseMatrix(A, comm)
    assert op.halo_rows() == 2 * 40 * 40                  # the forced self-exchange is really there: one plane each way
    X, Y = ShardedVectors(x, comm=comm), ShardedVectors(n, 9, dt, comm=comm)
    op.apply(X, Y)
    ref = ops.csr_sym_apply(sp.triu(A, format='csr'), x)
    assert rel(Y.data(), ref) < (3e-6 if dt == np.float32 else 1e-13)
    # the fused Chebyshev step through the same exchange
    p0, b0 = rng.standard_normal((9, n)).astype(dt), rng.standard_normal((9, n)).astype(dt)
    P, B = ShardedVectors(p0.copy(), comm=comm), ShardedVectors(b0.copy(), comm=comm)
    op.cheb_step(X, P, B, 1.25, -0.25, 0.5)
    assert rel(P.data(), 1.25 * x - 0.25 * p0 + 0.5 * (b0 - ref)) < (5e-6 if dt == np.float32 else 1e-13)


def test_sharded_driver_with_forced_collectives(comm):
    """The block-JCG driver on row-sharded vectors: every batch of reductions goes through one all-reduce."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.dist import ShardedVectors, ShardedSparseMatrix
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    from oracle.sparse import lap3d_eigenvalues
    A = lap3d(30, 30, 30, 1.0, 1.01, 1.02)
    n = A.shape[0]
    op = ShardedSparseMatrix(A, comm)
    T = ChebyshevPreconditioner(op, gershgorin_upper_bound(A), ratio=300.0, degree=10)
    np.random.seed(1)
    lmd, x, status = partial_hevp(None, T=T, which=6, tol=1e-6, verb=-1, operator=op,
                                  vectors=lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm))
    assert status == 0
    assert np.allclose(lmd[:6], lap3d_eigenvalues(30, 30, 30, 1.0, 1.01, 1.02, 6), rtol=1e-10)


def test_sharded_pca_matches_single_gpu(comm):
    """BASELINE config 4's layout on hardware with the collectives forced: the row-sharded dense operator (rows built
    in HBM, rank-one epilogue on the local rows, the transposed product reduced in column chunks whose all-reduce
    overlaps the next chunk's GEMM) under pca() against the plain single-GPU path on the same data, and the two
    products against the oracle."""
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.algebra.hip.dist import ShardedDenseMatrix, ShardedAMatrix
    from raleigh_amd.interfaces import pca
    from oracle.pca_data import generate
    np.random.seed(1)
    A, sigma, u, v = generate(3000, 4200, 300, dtype=np.float32, pca=True)
    rows = Vectors(np.ascontiguousarray(A))                       # this rank's rows, one vector per row, in HBM
    Ad = ShardedDenseMatrix(rows, comm)
    assert Ad.shape() == (3000, 4200)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((8, 4200)).astype(np.float32)
    X, Y = Vectors(x), Ad.new_vectors(3000, 8)
    Ad.apply(X, Y)
    ref = ops.dense_apply(A, x)
    assert rel(Y.data(), ref) < 1e-5
    W = Ad.new_vectors(4200, 8)
    trips = Ad.round_trips
    Ad.apply(Y, W, transp=True)
    assert Ad.round_trips - trips == 4                            # four overlapped chunk reductions (RCCL, to itself)
    assert rel(W.data(), ops.dense_apply(A, ref, True)) < 1e-4
    np.random.seed(1)
    mean, trans, comps = pca(ShardedAMatrix(rows, comm), npc=40)
    np.random.seed(1)
    mean1, trans1, comps1 = pca(A, npc=40)
    sv, sv1 = np.linalg.norm(trans, axis=0), np.linalg.norm(trans1, axis=0)
    exact = np.linalg.svd((A - A.mean(axis=0, keepdims=True)).astype(np.float64), compute_uv=False)[:40]
    assert np.max(np.abs(sv - exact)) <= 1e-3 * exact[0] and np.max(np.abs(sv1 - exact)) <= 1e-3 * exact[0]
    assert np.max(np.abs(sv - sv1)) <= 1e-3 * exact[0]
    assert rel(mean, mean1) < 1e-5
    # the same subspace: the leading 30 components of one lie in the span of the other's 40
    c = np.linalg.svd(comps[:30] @ comps1.T, compute_uv=False)
    assert c.min() > 1 - 1e-3


def test_sharded_shift_invert_stays_on_the_device(comm):
    """BASELINE config 5's layout with the collectives forced: shift-invert on row-sharded complex blocks -- the block is
    gathered on the device (all_gather on the kernels' stream), solved by the persistent triangular chain, and this
    rank's rows are taken back: eigenvalues nearest an interior shift against the dense spectrum."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.algebra.hip.dist import ShardedVectors
    N = 12
    A = sp.csr_matrix(lap3d(N, N, N, 1.0, 1.01, 1.02).astype(np.complex128))
    n = A.shape[0]
    S = sp.diags([np.full(n - 1, 0.3)], [1])
    A = sp.csr_matrix(A + 1j * S - 1j * S.T)
    exact = np.linalg.eigvalsh(A.toarray())
    sigma = 0.5 * (exact[40] + exact[41])
    np.random.seed(1)
    lmd, x, status = partial_hevp(A, sigma=sigma, which=(4, 4), tol=1e-8, verb=-1,
                                  vectors=lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm))
    assert status == 0
    want = np.sort(exact[37:45])
    assert np.max(np.abs(np.sort(lmd) - want)) < 1e-10 * np.abs(exact).max()


def test_config5_sharded_inexact_shift_invert_at_full_size(comm):
    """BASELINE config 5's own layout at its own size with the collectives forced: row-sharded complex128 blocks of 64 at
    n = 126^3, the sharded operator (halo exchange with itself), every reduction of the eigensolver AND of the block MINRES
    inside it through RCCL, nothing gathered and nothing factorised; 20 eigenvalues nearest the shift to 1e-10."""
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from raleigh_amd.algebra.hip.dist import ShardedVectors, ShardedSparseMatrix, partition
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    from raleigh_amd.synthetic import hermitian_lap3d_rows, hermitian_lap3d_eigenvalues, lap3d_coefficients
    N, below = 126, 40
    n = N ** 3
    exact = hermitian_lap3d_eigenvalues(N, N, N, 1.0, 1.01, 1.02)
    sigma = 0.5 * (exact[below - 1] + exact[below])
    off = partition(n, comm.size)
    comm.forced_halo_rows = 2 * N * N
    op = ShardedSparseMatrix.from_local_rows(hermitian_lap3d_rows(N, N, N, 1.0, 1.01, 1.02, 0, n), 0, n, comm, off)
    hi = 4.0 * sum(lap3d_coefficients(N, N, N, 1.0, 1.01, 1.02)) + 0.6
    sol = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=16, ratio=250.0, hi=hi)
    opt = Options()
    opt.block_size = 64
    np.random.seed(1)
    lmd, x, status = partial_hevp(None, sigma=sigma, which=20, tol=1e-6, verb=-1, opt=opt, solver=sol, operator=op,
                                  vectors=lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm, offsets=off))
    assert status == 0 and len(lmd) >= 20
    assert sol.inertia(vectors=lambda nn, nv, data_type: ShardedVectors(nn, nv, data_type, comm=comm, offsets=off))[0] == below
    for e in exact[np.argsort(np.abs(exact - sigma))[:20]]:
        assert np.min(np.abs(lmd - e)) < 1e-10 * abs(e)
