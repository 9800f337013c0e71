"""Worker of tests/test_dist_gloo.py: launched by torch.distributed.run with the
gloo backend (CPU); installs tests/fake_lib.py as the C-ABI library and checks
the row-sharded Vectors / sparse operator against the oracle on global data."""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch.distributed as dist
    dist.init_process_group('gloo')
    import fake_lib
    fake_lib.install()
    from raleigh_amd.algebra.hip.dist import Comm, ShardedVectors, ShardedSparseMatrix, partition
    from oracle import ops
    from oracle.sparse import lap3d
    comm = Comm()
    rank, size = comm.rank, comm.size
    assert size == int(os.environ['WORLD_SIZE'])
    off = partition(1000, size)
    assert off[0] == 0 and off[-1] == 1000 and np.all(np.diff(off) >= 0)

    # small reductions go through the shared-memory segment of the node (all ranks of this test share one): the same
    # numbers as the backend's all-reduce, the same BITS on every rank, and a reduction too large for a slot still works
    import torch
    assert (comm._shm is not None) == (os.environ.get('RLH_HOST_REDUCE', '1') != '0')
    rng = np.random.default_rng(100 + rank)
    for dtp, count in ((np.float64, 1), (np.float64, 1024), (np.complex128, 64 * 64), (np.float32, 300), (np.float64, 70000)):
        part = rng.standard_normal(count).astype(dtp)
        if np.dtype(dtp).kind == 'c':
            part = part + 1j * rng.standard_normal(count)
        buf = comm.reduction_buffer(part.nbytes)
        buf[:part.nbytes] = torch.from_numpy(part.view(np.uint8).copy())
        got = comm.allreduce_from_device(buf, dtp, count)
        ref = torch.from_numpy(part.view(np.float64 if np.dtype(dtp).itemsize % 8 == 0 else np.float32).copy())
        dist.all_reduce(ref)
        want = ref.numpy().view(dtp)
        assert np.allclose(got, want, rtol=1e-5 if dtp == np.float32 else 1e-13, atol=0)
        everyone = [None] * size
        dist.all_gather_object(everyone, got.tobytes())
        assert all(b == everyone[0] for b in everyone)

    for key, dt in (('d', np.float64), ('z', np.complex128), ('s', np.float32)):
        rng = np.random.default_rng(3)         # same global data on every rank
        n, m, k = 1003, 6, 4
        x = rng.standard_normal((m, n)).astype(dt)
        y = rng.standard_normal((k, n)).astype(dt)
        if key == 'z':
            x = x + 1j * rng.standard_normal((m, n))
            y = y - 2j * rng.standard_normal((k, n))
        tol = 1e-5 if key == 's' else 1e-13
        X, Y = ShardedVectors(x, comm=comm), ShardedVectors(y, comm=comm)
        assert X.dimension() == n and X.local_dimension() == off_len(n, size, rank)
        g = X.dot(Y)
        assert g.shape == (k, m)
        assert np.linalg.norm(g - ops.gram(x, y)) < tol * np.linalg.norm(g)
        X.select(3, 2)
        d = X.dots(X)
        assert np.linalg.norm(d - ops.dots(x[2:5], x[2:5])) < tol * np.linalg.norm(d)
        X.select(m)
        assert np.array_equal(X.data(), x)
        # transposed dots: per-row sums over the vectors, assembled on every rank
        X.select(4, 1)
        Y.select(4)
        dtr = X.dots(Y, transp=True)
        assert dtr.shape == (n,)
        assert np.linalg.norm(dtr - np.sum(np.conj(y[:4]) * x[1:5], axis=0)) < 10 * tol * np.linalg.norm(dtr)
        X.select(m)
        Y.select(k)
        # a batch of stacked Grams and dots: one all-reduce, one fetch
        rb = X.reduction_batch()
        rb.gram([X], [Y, X])
        rb.dots(Y, Y)
        g2, d2 = rb.run()
        assert np.linalg.norm(g2[:k] - ops.gram(x, y)) < tol * np.linalg.norm(g2[:k])
        assert np.linalg.norm(g2[k:] - ops.gram(x, x)) < tol * np.linalg.norm(g2[k:])
        assert np.linalg.norm(d2 - ops.dots(y, y)) < tol * np.linalg.norm(d2)
        q = rng.standard_normal((m, k)).astype(dt)
        W = Y.new_vectors(k)
        X.multiply(q, W)
        W.add(Y, -0.5)
        ref = ops.axpy(ops.multiply(x, q), y, -0.5)
        assert np.linalg.norm(W.data() - ref) < 10 * tol * np.linalg.norm(ref)
        # the random start block does not depend on the number of ranks
        np.random.seed(11)
        R = X.new_vectors(3)
        R.fill_random()
        np.random.seed(11)
        expect = (2 * np.random.rand(3, n) - 1).astype(dt)
        assert np.linalg.norm(R.data() - expect) < 1e-6 * np.linalg.norm(expect)
        # ... nor does the large-block form (library generator keyed by the GLOBAL row)
        old_thr = ShardedVectors.DEVICE_RANDOM_THRESHOLD
        ShardedVectors.DEVICE_RANDOM_THRESHOLD = 1000
        np.random.seed(11)
        R.fill_random()
        ShardedVectors.DEVICE_RANDOM_THRESHOLD = old_thr
        np.random.seed(11)
        seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
        assert np.array_equal(R.data(), ops.uniform_block(seed, n, 3, dt))
        c = X.clone()
        assert isinstance(c, ShardedVectors) and np.array_equal(c.data(), x)

    # row-sharded sparse operator with halo exchange: stencil, and an unstructured pattern
    A = lap3d(7, 6, 11, 1.0, 1.01, 1.02)
    R = sp.random(A.shape[0], A.shape[0], density=0.02, random_state=5, format='csr')
    for mat in (A, sp.csr_matrix(A + R + R.T)):
        n = mat.shape[0]
        rng = np.random.default_rng(9)
        x = rng.standard_normal((5, n))
        op = ShardedSparseMatrix(mat, comm)
        X = ShardedVectors(x, comm=comm, offsets=op._offsets)
        Y = X.new_vectors(5)
        op.apply(X, Y)
        ref = ops.csr_sym_apply(sp.triu(mat, format='csr'), x)
        assert np.linalg.norm(Y.data() - ref) < 1e-13 * np.linalg.norm(ref)
        if size > 1:
            assert op.halo_rows() > 0
    # the block-JCG driver on row-sharded blocks: same eigenvalues on any number of ranks
    from raleigh_amd.interfaces import partial_hevp
    from raleigh_amd.core.solver import Options
    from oracle.sparse import lap3d_eigenvalues
    A = lap3d(10, 9, 8, 1.0, 1.01, 1.02)
    n = A.shape[0]
    op = ShardedSparseMatrix(A, comm)
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 800
    mk = lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm, offsets=op._offsets)
    lmd, x, status = partial_hevp(None, T=True, which=4, tol=1e-8, verb=-1, opt=opt, vectors=mk, operator=op)
    assert status == 0, status
    ana = lap3d_eigenvalues(10, 9, 8, 1.0, 1.01, 1.02, 4)
    assert np.max(np.abs(lmd - ana) / ana) < 1e-10
    assert x.shape == (n, 4)
    r = A @ x - x * lmd
    assert np.max(np.linalg.norm(r, axis=0)) < 1e-5
    # the same with the device polynomial preconditioner (fused Chebyshev step + halo exchange)
    from raleigh_amd.algebra.hip.precond import ChebyshevPreconditioner, gershgorin_upper_bound
    np.random.seed(1)
    opt = Options()
    opt.max_iter = 300
    T = ChebyshevPreconditioner(op, gershgorin_upper_bound(A), ratio=100, degree=6)
    lmd2, x2, status = partial_hevp(None, T=T, which=4, tol=1e-8, verb=-1, opt=opt, vectors=mk, operator=op)
    assert status == 0, status
    assert np.max(np.abs(lmd2[:4] - ana) / ana) < 1e-10
    assert fake_lib_calls().get('spmm_cheb', 0) > 10
    # ... and with the polynomial evaluated in float32 on a float32 copy of the sharded operator
    np.random.seed(1)
    op32 = ShardedSparseMatrix(A.astype(np.float32), comm)
    T = ChebyshevPreconditioner(None, gershgorin_upper_bound(A), ratio=100, degree=6, low_precision_op=op32)
    lmd3, x3, status = partial_hevp(None, T=T, which=4, tol=1e-8, verb=-1, opt=opt, vectors=mk, operator=op)
    assert status == 0, status
    assert np.max(np.abs(lmd3[:4] - ana) / ana) < 1e-10
    assert fake_lib_calls().get('convert', 0) > 10
    # ... and with bfloat16 work blocks (the halo rows travel as 2-byte elements)
    np.random.seed(1)
    T = ChebyshevPreconditioner(None, gershgorin_upper_bound(A), ratio=100, degree=6, low_precision_op=op32,
                                storage='bf16')
    lmd4, x4, status = partial_hevp(None, T=T, which=4, tol=1e-8, verb=-1, opt=opt, vectors=mk, operator=op)
    assert status == 0, status
    assert np.max(np.abs(lmd4[:4] - ana) / ana) < 1e-10
    assert fake_lib_calls().get('spmm_cheb_bf16', 0) > 10
    # ONE rank's shard cannot take the 2-byte staging: all ranks learn it from supports_bf16() -- an all-reduce
    # MIN of the library's own layout test -- BEFORE any exchange is posted, and all of them run the float32 work blocks;
    # decided per rank by a failed launch, the refusing rank would restart with 4-byte halo messages against its peers' 2-byte ones
    from raleigh_amd import _lib
    op32b = ShardedSparseMatrix(A.astype(np.float32), comm)
    _lib.library().bf16_refused = (rank == size - 1)
    assert op32b.supports_bf16() is False
    _lib.library().bf16_refused = False
    before = fake_lib_calls().get('spmm_cheb_bf16', 0)
    np.random.seed(1)
    T = ChebyshevPreconditioner(None, gershgorin_upper_bound(A), ratio=100, degree=6, low_precision_op=op32b, storage='bf16')
    lmd4b, x4b, status = partial_hevp(None, T=T, which=4, tol=1e-8, verb=-1, opt=opt, vectors=mk, operator=op)
    assert status == 0 and np.max(np.abs(lmd4b[:4] - ana) / ana) < 1e-10
    assert fake_lib_calls().get('spmm_cheb_bf16', 0) == before                  # nobody took the bfloat16 step

    # shift-invert on row-sharded vectors (BASELINE config 5's layout): the factors are replicated, the block is
    # gathered on every rank's device (all_gather), solved by the triangular chain there, and no host solve runs
    N5 = 9
    A5 = sp.csr_matrix(lap3d(N5, N5, N5, 1.0, 1.01, 1.02).astype(np.complex128))
    n5 = A5.shape[0]
    S5 = sp.diags([np.full(n5 - 1, 0.3)], [1])
    A5 = sp.csr_matrix(A5 + 1j * S5 - 1j * S5.T)
    exact5 = np.linalg.eigvalsh(A5.toarray())
    sigma5 = 0.5 * (exact5[20] + exact5[21])
    calls_before = fake_lib_calls().get('rlh_sptrsv_solve_chain', 0)
    np.random.seed(1)
    lmd5, x5, status = partial_hevp(A5, sigma=sigma5, which=(3, 3), tol=1e-8, verb=-1,
                                    vectors=lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm))
    assert status == 0, status
    want5 = np.sort(np.concatenate((exact5[18:21], exact5[21:24])))
    assert np.max(np.abs(np.sort(lmd5) - want5)) < 1e-9 * np.abs(exact5).max()
    assert fake_lib_calls().get('rlh_sptrsv_solve_chain', 0) > calls_before + 3      # the device chain did the solves
    r5 = A5 @ x5 - x5 * lmd5
    assert np.max(np.linalg.norm(r5, axis=0)) < 1e-6 * np.abs(exact5).max()

    # the same problem by INEXACT shift-invert (block MINRES on the row-sharded operator: every reduction one all-reduce of
    # an m x m matrix, the operator exchanging its halo, nothing replicated, nothing gathered): BASELINE config 5's path
    from raleigh_amd.algebra.hip.shift_invert import IterativeSymmetricSolver
    op5 = ShardedSparseMatrix(A5, comm)
    mk5 = lambda nn, data_type: ShardedVectors(nn, 0, data_type, comm=comm)
    sol5 = IterativeSymmetricSolver(dtype=np.complex128, pos_def=True, degree=6, ratio=20.0)       # (hi: Lanczos bound on the sharded operator)
    calls_before = fake_lib_calls().get('rlh_sptrsv_solve_chain', 0)
    np.random.seed(1)
    lmd6, x6, status = partial_hevp(None, sigma=sigma5, which=(3, 3), tol=1e-7, verb=-1, vectors=mk5, operator=op5, solver=sol5)
    assert status == 0, status
    assert np.max(np.abs(np.sort(lmd6) - want5)) < 1e-10 * np.abs(exact5).max()
    assert fake_lib_calls().get('rlh_sptrsv_solve_chain', 0) == calls_before                  # no factors anywhere
    assert sol5.inertia() == (21, n5 - 21) and sol5.solves > 2
    assert exact5[-1] < sol5.hi < 1.3 * exact5[-1]
    r6 = A5 @ x6 - x6 * lmd6
    assert np.max(np.linalg.norm(r6, axis=0)) < 1e-6 * np.abs(exact5).max()

    # row-sharded dense operator and PCA (BASELINE config 4 layout): same answer as one rank
    from raleigh_amd.algebra.hip.dist import ShardedDenseMatrix, ShardedAMatrix
    from raleigh_amd.algebra.hip import Vectors
    from raleigh_amd.interfaces import pca, pca_error
    from oracle.pca_data import generate
    rng = np.random.default_rng(21)
    a = rng.standard_normal((203, 57)).astype(np.float32)
    Ad = ShardedDenseMatrix.from_global(a, comm)
    x = rng.standard_normal((4, 57)).astype(np.float32)
    X = Vectors(x)
    Yv = Ad.new_vectors(203, 4)
    Ad.apply(X, Yv)
    ref = ops.dense_apply(a, x)
    assert np.linalg.norm(Yv.data() - ref) < 1e-5 * np.linalg.norm(ref)
    W = Ad.new_vectors(57, 4)
    Ad.apply(Yv, W, transp=True)
    ref2 = ops.dense_apply(a, ref, True)
    assert np.linalg.norm(W.data() - ref2) < 1e-4 * np.linalg.norm(ref2)
    # the same product in three column chunks whose all-reduces overlap the next chunk's product (at config-4 sizes the
    # default; forced here), and the rank-one epilogue of the non-transposed product on the local rows
    Ad.reduce_chunks, Ad.chunk_min_cols = 3, 16
    trips = Ad.round_trips
    W2 = Ad.new_vectors(57, 4)
    Ad.apply(Yv, W2, transp=True)
    assert Ad.round_trips - trips == (2 if size > 1 else 0)      # 57 columns in chunks of 32 (16-column granularity)
    assert np.linalg.norm(W2.data() - ref2) < 1e-4 * np.linalg.norm(ref2)
    from raleigh_amd.algebra.hip.matrix import coefficients_into
    from raleigh_amd.algebra.hip.memory import DeviceBuffer
    cvec = Vectors(rng.standard_normal((1, 57)).astype(np.float32))
    cbuf = DeviceBuffer(64, zero=False)
    coefficients_into(cbuf.ptr, X, cvec)                       # c_j = <x_j, cvec>
    Y2 = Ad.new_vectors(203, 4)
    Ad.apply_r1(X, Y2, False, None, cbuf.ptr)                  # A x - e c^T
    ref3 = ref - (x @ cvec.data()[0])[:, None]
    assert np.linalg.norm(Y2.data() - ref3) < 1e-5 * np.linalg.norm(ref3)
    assert abs(Ad.frobenius2() - float(np.sum(a.astype(np.float64) ** 2))) < 1e-3 * np.sum(a ** 2)
    np.random.seed(1)
    A, sigma, u, v = generate(400, 150, 60, pca=True)
    off = partition(400, size)
    mat = ShardedAMatrix(A[off[rank]:off[rank + 1], :], comm)
    np.random.seed(1)
    mean, trans, comps = pca(mat, npc=12)
    assert trans.shape == (400, 12) and comps.shape == (12, 150)
    As = A - A.mean(axis=0, keepdims=True)
    exact = np.linalg.svd(As.astype(np.float64), compute_uv=False)[:12]
    sv = np.linalg.norm(trans, axis=0)
    assert np.max(np.abs(sv - exact) / exact[0]) < 2e-3
    em, ef = pca_error(A, mean, trans, comps)
    np.random.seed(1)
    mean1, trans1, comps1 = pca(A, npc=12)           # the unsharded path on the same data
    em1, ef1 = pca_error(A, mean1, trans1, comps1)
    assert abs(ef - ef1) < 0.02 * ef1
    dist.barrier()
    if rank == 0:
        print('DIST_OK world=%d' % size)
    dist.destroy_process_group()


def fake_lib_calls():
    from raleigh_amd import _lib
    return _lib.library().calls


def off_len(n, size, rank):
    from raleigh_amd.algebra.hip.dist import partition
    off = partition(n, size)
    return int(off[rank + 1] - off[rank])


if __name__ == '__main__':
    main()
