"""Row-sharding of the vector blocks over the GPUs of one node.

New relative to the reference, which has no distributed layer at all
(SURVEY 2.1): one process per GPU (``torch.distributed``, backend "nccl" = RCCL
over xGMI), GPU p owns rows [off[p], off[p+1]) of every block.

* ``multiply / add / copy / scale`` are local (coefficients are replicated);
* ``dot / dots`` = local partial + ONE all-reduce(sum) of the m x k (or m)
  scalars, issued on the stream the kernels run on;
* the sparse operator is row-sharded with the same partition and needs one
  exchange step per application: the rows of X referenced by off-shard column
  indices (send/receive lists fixed at construction).

torch is plumbing here (process group, communication buffers); all arithmetic
stays in librlhip.so.  With the "gloo" backend the same code runs on host
buffers, which is how the CPU test tier covers it (world_size 2).
"""

import ctypes

import numpy as np
import scipy.sparse as scs

from ... import _lib
from .vectors import Vectors, ReductionBatch
from .memory import upload, download
from .sparse import CsrOperator, full_from_upper

_REAL = {np.float32: np.float32, np.float64: np.float64,
         np.complex64: np.float32, np.complex128: np.float64}


class Comm:
    """Process group + communication buffers of one rank."""

    def __init__(self, group=None, force_collectives=None):
        import os
        import torch
        import torch.distributed as dist
        # RLH_FORCE_COLLECTIVES=1 (or force_collectives=True): issue the all-reduce and the halo
        # send / receive even with ONE rank (a rank then exchanges with itself), so that the RCCL calls
        # and their ordering with the kernels' stream run on a single GPU
        self.force = bool(int(os.environ.get('RLH_FORCE_COLLECTIVES', '0'))) if force_collectives is None \
            else bool(force_collectives)
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.on_device = (dist.get_backend(group) == 'nccl')
        if self.on_device:
            dev = _lib.default_device()
            torch.cuda.set_device(dev)
            L = _lib.lib(dev)
            # One explicit (non-default) stream carries the kernels of librlhip AND is torch's
            # current stream, so RCCL collectives / copies issued through torch are ordered
            # with them.  (The legacy default stream has handle 0, which rlh_set_stream
            # reads as "use the library's own stream" -- hence a dedicated stream.)
            self.stream = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(self.stream)
            _lib.check(L.rlh_set_stream(self.stream.cuda_stream))
            self.device = torch.device('cuda', dev)
        else:
            self.device = torch.device('cpu')
        self._red = None
        self._views = {}
        self.forced_halo_rows = None       # (kept for callers of earlier rounds; the forced one-rank run now cuts the shard in two)
        self._shm, self._shm_slot = None, 0
        self._setup_host_reduce()

    def _setup_host_reduce(self):
        """Small reductions end on the host (the solver reads them as NumPy arrays): with all ranks on ONE node they are
        summed in a shared-memory segment (rlh_shm_allreduce: every rank fetches its partial, the slots are added in
        rank order -- the same bits on every rank) instead of one RCCL launch + copy each.  RLH_HOST_REDUCE=0 keeps RCCL
        for everything; a one-rank run (forced collectives) keeps RCCL unless RLH_HOST_REDUCE=2.  Larger reductions (the
        N x k blocks of the dense transposed product) and the halo exchange always go through RCCL."""
        import os
        import socket
        import uuid
        mode = os.environ.get('RLH_HOST_REDUCE', '1')
        if mode == '0' or (self.size == 1 and mode != '2'):
            return
        dist = self.dist
        hosts = [None] * self.size
        dist.all_gather_object(hosts, socket.gethostname(), group=self.group)
        name = ['/rlh_%d_%s' % (os.getpid(), uuid.uuid4().hex[:10])] if self.rank == 0 else [None]
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast_object_list(name, src=src, group=self.group)
        slot = 1 << 18
        handle, ok = ctypes.c_void_p(), False
        if len(set(hosts)) == 1:
            try:
                _lib.check(_lib.library().rlh_shm_create(ctypes.byref(handle), name[0].encode(), self.rank, self.size, slot))
                ok = True
            except Exception:                                   # (whatever it was: the ranks decide together below)
                ok = False
        oks = [None] * self.size
        dist.all_gather_object(oks, ok, group=self.group)          # (also: every rank holds its mapping from here on)
        if self.rank == 0 and oks[0]:
            _lib.library().rlh_shm_unlink(name[0].encode())         # the name goes, the mappings stay: nothing is left in /dev/shm
        if all(oks):
            self._shm, self._shm_slot = handle, slot
        elif ok:
            _lib.library().rlh_shm_destroy(handle)

    def __del__(self):
        h, self._shm = getattr(self, '_shm', None), None
        if h:
            try:
                _lib.library().rlh_shm_destroy(h)
            except Exception:
                pass

    def buffer(self, nbytes):
        """A communication buffer (device memory under RCCL, host memory under gloo)."""
        return self.torch.empty(max(int(nbytes), 1), dtype=self.torch.uint8, device=self.device)

    def reduction_buffer(self, nbytes):
        if self._red is None or self._red.numel() < nbytes:
            self._red = self.buffer(max(nbytes, 1 << 16))
            self._views.clear()
        return self._red

    def sums_on_host(self, nbytes):
        """Whether a reduction of `nbytes` goes through the node's shared-memory segment (the same answer on every rank)."""
        return self._shm is not None and nbytes <= self._shm_slot and (self.size > 1 or self.force)

    def sum_on_host(self, out):
        """out (this rank's partial, a host array) <- the sum over the ranks, in rank order; returns out."""
        real = np.dtype(_REAL[out.dtype.type])
        _lib.check(_lib.library().rlh_shm_allreduce(self._shm, _lib.DTYPE_CODE[real.type], out.nbytes // real.itemsize,
                                                    _lib.host_ptr(out)))
        return out

    def allreduce_from_device(self, buf, np_dtype, count):
        """Sums `count` elements of dtype np_dtype held in `buf` over the ranks and
        returns them as a host array."""
        np_dtype = np.dtype(np_dtype)
        if self.sums_on_host(count * np_dtype.itemsize):
            out = np.empty((count,), dtype=np_dtype)
            if self.on_device:      # this rank's partial: pinned staging + one stream synchronisation inside the library
                _lib.check(_lib.lib().rlh_fetch(_lib.host_ptr(out), buf.data_ptr(), out.nbytes))
            else:
                out.view(np.uint8)[:] = buf[:out.nbytes].numpy()
            return self.sum_on_host(out)
        key = (buf.data_ptr(), np_dtype.str, count)
        view = self._views.get(key)
        if view is None:
            real = np.dtype(_REAL[np_dtype.type])
            nreal = count * (2 if np_dtype.kind == 'c' else 1)
            tdt = self.torch.float32 if real == np.float32 else self.torch.float64
            view = buf[:nreal * real.itemsize].view(tdt)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = view
        if self.size > 1 or self.force:
            self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group)
        out = np.empty((count,), dtype=np_dtype)
        if self.on_device:      # pinned staging + one stream synchronisation inside the library
            _lib.check(_lib.lib().rlh_fetch(_lib.host_ptr(out), buf.data_ptr(), out.nbytes))
        else:
            out.view(np.uint8)[:] = buf[:out.nbytes].numpy()
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)


def partition(n, size):
    """Row offsets of a balanced contiguous partition, rounded to 64 rows so that
    every shard starts on a wavefront / cache-line boundary."""
    per = -(-n // size)
    per = -(-per // 64) * 64
    off = [min(p * per, n) for p in range(size + 1)]
    off[size] = n
    return np.array(off, dtype=np.int64)


class ShardedVectors(Vectors):
    """Vectors whose rows are distributed over the ranks of `comm`."""

    def __init__(self, arg, nvec=0, data_type=None, shallow=False, comm=None, offsets=None):
        if isinstance(arg, ShardedVectors):
            comm, offsets, gdim = arg._comm, arg._offsets, arg._gdim
            super().__init__(arg, shallow=shallow)
        elif isinstance(arg, np.ndarray):          # a GLOBAL (nvec, n) array, same on every rank
            if comm is None:
                raise ValueError('ShardedVectors needs a communicator')
            gdim = arg.shape[1]
            offsets = partition(gdim, comm.size) if offsets is None else offsets
            r0, r1 = offsets[comm.rank], offsets[comm.rank + 1]
            super().__init__(np.ascontiguousarray(arg[:, r0:r1]))
        else:
            if comm is None:
                raise ValueError('ShardedVectors needs a communicator')
            gdim = int(arg)
            offsets = partition(gdim, comm.size) if offsets is None else offsets
            nloc = int(offsets[comm.rank + 1] - offsets[comm.rank])
            super().__init__(nloc, nvec, data_type)
        self._comm, self._offsets, self._gdim = comm, offsets, gdim

    # ---- global views
    def dimension(self):
        return self._gdim

    def local_dimension(self):
        return self._vdim

    def comm(self):
        return self._comm

    def offsets(self):
        return self._offsets

    def new_vectors(self, arg=0, dim=None, data_type=None):
        if isinstance(arg, np.ndarray):
            return ShardedVectors(arg, comm=self._comm)
        dt = self.data_type() if data_type is None else data_type
        if dim is not None and dim != self._gdim:
            return ShardedVectors(dim, int(arg), dt, comm=self._comm)
        return ShardedVectors(self._gdim, int(arg), dt, comm=self._comm, offsets=self._offsets)

    def clone(self):
        return ShardedVectors(self)

    def reference(self):
        return ShardedVectors(self, shallow=True)

    def fill_random(self):
        # every rank draws the same global stream (one vector at a time) and keeps its rows:
        # the start vectors do not depend on the number of GPUs
        f, m = self.selected()
        r0, r1 = self._offsets[self._comm.rank], self._offsets[self._comm.rank + 1]
        if m * self._gdim >= self.DEVICE_RANDOM_THRESHOLD:
            # counter-based device generator: a function of (seed, vector, GLOBAL row), the seed
            # drawn from the numpy stream that every rank seeds alike
            if m > 0:
                self._fill_random_device(r0)
            return
        for i in range(m):
            row = np.random.rand(1, self._gdim)[:, r0:r1].astype(self.data_type())
            row *= 2
            row -= 1
            upload(self._ptr(f + i), self._ld * self._es, row)

    def fill(self, data):
        if isinstance(data, np.ndarray) and data.shape[1] == self._gdim and self._gdim != self._vdim:
            r0, r1 = self._offsets[self._comm.rank], self._offsets[self._comm.rank + 1]
            data = np.ascontiguousarray(data[:, r0:r1])
        super().fill(data)

    def local_data(self):
        return Vectors.data(self)

    def data(self, i=None):
        """GLOBAL host copy (nvec, n), assembled on every rank."""
        c = self._comm
        if c.on_device:
            # gathered on the devices (one all_gather straight from the block), then ONE copy per shard into the result:
            # no host staging of the local part, no re-upload (the ten eigenvectors of the 10^7-row solve: 0.30 -> 0.1 s)
            L = _lib.lib()
            first, m = (self._sel[0] + i, 1) if i is not None else self._sel
            es = self._es
            out = np.empty((m, self._gdim), dtype=self.data_type())
            if m > 0 and self._gdim > 0:
                maxloc = int(np.max(np.diff(self._offsets)))
                send = c.buffer(maxloc * m * es)
                if self._vdim > 0:
                    _lib.check(L.rlh_copy(self._code, self._vdim, m, self._ptr(first), self.ld(), send.data_ptr(), maxloc))
                if c.size > 1 or c.force:
                    recv = c.buffer(c.size * maxloc * m * es)
                    c.dist.all_gather_into_tensor(recv, send, group=c.group)
                else:
                    recv = send
                for p in range(c.size):
                    r0, r1 = int(self._offsets[p]), int(self._offsets[p + 1])
                    if r1 > r0:
                        _lib.check(L.rlh_copy2d(out.ctypes.data + r0 * es, self._gdim * es, recv.data_ptr() + p * maxloc * m * es,
                                                maxloc * es, (r1 - r0) * es, m, 1))
            return out[0] if i is not None else out
        loc = Vectors.data(self, i)
        if i is not None:
            loc = loc.reshape(1, -1)
        m = loc.shape[0]
        out = np.zeros((m, self._gdim), dtype=self.data_type())
        tt = c.torch
        maxloc = int(np.max(np.diff(self._offsets)))
        send = np.zeros((m, maxloc), dtype=self.data_type())
        send[:, :loc.shape[1]] = loc
        st = tt.from_numpy(send.view(_REAL[self.data_type()]))
        if c.on_device:
            st = st.to(c.device)
        parts = [tt.empty_like(st) for _ in range(c.size)]
        c.dist.all_gather(parts, st, group=c.group)
        for p in range(c.size):
            r0, r1 = self._offsets[p], self._offsets[p + 1]
            arr = parts[p].cpu().numpy().view(self.data_type())
            out[:, r0:r1] = arr[:, :r1 - r0]
        return out[0] if i is not None else out

    def gather_into(self, full):
        """The whole block on EVERY rank's device: `full` (plain Vectors of the global dimension, as many vectors
        selected) receives the selected vectors of all shards -- one all_gather on the kernels' stream, no host copy."""
        c, L = self._comm, _lib.lib()
        m = self.nvec()
        if full.nvec() != m or full.dimension() != self._gdim or full.data_type() != self.data_type():
            raise ValueError('gather_into needs a plain block of the global dimension and the same type')
        if m < 1:
            return
        es = self._es
        maxloc = int(np.max(np.diff(self._offsets)))
        nloc = self._vdim
        send = c.buffer(maxloc * m * es)
        if nloc > 0:
            _lib.check(L.rlh_copy(self._code, nloc, m, self.data_ptr(), self.ld(), send.data_ptr(), maxloc))
        recv = c.buffer(c.size * maxloc * m * es)
        if c.size > 1 or c.force:
            c.dist.all_gather_into_tensor(recv, send, group=c.group)
        else:
            recv = send
        for p in range(c.size):
            r0, r1 = int(self._offsets[p]), int(self._offsets[p + 1])
            if r1 > r0:
                _lib.check(L.rlh_copy(self._code, r1 - r0, m, recv.data_ptr() + p * maxloc * m * es, maxloc,
                                      full.data_ptr() + r0 * es, full.ld()))

    def take_rows_of(self, full):
        """This rank's rows of a plain block of the global dimension (the inverse of gather_into, local)."""
        m = self.nvec()
        if full.nvec() != m or full.dimension() != self._gdim:
            raise ValueError('take_rows_of needs a plain block of the global dimension')
        r0 = int(self._offsets[self._comm.rank])
        if m > 0 and self._vdim > 0:
            _lib.check(_lib.lib().rlh_copy(self._code, self._vdim, m, full.data_ptr() + r0 * self._es, full.ld(),
                                           self.data_ptr(), self.ld()))

    def reduction_batch(self):
        return ShardedReductionBatch(self)

    # ---- reductions: local partial + one all-reduce
    def dot(self, other):
        m, k = self.nvec(), other.nvec()
        q = np.zeros((k, m), dtype=self.data_type())
        if m == 0 or k == 0:
            return q
        c = self._comm
        buf = c.reduction_buffer(m * k * self._es)
        _lib.check(_lib.lib().rlh_gram(self._code, self._vdim, m, self._ptr(), self._ld, k,
                                       other._ptr(), other._ld, buf.data_ptr(), None))
        return c.allreduce_from_device(buf, self.data_type(), m * k).reshape(k, m).copy()

    def dots(self, other, transp=False):
        if transp:
            # per-ROW sums over the selected vectors (dense_cublas.py:175-221, used by
            # truncated_svd.py:183,195 and lra.py:321): every rank computes its own rows on the device,
            # the global vector of length n is assembled on every rank by one all-gather
            loc = Vectors.dots(self, other, transp=True)
            c = self._comm
            tt = c.torch
            maxloc = int(np.max(np.diff(self._offsets)))
            send = np.zeros((maxloc,), dtype=self.data_type())
            send[:loc.shape[0]] = loc
            st = tt.from_numpy(send.view(_REAL[self.data_type()]))
            if c.on_device:
                st = st.to(c.device)
            parts = [tt.empty_like(st) for _ in range(c.size)]
            c.dist.all_gather(parts, st, group=c.group)
            out = np.zeros((self._gdim,), dtype=self.data_type())
            for p in range(c.size):
                r0, r1 = self._offsets[p], self._offsets[p + 1]
                out[r0:r1] = parts[p].cpu().numpy().view(self.data_type())[:r1 - r0]
            return out
        m = self.nvec()
        if m == 0:
            return np.zeros((0,), dtype=self.data_type())
        c = self._comm
        buf = c.reduction_buffer(m * self._es)
        _lib.check(_lib.lib().rlh_dots(self._code, self._vdim, m, self._ptr(), self._ld,
                                       other._ptr(), other._ld, buf.data_ptr(), None))
        return c.allreduce_from_device(buf, self.data_type(), m).copy()


class ShardedReductionBatch(ReductionBatch):
    """Local partial results side by side in the communication buffer, ONE all-reduce, one fetch."""

    def _buffer(self, nbytes):
        self._buf = self._proto._comm.reduction_buffer(nbytes)
        return self._buf.data_ptr()

    def _collect(self, ptr, count):
        return self._proto._comm.allreduce_from_device(self._buf, self._proto.data_type(), count)


class ShardedSparseMatrix:
    """Row-sharded symmetric/Hermitian operator with a halo exchange per application.

    Every rank passes the same global SciPy matrix (upper triangle significant,
    as in raleigh/algebra/sparse_mkl.py:18-31) or, to avoid holding it everywhere,
    its own row block via ``local_rows=(csr_block, row0)``."""

    def __init__(self, matrix, comm, offsets=None):
        full = full_from_upper(matrix)
        n = full.shape[0]
        off = partition(n, comm.size) if offsets is None else offsets
        r0, r1 = int(off[comm.rank]), int(off[comm.rank + 1])
        self._setup(scs.csr_matrix(full[r0:r1, :]), r0, n, comm, off, int(full.nnz))

    @classmethod
    def from_local_rows(cls, rows, row0, n, comm, offsets):
        """From this rank's rows [row0, row0 + rows.shape[0]) of the FULL (both triangles)
        matrix with global column indices; nothing global is materialised."""
        self = cls.__new__(cls)
        rows = scs.csr_matrix(rows)
        assert rows.shape[1] == n and row0 == int(offsets[comm.rank])
        t = comm.torch.tensor([rows.nnz], dtype=comm.torch.int64, device=comm.device)
        comm.dist.all_reduce(t, group=comm.group)
        self._setup(rows, row0, n, comm, offsets, int(t.item()))
        return self

    def _setup(self, loc, r0, n, comm, off, nnz_global):
        self._comm = comm
        self._n = n
        self._dtype = loc.data.dtype.type
        self._offsets = off
        r1 = r0 + loc.shape[0]
        assert r1 == int(off[comm.rank + 1])
        loc.sort_indices()
        cols = loc.indices.astype(np.int64)
        own = (cols >= r0) & (cols < r1)
        if comm.size == 1 and comm.force:
            # one rank exchanging with itself, as TWO VIRTUAL RANKS: the shard is cut in the middle, a column is an own
            # column for the rows on its side of the cut and a halo column for the rows on the other side -- what the two
            # halves would be as real shards (for a stencil: one plane of halo rows in each direction; a row's own
            # column always an own column, every block next to the cut with one near and one far window).  The halo rows
            # go pack -> send to self -> receive -> halo block, so the whole exchange runs.  (Earlier rounds re-routed
            # EVERY reference to the last rows of the shard, diagonals included, which no real shard does: it hid two
            # layout rules that only real shards trip -- DESIGN 5.)
            cut = r0 + (((r1 - r0) // 2) // 8) * 8
            rows_of = np.repeat(np.arange(r0, r1, dtype=np.int64), np.diff(loc.indptr))
            own = (cols >= cut) == (rows_of >= cut)
        halo_cols = np.unique(cols[~own])                     # global ids, sorted => grouped by owner
        owner = np.searchsorted(off, halo_cols, side='right') - 1
        # local column numbering: own rows first, then the halo rows from a multiple of 8 on (the
        # 1-7 columns in between are never referenced: they are the padding of the local block's
        # leading dimension), so that no 16-byte piece of the library's staging loads lies across
        # the own / halo boundary whatever the shard size
        n_own_pad = -(-(r1 - r0) // 8) * 8
        new_idx = np.empty_like(cols)
        new_idx[own] = cols[own] - r0
        new_idx[~own] = n_own_pad + np.searchsorted(halo_cols, cols[~own])
        self._n_own = r1 - r0
        self._n_halo = int(halo_cols.size)
        self._ld_halo = -(-self._n_halo // 8) * 8        # leading dimension of the halo block (16-byte rows of bfloat16)
        # the column count is padded like the halo block (the extra columns are never referenced): the layout moves a
        # window that would reach past the last column to END there, and only a column count that is a multiple of 8
        # keeps such a window on the 16-byte pieces the bfloat16 staging needs
        ext = scs.csr_matrix((loc.data, new_idx.astype(np.int32), loc.indptr),
                             shape=(r1 - r0, n_own_pad + self._ld_halo))
        self._op = CsrOperator(ext, n_own=n_own_pad)
        self._nnz = nnz_global
        # receive plan: contiguous runs of halo rows per owner
        self._recv = []                                       # (peer, halo_start, count)
        for p in range(comm.size):
            idx = np.nonzero(owner == p)[0]
            if idx.size:
                self._recv.append((p, int(idx[0]), int(idx.size)))
        # tell every owner which of its rows this rank needs
        wants = [None] * comm.size
        mine = {p: (halo_cols[s:s + c] - off[p]).astype(np.int64) for p, s, c in self._recv}
        comm.dist.all_gather_object(wants, mine, group=comm.group)
        self._send = []                                       # (peer, device index list, count)
        for p in range(comm.size):
            if (p != comm.rank or comm.force) and wants[p] and comm.rank in wants[p]:
                idx = np.ascontiguousarray(wants[p][comm.rank])
                dbuf = comm.buffer(idx.nbytes)
                dbuf.copy_(comm.torch.from_numpy(idx.view(np.uint8)))
                self._send.append((p, dbuf, int(idx.size)))
        self._bufs = {}
        self._bf16 = None

    def size(self):
        return self._n

    def data_type(self):
        return np.dtype(self._dtype)

    def nnz_full(self):
        return self._nnz

    def halo_rows(self):
        return self._n_halo

    def _buffers(self, m, es):
        key = (m, es)
        if key not in self._bufs:
            c = self._comm
            ns = sum(cnt for _, _, cnt in self._send)
            nr = self._ld_halo
            self._bufs[key] = (c.buffer(ns * m * es), c.buffer(nr * m * es), c.buffer(nr * m * es))
        return self._bufs[key]

    def _start_exchange(self, x):
        """Packs the rows the peers need and posts the sends / receives; returns what
        `_finish_exchange` needs, or None when nothing is off-shard."""
        L = _lib.lib()
        code, ptr, ld = x._code, x.data_ptr(), x.ld()
        return self._start_exchange_raw(x.nvec(), x._es, lambda cnt, idx, out: _lib.check(
            L.rlh_gather_rows(code, cnt, idx, x.nvec(), ptr, ld, out, cnt)))

    def _start_exchange_raw(self, m, es, gather):
        """`gather(count, device index list, packed output)` packs rows of the block being exchanged;
        es: bytes per element (2 for the bfloat16 work blocks of the preconditioner)."""
        c = self._comm
        if not (self._n_halo > 0 or self._send):
            return None
        sendbuf, recvbuf, halo = self._buffers(m, es)
        ops, soff, roff = [], 0, 0
        for p, didx, cnt in self._send:                   # pack the rows each peer needs
            gather(cnt, didx.data_ptr(), sendbuf.data_ptr() + soff)
            ops.append(c.dist.P2POp(c.dist.isend, sendbuf[soff:soff + cnt * m * es], p, group=c.group))
            soff += cnt * m * es
        for p, hs, cnt in self._recv:
            ops.append(c.dist.P2POp(c.dist.irecv, recvbuf[roff:roff + cnt * m * es], p, group=c.group))
            roff += cnt * m * es
        works = c.dist.batch_isend_irecv(ops) if ops else []
        return works, recvbuf, halo, m, es

    def _finish_exchange(self, pending):
        """Waits for the transfers and assembles the halo block; returns (device pointer, leading
        dimension) or (None, 0)."""
        if pending is None:
            return None, 0
        works, recvbuf, halo, m, es = pending
        L = _lib.lib()
        for w in works:
            w.wait()
        roff = 0
        for p, hs, cnt in self._recv:                     # peer blocks (ld = cnt) -> one halo block (ld = n_halo)
            _lib.check(L.rlh_copy2d(halo.data_ptr() + hs * es, self._ld_halo * es,
                                    recvbuf.data_ptr() + roff, cnt * es, cnt * es, m, 2))
            roff += cnt * m * es
        if self._n_halo > 0:
            return halo.data_ptr(), self._ld_halo
        return None, 0

    def _exchange_halo(self, x):
        """Brings the off-shard rows of x referenced by this rank's rows into one halo block;
        returns (device pointer, leading dimension) or (None, 0) when nothing is off-shard."""
        return self._finish_exchange(self._start_exchange(x))

    def _halo_slot(self, x):
        """The (not yet filled) halo block the interior pass may be handed."""
        if self._n_halo == 0:
            return None, 0
        return self._buffers(x.nvec(), x._es)[2].data_ptr(), self._ld_halo

    def apply(self, x, y):
        m = x.nvec()
        if m != y.nvec():
            raise ValueError('Numbers of input and output vectors differ')
        if x.dimension() != self._n or y.dimension() != self._n:
            raise ValueError('Matrix and vectors dimensions incompatible')
        # the rows that reference no off-shard column are multiplied while the exchange is in
        # flight (the transfers run on the communication library's own stream; only `wait`
        # orders them with the kernels' stream); the other rows follow it
        pending = self._start_exchange(x)
        if pending is None:
            self._op.apply_ptr(m, x.data_ptr(), x.ld(), y.data_ptr(), y.ld())
            return
        hp, ldh = self._halo_slot(x)
        self._op.apply_ptr(m, x.data_ptr(), x.ld(), y.data_ptr(), y.ld(), hp, ldh, part=1)
        halo_ptr, ldh = self._finish_exchange(pending)
        self._op.apply_ptr(m, x.data_ptr(), x.ld(), y.data_ptr(), y.ld(), halo_ptr, ldh, part=2)

    def supports_bf16(self):
        """Whether EVERY rank's shard takes the bfloat16 Chebyshev step: the layout conditions (staging groups inside the
        column range and on multiples of 8 columns, halo block on 16-byte rows) depend on the shard, so the ranks agree on
        the answer once (all-reduce MIN) -- a rank that found out by a failed launch would already have posted its 2-byte
        halo messages, and would restart with 4-byte ones while its peers went on."""
        if self._bf16 is None:
            c = self._comm
            mine = 1 if self._op.bf16_ready(self._ld_halo) else 0
            t = c.torch.tensor([mine], dtype=c.torch.int32, device=c.device)
            c.dist.all_reduce(t, op=c.dist.ReduceOp.MIN, group=c.group)
            self._bf16 = bool(int(t.item()))
        return self._bf16

    def cheb_step_bf16(self, m, y, p, b, cy, cp, cb):
        """The fused step on bfloat16 work blocks (sparse.Bf16Block) of the local rows: the halo rows
        travel as 2-byte elements, the exchange is overlapped with the interior rows as in apply()."""
        L = _lib.lib()
        pending = self._start_exchange_raw(m, 2, lambda cnt, idx, out: _lib.check(
            L.rlh_gather_rows_bf16(cnt, idx, m, y.ptr(), y.ld, out, cnt)))
        if pending is None:
            self._op.cheb_step_bf16(m, y, p, b, cy, cp, cb)
            return
        hp = self._buffers(m, 2)[2].data_ptr() if self._n_halo else None
        self._op.cheb_step_bf16(m, y, p, b, cy, cp, cb, hp, self._ld_halo, part=1)
        halo_ptr, ldh = self._finish_exchange(pending)
        self._op.cheb_step_bf16(m, y, p, b, cy, cp, cb, halo_ptr, ldh, part=2)

    def cheb_step(self, y, p, b, cy, cp, cb):
        """Fused step of the three-term Chebyshev semi-iteration on row-sharded blocks
        (p = cy y + cp p + cb (b - A y), one halo exchange of y overlapped with the interior rows)."""
        pending = self._start_exchange(y)
        if pending is None:
            self._op.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb)
            return
        hp, ldh = self._halo_slot(y)
        self._op.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb, hp, ldh, part=1)
        halo_ptr, ldh = self._finish_exchange(pending)
        self._op.cheb_step_ptr(y.nvec(), y, p, b, cy, cp, cb, halo_ptr, ldh, part=2)


class ShardedDenseMatrix:
    """Dense operator whose ROWS are distributed over the ranks (BASELINE configs 2/4: the PCA
    data matrix, samples sharded).  Vectors of the row dimension M are row-sharded
    (ShardedVectors with this matrix' partition); vectors of the column dimension N are
    REPLICATED plain Vectors (every rank holds and updates an identical copy).

      apply(x, y)              y_p = A_p x            local GEMM, no communication
      apply(z, y, transp=True) y = sum_p A_p^H z_p    local GEMM + one all-reduce of the N x k block

    Same surface as Matrix (dense_cublas.py:635-776: shape/order/data_type/dots/new_vectors/apply)."""

    def __init__(self, local_rows, comm, offsets=None, global_rows=None):
        from .matrix import Matrix
        self._comm = comm
        if isinstance(local_rows, Matrix):         # this rank's rows already in HBM (e.g. built on the device)
            self._loc = local_rows
        elif isinstance(local_rows, Vectors):
            self._loc = Matrix(local_rows)
        else:
            self._loc = Matrix(np.ascontiguousarray(local_rows))
        if self._loc.order() != 'C_CONTIGUOUS':
            raise ValueError('the local rows of a row-sharded matrix must be C-contiguous')
        mloc, n = self._loc.shape()
        counts = [None] * comm.size
        comm.dist.all_gather_object(counts, int(mloc), group=comm.group)
        off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        if offsets is not None:
            assert np.array_equal(off, offsets), 'local row counts do not match the partition'
        self._offsets = off
        self._shape = (int(off[-1]), n)
        if global_rows is not None:
            assert global_rows == self._shape[0]
        self._dtype = self._loc.data_type()
        self._chunk_bufs = {}
        self.reduce_chunks = 4          # column chunks of A_p whose all-reduce overlaps the next chunk's product
        self.chunk_min_cols = 1024      # ... when every chunk keeps at least this many columns
        self.round_trips = 0

    # the rank-one epilogue is local for y = A x - u c^T (every rank subtracts its own rows of u c^T); the transposed
    # product would need the term once, after the reduction: not offered (interfaces/pca.py then takes the unfused steps)
    r1_transposed = False

    @classmethod
    def from_global(cls, a, comm):
        """Every rank passes the same global array and keeps its row block."""
        off = partition(a.shape[0], comm.size)
        return cls(a[off[comm.rank]:off[comm.rank + 1], :], comm, offsets=off)

    def shape(self):
        return self._shape

    def order(self):
        return 'C_CONTIGUOUS'

    def data_type(self):
        return self._dtype

    def is_complex(self):
        return self._loc.is_complex()

    def comm(self):
        return self._comm

    def offsets(self):
        return self._offsets

    def local(self):
        return self._loc

    def new_vectors(self, dim=None, nv=0):
        m, n = self._shape
        if dim is None:
            dim = n
        if dim == m and m != n:
            return ShardedVectors(m, nv, self._dtype, comm=self._comm, offsets=self._offsets)
        if dim == n:
            return Vectors(n, nv, self._dtype)
        raise ValueError('a row-sharded matrix creates vectors of its row or column dimension only')

    def dots(self):
        """Squared row norms of the LOCAL rows (use frobenius2() for the global sum)."""
        return self._loc.dots()

    def frobenius2(self):
        loc = np.array([float(np.sum(np.abs(self._loc.dots())))], dtype=np.float64)
        t = self._comm.torch.from_numpy(loc)
        if self._comm.on_device:
            t = t.to(self._comm.device)
        self._comm.dist.all_reduce(t, group=self._comm.group)
        return float(t.cpu().numpy()[0])

    def apply(self, x, y, transp=False):
        self.apply_r1(x, y, transp)

    def apply_r1(self, x, y, transp=False, u=None, c=None):
        """y = Op(A) x (- u c^T for the non-transposed product: Matrix.apply_r1; u None = ones)."""
        m, n = self._shape
        k = x.nvec()
        if k != y.nvec():
            raise ValueError('Numbers of input and output vectors differ')
        if not transp:
            if not isinstance(y, ShardedVectors) or isinstance(x, ShardedVectors):
                raise ValueError('apply needs replicated input and row-sharded output vectors')
            if x.dimension() != n or y.dimension() != m:
                raise ValueError('Matrix and vectors dimensions incompatible')
            if u is not None and not isinstance(u, ShardedVectors):
                raise ValueError('the rank-one vector must be row-sharded like the result')
            self._loc.apply_r1(x, _LocalView(y), False, None if u is None else _LocalView(u), c)
            return
        if c is not None:
            raise ValueError('the transposed product of a row-sharded matrix takes no rank-one term')
        if not isinstance(x, ShardedVectors) or isinstance(y, ShardedVectors):
            raise ValueError('apply(transp=True) needs row-sharded input and replicated output vectors')
        if x.dimension() != m or y.dimension() != n:
            raise ValueError('Matrix and vectors dimensions incompatible')
        cm, L = self._comm, _lib.lib()
        es = y._es
        mloc = self._loc.shape()[0]
        real = np.dtype(_REAL[np.dtype(self._dtype).type])
        tdt = cm.torch.float32 if real == np.float32 else cm.torch.float64
        per_elem = 2 if np.dtype(self._dtype).kind == 'c' else 1
        reduce_ = cm.size > 1 or cm.force
        # y = sum_p A_p^H x_p in column chunks of A_p (= row chunks of y): the all-reduce of chunk i runs on RCCL's
        # stream while the matrix cores work on chunk i + 1; every element of A_p is still read once
        nch = self.reduce_chunks if (reduce_ and n >= self.chunk_min_cols * self.reduce_chunks) else 1
        step = -(-n // nch)
        step = -(-step // 16) * 16
        pending = []
        for i in range(nch):
            j0, j1 = i * step, min(n, (i + 1) * step)
            if j1 <= j0:
                break
            nc = j1 - j0
            key = (i, nc, k, es)
            buf = self._chunk_bufs.get(key)
            if buf is None:
                if len(self._chunk_bufs) > 16:
                    self._chunk_bufs.clear()
                buf = self._chunk_bufs[key] = cm.buffer(nc * k * es)
            _lib.check(L.rlh_dense_apply(y._code, mloc, nc, self._loc.data_ptr() + j0 * es, self._loc.lda(), 0, 1, k,
                                         x.data_ptr(), x.ld(), buf.data_ptr(), nc))
            work = None
            if reduce_:
                work = cm.dist.all_reduce(buf[:nc * k * per_elem * real.itemsize].view(tdt), group=cm.group, async_op=True)
                self.round_trips += 1
            pending.append((j0, nc, buf, work))
        for j0, nc, buf, work in pending:
            if work is not None:
                work.wait()
            _lib.check(L.rlh_copy(y._code, nc, k, buf.data_ptr(), nc, y.data_ptr() + j0 * es, y.ld()))


class _LocalView:
    """Presents the local shard of a ShardedVectors to a local operator (dimension = local rows)."""

    def __init__(self, v):
        self._v = v

    def __getattr__(self, name):
        return getattr(self._v, name)

    def dimension(self):
        return self._v.local_dimension()


class ShardedAMatrix:
    """Counterpart of AMatrix (raleigh/algebra/dense_matrix.py) for a row-sharded data matrix."""

    def __init__(self, local_rows, comm, offsets=None):
        """local_rows: this rank's rows as an ndarray, or already in HBM as a Matrix / Vectors (one vector per row)."""
        self.__op = ShardedDenseMatrix(local_rows, comm, offsets)
        self.__comm = comm

    def as_operator(self):
        return self.__op

    def arch(self):
        return 'hip'

    def gpu(self):
        return None

    def data_type(self):
        return self.__op.data_type()

    def shape(self):
        return self.__op.shape()

    def order(self):
        return self.__op.order()

    def frobenius2(self):
        return self.__op.frobenius2()
