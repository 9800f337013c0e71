"""MI355X implementation of RALEIGH's abstract Vectors type.

Same method surface, argument meaning and error behaviour as the reference
backends (contract: raleigh/core/solver.py:22-96; semantic oracle:
raleigh/algebra/dense_numpy.py; the backend this one stands in for:
raleigh/algebra/dense_cublas.py), so ``raleigh/core/solver.py`` can drive it
unchanged.  All arithmetic runs in librlhip.so (include/rlhip.h); this file is
bookkeeping: storage, selection windows, host<->device conventions.

Storage: one device allocation holding `capacity` vectors of `ld >= dim`
elements each (column-major dim x capacity, the reference's (nvec, dim)
C-ordered array); ld is padded to a multiple of 32 elements so every vector
starts 128-byte aligned and the kernels take their 16-byte vector-load paths.
"""

import numbers

import numpy as np

from ... import _lib
from .memory import DeviceBuffer, upload, download

_LD_ALIGN = 32          # elements
_MIN_INC, _MAX_INC = 16, 1024     # append() growth (dense_cublas.py:424-425)


def _padded(n):
    return max(_LD_ALIGN, (int(n) + _LD_ALIGN - 1) // _LD_ALIGN * _LD_ALIGN)


class Vectors:
    '''HIP/CDNA4 implementation of the Vectors type.'''

    # ------------------------------------------------------------ construction
    def __init__(self, arg, nvec=0, data_type=None, shallow=False):
        self._off = 0          # element offset of vector 0 inside the buffer
        if isinstance(arg, Vectors):
            f, k = arg.selected()
            self._set_type(arg.data_type())
            self._vdim, self._ld = arg._vdim, arg._ld
            if shallow:          # a view of the selected window (dense_ndarray.py:55-56)
                self._buf = arg._buf
                self._off = arg._off + f * arg._ld
            else:
                self._buf = DeviceBuffer(max(k, 1) * self._ld * self._es, zero=False)
                if k > 0:
                    _lib.check(_lib.lib().rlh_copy(self._code, self._vdim, k, arg._ptr(), arg._ld,
                                                   self._buf.ptr, self._ld))
            nv = k
        elif _is_matrix(arg):
            if arg.order() != 'C_CONTIGUOUS':
                raise ValueError('Vectors data must be C_CONTIGUOUS')
            nv, n = arg.shape()
            self._set_type(arg.data_type())
            self._vdim, self._ld = n, arg.lda()
            if shallow:
                self._buf = arg.matrix_data()
            else:
                self._buf = DeviceBuffer(max(nv, 1) * self._ld * self._es, zero=False)
                _lib.check(_lib.lib().rlh_d2d(self._buf.ptr, arg.data_ptr(), nv * self._ld * self._es))
        elif isinstance(arg, np.ndarray):
            if arg.ndim != 2:
                raise ValueError('Vectors data must be a 2D array')
            nv, n = arg.shape
            self._set_type(arg.dtype.type)
            self._vdim, self._ld = n, _padded(n)
            self._buf = DeviceBuffer(max(nv, 1) * self._ld * self._es)
            upload(self._buf.ptr, self._ld * self._es, np.ascontiguousarray(arg))
        elif isinstance(arg, numbers.Number):
            self._set_type(np.float64 if data_type is None else data_type)
            n, nv = int(arg), int(nvec)
            assert nv >= 0
            self._vdim, self._ld = n, _padded(n)
            self._buf = DeviceBuffer(nv * self._ld * self._es) if nv > 0 else None
        else:
            raise ValueError('wrong argument %s in constructor' % repr(type(arg)))
        self._nvec = nv
        self._mvec = nv
        self._sel = (0, nv)
        self._inc = _MIN_INC

    def _set_type(self, dt):
        dt = np.dtype(dt).type
        if dt not in _lib.DTYPE_CODE:
            raise ValueError('data type %s not supported' % repr(dt))
        self._dtype = dt
        self._code = _lib.DTYPE_CODE[dt]
        self._es = _lib.DTYPE_SIZE[dt]
        self._is_complex = dt in (np.complex64, np.complex128)

    # ------------------------------------------------------------ device addressing
    def _ptr(self, first=None):
        """Device address of the first selected vector (or of vector `first`)."""
        f = self._sel[0] if first is None else first
        base = self._buf.ptr if self._buf is not None else 0
        return base + (self._off + f * self._ld) * self._es

    def data_ptr(self):
        return self._ptr()

    def all_data_ptr(self):
        return self._ptr(0)

    def ld(self):
        return self._ld

    def data_size(self):
        return self._es

    def vectors_data(self):
        return self._buf

    # ------------------------------------------------------------ solver-facing methods
    def new_vectors(self, arg=0, dim=None, data_type=None):
        # data_type is an extension (mixed-precision work blocks); the reference has (arg, dim)
        if isinstance(arg, numbers.Number):
            return Vectors(self.dimension() if dim is None else dim, int(arg),
                           self.data_type() if data_type is None else data_type)
        return Vectors(arg)

    def dimension(self):
        return self._vdim

    def select(self, nv, first=0):
        assert nv <= self._nvec and first >= 0
        self._sel = (int(first), int(nv))

    def selected(self):
        return self._sel

    def clone(self):
        return Vectors(self)

    def append(self, other, axis=0):
        if other.nvec() < 1:
            return
        L = _lib.lib()
        if axis == 1:          # lengthen every vector (dense_cublas.py:44-72)
            m, n = self.shape()
            l, n_other = other.shape()
            if m != l:
                raise ValueError('Cannot append %d vectors to %d vectors' % (l, m))
            if self.data_type() != other.data_type():
                raise ValueError('Cannot append %s vectors to %s vectors'
                                 % (repr(other.data_type()), repr(self.data_type())))
            n_new = n + n_other
            ld_new = _padded(n_new)
            buf = DeviceBuffer(max(m, 1) * ld_new * self._es)
            es = self._es
            _lib.check(L.rlh_copy2d(buf.ptr, ld_new * es, self._ptr(0), self._ld * es, n * es, m, 2))
            _lib.check(L.rlh_copy2d(buf.ptr + n * es, ld_new * es, other._ptr(0), other._ld * es,
                                    n_other * es, m, 2))
            self._buf, self._off, self._ld, self._vdim, self._mvec = buf, 0, ld_new, n_new, m
            return
        if self.data_type() != other.data_type() or self.dimension() != other.dimension():
            raise ValueError('Cannot append vectors of different type or dimension')
        i, m = self.selected()
        j, l = other.selected()
        nvec = i + m + l
        if nvec > self._mvec or self._buf is None or self._off != 0:
            mvec = ((nvec - 1) // self._inc + 1) * self._inc
            buf = DeviceBuffer(mvec * self._ld * self._es, zero=False)
            if i + m > 0:
                _lib.check(L.rlh_copy(self._code, self._vdim, i + m, self._ptr(0), self._ld,
                                      buf.ptr, self._ld))
            self._buf, self._off, self._mvec = buf, 0, mvec
            if self._inc < _MAX_INC:
                self._inc *= 2
        _lib.check(L.rlh_copy(self._code, self._vdim, l, other._ptr(), other._ld,
                              self._ptr(i + m), self._ld))
        self._nvec = nvec
        self.select_all()

    def nvec(self):
        return self._sel[1]

    def data_type(self):
        return self._dtype

    # blocks of at least this many elements are filled by the device generator
    DEVICE_RANDOM_THRESHOLD = 1 << 22

    def fill_random(self):
        # Small blocks: host RNG then H2D, as the reference does (dense_cublas.py:119-131), so
        # that numpy.random.seed(...) gives the same start vectors on every backend.  Large
        # blocks (where the host draw alone costs 0.6 s at 10^7 x 20): the library's counter-based
        # generator, seeded from the same numpy stream (still reproducible under numpy.random.seed).
        m, n = self.nvec(), self._vdim
        if m < 1:
            return
        if m * n >= self.DEVICE_RANDOM_THRESHOLD:
            self._fill_random_device(0)
            return
        data = np.random.rand(m, n).astype(self.data_type())
        data *= 2
        data -= 1
        upload(self._ptr(), self._ld * self._es, data)

    def _fill_random_device(self, row0):
        seed = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
        _lib.check(_lib.lib().rlh_fill_random(self._code, self._vdim, self.nvec(), self._ptr(), self._ld,
                                              seed, int(row0), 0))

    def copy(self, other, ind=None):
        L = _lib.lib()
        i, m = self.selected()
        j, l = other.selected()
        if ind is None:
            assert m == l
            _lib.check(L.rlh_copy(self._code, self._vdim, m, self._ptr(), self._ld,
                                  other._ptr(), other._ld))
        else:
            idx = np.ascontiguousarray(ind, dtype=np.int64)
            if idx.size and (idx.min() < 0 or idx.max() >= self._nvec):
                raise IndexError('vector index out of range in copy()')
            _lib.check(L.rlh_copy_cols(self._code, self._vdim, idx.size, _lib.host_ptr(idx),
                                       self._ptr(0), self._ld, other._ptr(), other._ld))

    def scale(self, s, multiply=False):
        m = self.nvec()
        if m < 1:
            return
        s = np.asarray(s)[:m]
        if self._is_complex:
            sd = np.ascontiguousarray(s.astype(np.complex128)).view(np.float64)
        else:
            sd = np.ascontiguousarray(s.real if np.iscomplexobj(s) else s, dtype=np.float64)
        _lib.check(_lib.lib().rlh_scale_cols(self._code, self._vdim, m, _lib.host_ptr(sd),
                                             1 if multiply else 0, self._ptr(), self._ld))

    def dots(self, other, transp=False):
        L = _lib.lib()
        m, n = self.nvec(), self._vdim
        if transp:
            w = np.zeros((n,), dtype=self.data_type())
            if n > 0:
                tmp = DeviceBuffer(n * self._es, zero=False)
                _lib.check(L.rlh_dots_transp(self._code, n, m, self._ptr(), self._ld,
                                             other._ptr(), other._ld, tmp.ptr))
                _lib.check(L.rlh_d2h(_lib.host_ptr(w), tmp.ptr, n * self._es))
            return w
        v = np.zeros((m,), dtype=self.data_type())
        if m > 0:
            _lib.check(L.rlh_dots(self._code, n, m, self._ptr(), self._ld, other._ptr(), other._ld,
                                  None, _lib.host_ptr(v)))
        return v

    def dot(self, other):
        m, k = self.nvec(), other.nvec()
        q = np.zeros((k, m), dtype=self.data_type())
        if m > 0 and k > 0:
            _lib.check(_lib.lib().rlh_gram(self._code, self._vdim, m, self._ptr(), self._ld,
                                           k, other._ptr(), other._ld, None, _lib.host_ptr(q)))
        return q

    def _update(self, q, src, dst, alpha, beta):
        q = np.asarray(q)
        if q.dtype.type != self._dtype:
            q = q.astype(self._dtype)
        if q.ndim != 2 or any(st < 0 for st in q.strides) or any(st % q.itemsize for st in q.strides):
            q = np.ascontiguousarray(q)
        k, m = q.shape
        if k != src.nvec() or m != dst.nvec():
            raise ValueError('coefficient matrix shape %s does not match %d -> %d vectors'
                             % (repr(q.shape), src.nvec(), dst.nvec()))
        a = np.array([np.real(alpha), np.imag(alpha)], dtype=np.float64)
        _lib.check(_lib.lib().rlh_block_update(
            self._code, self._vdim, k, src._ptr(), src._ld, m, dst._ptr(), dst._ld,
            _lib.host_ptr(q), q.strides[0] // q.itemsize, q.strides[1] // q.itemsize,
            _lib.host_ptr(a), beta))

    def multiply(self, q, output):
        assert output.nvec() == q.shape[1]
        self._update(q, self, output, 1.0, 0)

    def add(self, other, s, q=None):
        L = _lib.lib()
        m = other.nvec()
        if np.isscalar(s):
            if q is None:
                a = np.array([np.real(s), np.imag(s)], dtype=np.float64)
                _lib.check(L.rlh_axpy(self._code, self._vdim, m, _lib.host_ptr(a),
                                      other._ptr(), other._ld, self._ptr(), self._ld))
            else:
                self._update(q, other, self, s, 1)
        else:
            sv = np.ascontiguousarray(np.asarray(s)[:m], dtype=self._dtype)
            _lib.check(L.rlh_axpy_cols(self._code, self._vdim, m, _lib.host_ptr(sv),
                                       other._ptr(), other._ld, self._ptr(), self._ld))

    # ------------------------------------------------------------ fused forms (not in the reference API)
    def combine(self, q, other, q_other, output):
        """output = self * q + other * q_other in one pass over both sources (the reference
        issues multiply + add, solver.py:1609-1656)."""
        def prep(a):
            a = np.asarray(a)
            if a.dtype.type != self._dtype:
                a = a.astype(self._dtype)
            if a.ndim != 2 or any(st < 0 for st in a.strides) or any(st % a.itemsize for st in a.strides):
                a = np.ascontiguousarray(a)
            return a
        q, q2 = prep(q), prep(q_other)
        m = output.nvec()
        if q.shape != (self.nvec(), m) or q2.shape != (other.nvec(), m):
            raise ValueError('coefficient matrices do not match the numbers of vectors')
        a = np.array([1.0, 0.0], dtype=np.float64)
        _lib.check(_lib.lib().rlh_block_update2(
            self._code, self._vdim, q.shape[0], self._ptr(), self._ld, _lib.host_ptr(q),
            q.strides[0] // q.itemsize, q.strides[1] // q.itemsize,
            q2.shape[0], other._ptr(), other._ld, _lib.host_ptr(q2),
            q2.strides[0] // q2.itemsize, q2.strides[1] // q2.itemsize,
            m, output._ptr(), output._ld, _lib.host_ptr(a), 0))

    def combine2(self, q_a, q_b, other, q_other_a, q_other_b, out_a, out_b):
        """out_a = self * q_a + other * q_other_a and out_b = self * q_b + other * q_other_b from ONE
        pass over self and other (the Rayleigh-Ritz update forms the new block and the new
        directions from the same two blocks: solver.py:1609-1656)."""
        dt = self._dtype
        q = np.ascontiguousarray(np.concatenate((np.asarray(q_a, dtype=dt), np.asarray(q_b, dtype=dt)), axis=1))
        q2 = np.ascontiguousarray(np.concatenate((np.asarray(q_other_a, dtype=dt), np.asarray(q_other_b, dtype=dt)), axis=1))
        ma, mb = out_a.nvec(), out_b.nvec()
        if q.shape != (self.nvec(), ma + mb) or q2.shape != (other.nvec(), ma + mb) or np.asarray(q_a).shape[1] != ma:
            raise ValueError('coefficient matrices do not match the numbers of vectors')
        _lib.check(_lib.lib().rlh_block_update2x2(
            self._code, self._vdim, q.shape[0], self._ptr(), self._ld, _lib.host_ptr(q), ma + mb, 1,
            q2.shape[0], other._ptr(), other._ld, _lib.host_ptr(q2), ma + mb, 1,
            ma, out_a._ptr(), out_a._ld, mb, out_b._ptr(), out_b._ld))

    def reduction_batch(self):
        """A ReductionBatch for blocks of this type and dimension."""
        return ReductionBatch(self)

    def lincomb(self, a, x, b, y):
        """self[i] = a[i] * x[i] + b[i] * y[i] (a, b scalars or per-vector arrays); self may be x or y.
        One pass instead of copy + add (solver.py:942-952)."""
        m = self.nvec()
        av = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=self._dtype), (m,)))
        bv = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=self._dtype), (m,)))
        _lib.check(_lib.lib().rlh_lincomb_cols(self._code, self._vdim, m, _lib.host_ptr(av), x._ptr(), x._ld,
                                               _lib.host_ptr(bv), y._ptr(), y._ld, self._ptr(), self._ld))

    def convert_to(self, other):
        """other = self converted to other's precision (float32 <-> float64, complex64 <-> complex128)."""
        if other.nvec() != self.nvec() or other._vdim != self._vdim:
            raise ValueError('mismatching shapes in convert_to()')
        _lib.check(_lib.lib().rlh_convert(self._code, other._code, self._vdim, self.nvec(), self._ptr(), self._ld,
                                          other._ptr(), other._ld))

    # ------------------------------------------------------------ other methods of the reference backends
    def shape(self):
        return (self._nvec, self._vdim)

    def first(self):
        return self._sel[0]

    def select_all(self):
        self.select(self._nvec)

    def reference(self):
        return Vectors(self, shallow=True)

    def is_complex(self):
        return self._is_complex

    def conjugate(self):
        if self._is_complex and self.nvec() > 0:
            _lib.check(_lib.lib().rlh_conj(self._code, self._vdim, self.nvec(), self._ptr(), self._ld))

    def zero(self):
        m = self.nvec()
        if m < 1:
            return
        # one fill over the window; the padding rows between the vectors belong to this block
        _lib.check(_lib.lib().rlh_memset(self._ptr(), 0, ((m - 1) * self._ld + self._vdim) * self._es))

    def fill(self, data):
        if not isinstance(data, np.ndarray):        # fill(value), as dense_ndarray.py:98-100
            data = np.full((self.nvec(), self._vdim), data, dtype=self._dtype)
        m, n = data.shape
        if m != self.nvec() or n != self._vdim:
            raise ValueError('mismatching dimensions in fill()')
        if m < 1:
            return
        if data.dtype.type != self._dtype:
            raise ValueError('mismatching data types in fill()')
        upload(self._ptr(), self._ld * self._es, np.ascontiguousarray(data))

    def data(self, i=None):
        """Host COPY of the selected vectors, shape (nvec, dim) (dense_cublas.py:620-629)."""
        m, n = self.nvec(), self._vdim
        if i is not None:
            v = np.ndarray((1, n), dtype=self._dtype)
            download(v, self._ptr(self._sel[0] + i), self._ld * self._es)
            return v[0]
        v = np.ndarray((m, n), dtype=self._dtype)
        if m > 0:
            download(v, self._ptr(), self._ld * self._es)
        return v

    def asarray(self):
        return self.data().T

    def orthogonalize(self, other):
        # q = <self, other>; self -= other * q (dense_numpy.py:117-123)
        q = self.dot(other)
        self.add(other, -1.0, q)
        return self.new_vectors(q)

    def svd(self):
        """Thin SVD in place: self (as an n x m matrix) = W diag(sigma) V^H with W orthonormal replacing
        self; returns (sigma, V) like dense_numpy.py:125-128 / dense_cublas.py:537-591 (gesvd 'O').

        The solver calls it exactly when the block has lost rank (raleigh/core/solver.py:877-885), so the
        algorithm must survive that: a rank-revealing block Gram-Schmidt built from the block operations
        (every pass over the n x m data is a Gram or a block update on the device; all m x m algebra on the
        host in double precision).  Invariant: original = W T.  Per round, on the columns not yet finished:
          1. rotate them by the eigenvectors of their Gram matrix (descending energy);
          2. peel those whose energy lies within `theta` of the round's largest -- their mutual
             orthogonality is accurate to eps / theta -- remove what they still carry of the finished
             columns, and orthonormalise them by two Cholesky-QR passes;
          3. remove them from the remaining columns (twice) and go on with the rest, which is smaller by
             sqrt(theta): no quantity is ever formed as a difference of squares of very different sizes,
             which is what limits a single Gram pass to cond < eps^-1/2.
        Columns whose energy has fallen to rounding level are dropped from T (an error of eps sigma_max in
        the reconstruction) and replaced by random directions orthonormalised against the rest, so W is
        orthonormal whatever the rank.  Last, T = Ur S V^H on the host and W <- W Ur."""
        import scipy.linalg as sla
        m = self.nvec()
        real_dt = np.float32 if self._dtype in (np.float32, np.complex64) else np.float64
        wide = np.complex128 if self._is_complex else np.float64
        if m < 1:
            return np.zeros((0,), dtype=real_dt), np.zeros((0, 0), dtype=self._dtype)
        f0 = self.selected()[0]
        eps = float(np.finfo(real_dt).eps)
        theta = 1e-6 if real_dt == np.float64 else 1e-3
        work = self.new_vectors(m)
        T = np.eye(m, dtype=wide)
        herm = lambda g: (g.astype(wide) + g.astype(wide).conj().T) / 2

        def right_multiply(block, q):                     # block <- block q (q square, host)
            k = block.nvec()
            work.select(k)
            block.multiply(np.ascontiguousarray(q.astype(self._dtype)), work)
            work.copy(block)

        def remove(block, basis, rows_block, rows_basis):  # block -= basis (basis^H block), T follows
            for _ in range(2):
                c = block.dot(basis).astype(wide)
                block.add(basis, -1.0, c.astype(self._dtype))
                T[rows_basis, :] += c @ T[rows_block, :]

        def orthonormalise(block, rows):                   # two Cholesky-QR passes; False: not positive definite
            for _ in range(2):
                g = herm(block.dot(block))
                try:
                    r = sla.cholesky(g, lower=False)
                except sla.LinAlgError:
                    return False
                right_multiply(block, sla.solve_triangular(r, np.eye(r.shape[0], dtype=wide), lower=False))
                if rows is not None:
                    T[rows, :] = r @ T[rows, :]
            return True

        done, top = 0, None
        fin, act, new = self.reference(), self.reference(), self.reference()
        while done < m:
            a = m - done
            act.select(a, f0 + done)
            lam, P = sla.eigh(herm(act.dot(act)))
            lam, P = lam[::-1], P[:, ::-1]
            if top is None:
                top = max(float(lam[0]), np.finfo(np.float64).tiny)
            if lam[0] <= (8 * eps) ** 2 * m * top:
                # nothing but rounding noise left: these directions do not belong to the block's range
                T[done:, :] = 0
                for _ in range(3):
                    act.fill_random()
                    if done > 0:
                        fin.select(done, f0)
                        for _ in range(2):
                            act.add(fin, -1.0, act.dot(fin))
                    if orthonormalise(act, None):
                        break
                done = m
                break
            k = max(1, int(np.sum(lam >= theta * lam[0])))
            right_multiply(act, P)
            T[done:, :] = P.conj().T @ T[done:, :]
            new.select(k, f0 + done)
            if done > 0:
                fin.select(done, f0)
                remove(new, fin, slice(done, done + k), slice(0, done))
            if not orthonormalise(new, slice(done, done + k)):
                # (cannot happen for columns within theta of each other; keep going on the eigen-scaling alone)
                d = np.sqrt(np.maximum(np.abs(np.real(new.dots(new))), np.finfo(np.float64).tiny)).astype(np.float64)
                new.scale(d.astype(real_dt))
                T[done:done + k, :] = d[:, None] * T[done:done + k, :]
            if k < a:
                act.select(a - k, f0 + done + k)
                remove(act, new, slice(done + k, m), slice(done, done + k))
            done += k
        self.select(m, f0)
        Ur, S, Vh = np.linalg.svd(T)
        work.select(m)
        self.multiply(np.ascontiguousarray(Ur.astype(self._dtype)), work)
        work.copy(self)
        # X = W S V^H; the reference hands back conj(v) of numpy's data = v S w, i.e. V
        return S.astype(real_dt), np.ascontiguousarray(Vh.conj().T.astype(self._dtype))


class ReductionBatch:
    """Several Gram / dots reductions evaluated with ONE host synchronisation (and, row-sharded,
    one all-reduce): the kernels write their results side by side into a device buffer that is
    fetched once by run().  The reference issues every `dot` / `dots` as its own blocking call
    (raleigh/core/solver.py:854-861, 1321-1339, 1376-1381, 1444-1447 are back-to-back pairs on
    shared operands): SURVEY 8(f).3 "fused Gram pairs".

    gram(rights, lefts) requests [lefts...]^H [rights...] (the blocks of each list side by side:
    one pass, every block read once), i.e. the stacked `r.dot(l)` for r in rights, l in lefts;
    dots(a, b) requests a.dots(b).  run() returns the results in request order."""

    def __init__(self, proto):
        self._proto = proto
        self._reqs = []

    def gram(self, rights, lefts):
        rights = [rights] if isinstance(rights, Vectors) else list(rights)
        lefts = [lefts] if isinstance(lefts, Vectors) else list(lefts)
        if not 1 <= len(rights) <= 4 or not 1 <= len(lefts) <= 4:
            raise ValueError('1 to 4 blocks per side')
        for v in rights + lefts:
            self._check(v)
        self._reqs.append(('gram', rights, lefts))
        return len(self._reqs) - 1

    def dots(self, a, b):
        self._check(a)
        self._check(b)
        if a.nvec() != b.nvec():
            raise ValueError('Numbers of vectors differ')
        self._reqs.append(('dots', a, b))
        return len(self._reqs) - 1

    def _check(self, v):
        """Every block of a batch is read with the prototype's element type and (local) length."""
        p = self._proto
        if v.data_type() != p.data_type():
            raise ValueError('Vectors data types differ')
        if v._vdim != p._vdim:
            raise ValueError('Vectors dimensions differ')

    def _buffer(self, nbytes):
        """Device buffer the results are written to (a communication buffer for row shards)."""
        lib = _lib.library()                   # (the buffer lives and dies with the library object)
        buf = getattr(lib, '_rlh_reduction_scratch', None)
        if buf is None or buf.nbytes < nbytes:
            buf = DeviceBuffer(max(nbytes, 1 << 16), zero=False)
            lib._rlh_reduction_scratch = buf
        return buf.ptr

    def _collect(self, ptr, count):
        """The `count` result elements at `ptr` as a host array: one synchronisation."""
        p = self._proto
        out = np.empty((count,), dtype=p.data_type())
        _lib.check(_lib.lib().rlh_fetch(_lib.host_ptr(out), ptr, out.nbytes))
        return out

    def run(self):
        import ctypes
        p = self._proto
        L = _lib.lib()
        es, code = p._es, p._code
        n = p._vdim
        shapes, count = [], 0
        for kind, a, b in self._reqs:
            if kind == 'gram':
                shp = (sum(v.nvec() for v in b), sum(v.nvec() for v in a))
            else:
                shp = (a.nvec(),)
            shapes.append((count, shp))
            count += int(np.prod(shp))
        if count == 0:
            return [np.zeros(shp, dtype=p.data_type()) for _, shp in shapes]
        base = self._buffer(count * es)
        for (kind, a, b), (off, shp) in zip(self._reqs, shapes):
            if int(np.prod(shp)) == 0:
                continue
            dst = base + off * es
            if kind == 'dots':
                _lib.check(L.rlh_dots(code, n, a.nvec(), a._ptr(), a._ld, b._ptr(), b._ld, dst, None))
            elif len(a) == 1 and len(b) == 1:
                _lib.check(L.rlh_gram(code, n, a[0].nvec(), a[0]._ptr(), a[0]._ld, b[0].nvec(), b[0]._ptr(), b[0]._ld,
                                      dst, None))
            else:
                def pack(vs):
                    vs = [v for v in vs if v.nvec() > 0]
                    ptrs = (ctypes.c_void_p * len(vs))(*[v._ptr() for v in vs])
                    lds = np.array([v._ld for v in vs], dtype=np.int64)
                    ms = np.array([v.nvec() for v in vs], dtype=np.int64)
                    return len(vs), ptrs, lds, ms
                nx, xp, xl, xm = pack(a)
                ny, yp, yl, ym = pack(b)
                _lib.check(L.rlh_gram_multi(code, n, nx, xp, _lib.host_ptr(xl), _lib.host_ptr(xm), ny, yp,
                                            _lib.host_ptr(yl), _lib.host_ptr(ym), dst, None))
        flat = self._collect(base, count)
        return [flat[off:off + int(np.prod(shp))].reshape(shp).copy() for off, shp in shapes]


def _is_matrix(arg):
    from .matrix import Matrix
    return isinstance(arg, Matrix)
