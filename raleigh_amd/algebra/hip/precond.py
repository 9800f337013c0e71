"""Device-resident preconditioners.

The reference's preconditioner is MKL's ILUT applied on the host with two triangular solves
per vector (raleigh/algebra/mkl_wrap.py:279-347).  `IncompleteLU` is its device counterpart:
the same dual-threshold factorisation on the host once, the triangular solves on the whole
block in HBM (`TriangularChain`: one persistent launch per application).  `ChebyshevPreconditioner`
is what the survey lists as the alternative: a
fixed polynomial p(A) ~ A^-1 (Chebyshev semi-iteration on [lo, hi]) built only from the
operator's ``apply`` and the block operations, so every block stays in HBM.  p(A) is
symmetric positive definite for a positive definite A (0 < 1 - r(x) on (0, hi], r the
Chebyshev residual polynomial), as the solver requires of a preconditioner
(raleigh/interfaces/partial_hevp.py:41-49).  It changes iteration counts relative to ILU, so
runs using it are reported separately from the reference's.
"""

import ctypes

import numpy as np

from ... import _lib


class TriangularChain:
    """X = T_k^-1 ... T_1^-1 B on the device for sparse triangular factors given as SciPy matrices
    (one persistent launch over the whole n x m block: rlh_sptrsv_create / rlh_sptrsv_solve_chain).

    factors: list of (matrix, lower, unit_diag); a unit-diagonal factor must not store its diagonal.
    perm_in / perm_out: optional row permutations (row r of the internal block is row perm_in[r] of
    B, and is written to row perm_out[r] of X)."""

    def __init__(self, factors, dtype, perm_in=None, perm_out=None):
        import scipy.sparse as scs
        L = self._L = _lib.lib()                          # (the handles belong to THIS library object)
        self._dtype = np.dtype(dtype).type
        self._code = _lib.dtype_code(self._dtype)
        self._ops = []
        self._n = None
        self.levels, self.nnz = [], []
        prepared = []
        for mat, lower, unit in factors:
            a = scs.csr_matrix(mat, dtype=self._dtype)
            a.sort_indices()
            n = a.shape[0]
            if self._n not in (None, n) or a.shape[0] != a.shape[1]:
                raise ValueError('the triangular factors must be square and of one size')
            self._n = n
            prepared.append((np.ascontiguousarray(a.indptr, dtype=np.int64), np.ascontiguousarray(a.indices, dtype=np.int32),
                             np.ascontiguousarray(a.data), 1 if lower else 0, 1 if unit else 0))

        def create(args):
            indptr, indices, values, lower, unit = args
            h = ctypes.c_void_p()
            rc = L.rlh_sptrsv_create(ctypes.byref(h), self._code, self._n, _lib.host_ptr(indptr), _lib.host_ptr(indices),
                                     _lib.host_ptr(values), lower, unit)
            err = None
            if rc != 0:                                    # (the library's error text is per thread: read where it was set)
                try:
                    _lib.check(rc)
                except _lib.RlhError as e:
                    err = e
            return err, h
        # the host side of rlh_sptrsv_create (diagonal-block transform, levels, units) is serial per factor and the factors of
        # a chain are independent: large ones are set up side by side (ctypes releases the GIL for the call; 0.67 -> 0.35 s for
        # the two ILUT factors of the config-3 surrogate)
        if len(prepared) > 1 and sum(len(p[1]) for p in prepared) > 2_000_000:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(prepared)) as ex:
                made = list(ex.map(create, prepared))
        else:
            made = [create(p) for p in prepared]
        for err, h in made:
            if h:
                self._ops.append(h)
        for err, h in made:
            if err is not None:
                raise err
        for h in self._ops:
            nnz, lev, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            _lib.check(L.rlh_sptrsv_info(h, ctypes.byref(nnz), ctypes.byref(lev), ctypes.byref(nb)))
            self.levels.append(int(lev.value))
            self.nnz.append(int(nnz.value))
        self._arr = (ctypes.c_void_p * len(self._ops))(*[h.value for h in self._ops])
        self._perms = []
        for perm in (perm_in, perm_out):
            if perm is None:
                self._perms.append(None)
                continue
            from .memory import DeviceBuffer
            host = np.ascontiguousarray(perm, dtype=np.int64)
            buf = DeviceBuffer(host.nbytes, zero=False)
            _lib.check(L.rlh_h2d(buf.ptr, _lib.host_ptr(host), host.nbytes))
            self._perms.append(buf)

    def __del__(self):
        ops, self._ops = getattr(self, '_ops', []), []
        for h in ops:
            try:
                self._L.rlh_sptrsv_destroy(h)
            except Exception:
                pass

    def size(self):
        return self._n

    def algorithmic_bytes(self, m):
        """Entries of the factors (value + 4-byte column index) + one read of B and one write of X."""
        es = np.dtype(self._dtype).itemsize
        return sum(self.nnz) * (es + 4) + 2 * self._n * m * es

    def solve(self, b, x):
        """x = chain^-1 b for Vectors windows of equal size (b may be x)."""
        m = b.nvec()
        if m != x.nvec():
            raise ValueError('Numbers of input and output vectors differ')
        if b.data_type() != self._dtype or x.data_type() != self._dtype:
            raise ValueError('Factors and vectors data types differ')
        n = getattr(b, 'local_dimension', b.dimension)()
        if n != self._n:
            raise ValueError('Factors and vectors dimensions incompatible')
        pin, pout = self._perms
        _lib.check(self._L.rlh_sptrsv_solve_chain(
            len(self._ops), self._arr, pin.ptr if pin else None, pout.ptr if pout else None, m,
            b.data_ptr(), b.ld(), x.data_ptr(), x.ld()))

    apply = solve


def ilut(matrix, tol=1e-6, maxfil=None):
    """Dual-threshold incomplete LU of a SciPy sparse matrix (rlh_ilut_factor: the host
    factorisation behind IncompleteLU; counterpart of mkl dcsrilut, mkl_wrap.py:305-331).
    Returns (L strictly lower with unit diagonal implied, U upper) as CSR in double / complex double."""
    import scipy.sparse as scs
    a = scs.csr_matrix(matrix)
    a.sort_indices()
    n = a.shape[0]
    cplx = np.iscomplexobj(a.data)
    dt = np.complex128 if cplx else np.float64
    if maxfil is None:
        maxfil = min(n - 1, a.nnz // max(n, 1))
    indptr = np.ascontiguousarray(a.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(a.indices, dtype=np.int32)
    values = np.ascontiguousarray(a.data, dtype=dt)
    L = _lib.library()                         # host-only entry points: no device needed
    f = ctypes.c_void_p()
    _lib.check(L.rlh_ilut_factor(ctypes.byref(f), _lib.DTYPE_CODE[dt], n, _lib.host_ptr(indptr), _lib.host_ptr(indices),
                                 _lib.host_ptr(values), float(tol), int(maxfil)))
    try:
        nl, nu = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.rlh_factors_nnz(f, ctypes.byref(nl), ctypes.byref(nu)))
        out = []
        for which, nnz in ((0, nl.value), (1, nu.value)):
            ip = np.zeros(n + 1, dtype=np.int64)
            ix = np.zeros(max(nnz, 1), dtype=np.int32)
            va = np.zeros(max(nnz, 1), dtype=dt)
            _lib.check(L.rlh_factors_get(f, which, _lib.host_ptr(ip), _lib.host_ptr(ix), _lib.host_ptr(va)))
            m = scs.csr_matrix((va[:nnz], ix[:nnz], ip), shape=(n, n))
            m.sort_indices()
            out.append(m)
    finally:
        L.rlh_factors_destroy(f)
    return out[0], out[1]


class IncompleteLU:
    """Incomplete LU preconditioner applied on the device (same surface as the reference's
    IncompleteLU, raleigh/algebra/sparse_mkl.py:122-140: construct from the matrix, `factorize(tol,
    max_fill)`, `apply(x, y)`).  The reference factorises with mkl dcsrilut and applies two
    mkl_dcsrtrsv per vector on the host; here the same dual-threshold ILUT runs once on the host
    (rlh_ilut_factor) and the two triangular solves run on the whole block in HBM."""

    def __init__(self, matrix):
        import scipy.sparse as scs
        self._a = scs.csr_matrix(matrix)
        self._a.sort_indices()
        self._chain = None

    def factorize(self, tol=1e-6, max_fill=1):
        n = self._a.shape[0]
        maxfil = min(n - 1, (self._a.nnz // n) * max_fill)       # mkl_wrap.py:306
        lo, up = ilut(self._a, tol, maxfil)
        self._chain = TriangularChain([(lo, True, True), (up, False, False)], self._a.dtype)
        self.fill = (lo.nnz + up.nnz) / float(self._a.nnz)
        self.levels = tuple(self._chain.levels)

    def chain(self):
        return self._chain

    def apply(self, x, y):
        if self._chain is None:
            self.factorize()
        self._chain.solve(x, y)


def gershgorin_upper_bound(matrix):
    """max_i sum_j |a_ij| >= lambda_max for a symmetric / Hermitian matrix (SciPy sparse)."""
    a = abs(matrix)
    return float(np.max(np.asarray(a.sum(axis=1)).ravel()))


class ChebyshevPreconditioner:

    def __init__(self, op, hi, ratio=50.0, degree=6, low_precision_op=None, storage=None):
        """op: operator with apply(x, y); hi: upper bound of its spectrum; the polynomial
        approximates 1/x on [hi / ratio, hi]; degree: number of operator applications.

        low_precision_op: the same operator in float32 / complex64.  When given, the polynomial
        is evaluated in that precision (half the bytes through the gather-bound SpMM); the input
        and the result are converted on the device.  A preconditioner only has to be a fixed,
        (nearly) symmetric positive definite approximation of A^-1, which single precision is."""
        if degree < 1:
            raise ValueError('degree must be at least 1')
        if storage not in (None, 'bf16'):
            raise ValueError("storage must be None or 'bf16'")
        # storage='bf16' (with a float32 low_precision_op in the windowed device layout): the
        # three work blocks are kept in bfloat16, arithmetic stays float32 -- half the bytes again
        # per step; on lap3d 64^3 (degree 24) the iteration count goes from 27 to 28.  Ignored
        # (float32 storage) where the operator cannot do it (sharded, sliced layout, complex).
        self._bf16 = storage == 'bf16'
        self._work16 = None
        self._op = op if low_precision_op is None else low_precision_op
        self._low = low_precision_op is not None
        self._lo, self._hi = float(hi) / float(ratio), float(hi)
        self._degree = int(degree)
        self._work = None

    def apply(self, x, y):
        """y = p(A) x by the three-term Chebyshev semi-iteration for A u = x from u_0 = 0:
        u_1 = x / theta,  u_{k+1} = u_k + rho_k rho_{k-1} (u_k - u_{k-1}) + (2 rho_k / delta) (x - A u_k).
        u_{k+1} overwrites u_{k-1} (only its own row is needed), so two blocks ping-pong and a
        fused step reads u_k, u_{k-1}, x and writes u_{k+1} (`cheb_step`)."""
        m = x.nvec()
        if self._bf16 and self._low and not x.is_complex() and getattr(self._op, 'supports_bf16', lambda: False)():
            try:
                self._apply_bf16(x, y, m)
                return
            except _lib.RlhError as e:
                # An unsharded operator whose layout the 2-byte staging cannot take after all: the same polynomial on
                # float32 work blocks.  On a row-sharded operator the error is final: supports_bf16() above is agreed on
                # by all ranks BEFORE any exchange is posted, and a rank falling back alone would break the exchange.
                if 'rlh_spmm_cheb_bf16' not in str(e) or hasattr(x, 'comm'):
                    raise
                import warnings
                warnings.warn('bfloat16 work blocks refused by the operator (%s): float32 work blocks instead' % e)
                self._bf16 = False
        nwork = 3 if self._low else 2
        # (capacity = shape()[0]: nvec() is the current selection and shrinks with the solver's block)
        if self._work is None or self._work[0].shape()[0] < m or self._work[0].dimension() != x.dimension():
            dt = None
            if self._low:
                dt = np.complex64 if x.is_complex() else np.float32
            self._work = [x.new_vectors(m, data_type=dt) for _ in range(nwork)]
        for v in self._work:
            v.select(m)
        if self._low:
            b, ua, ub = self._work
            x.convert_to(b)
        else:
            b = x
            # the last step must land in y: with an even number of steps u_1 starts in y
            steps = self._degree - 1
            ua, ub = (y, self._work[0]) if steps % 2 == 0 else (self._work[0], y)
            t = self._work[1]
        theta, delta = 0.5 * (self._hi + self._lo), 0.5 * (self._hi - self._lo)
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        fused = hasattr(self._op, 'cheb_step')
        if not fused and self._low:
            raise ValueError('low-precision evaluation needs an operator with cheb_step')
        ua.lincomb(1.0 / theta, b, 0.0, b)          # u_1 = x / theta
        if self._degree > 1:
            ub.zero()                               # u_0 = 0 (a defined value: 0 * garbage could be NaN)
        for _ in range(self._degree - 1):
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            c, cb = rho_new * rho, 2.0 * rho_new / delta
            if fused:       # ub = (1 + c) ua - c ub + cb (b - A ua) in one pass: u_{k+1} over u_{k-1}
                self._op.cheb_step(ua, ub, b, 1.0 + c, -c, cb)
            else:
                self._op.apply(ua, t)               # t = A u_k
                t.lincomb(-cb, t, cb, b)            # t = cb (b - A u_k)
                ub.lincomb(-c, ub, 1.0, t)
                ub.add(ua, 1.0 + c)
            ua, ub = ub, ua
            rho = rho_new
        if self._low:
            ua.convert_to(y)

    def _apply_bf16(self, x, y, m):
        from .sparse import Bf16Block
        n = x._vdim                                 # local rows (a row shard packs its own part)
        if self._work16 is None or self._work16[0].m < m or self._work16[0].n != n:
            self._work16 = [Bf16Block(n, m) for _ in range(3)]
        b, ua, ub = self._work16
        theta, delta = 0.5 * (self._hi + self._lo), 0.5 * (self._hi - self._lo)
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        b.pack(x, 1.0)
        ua.pack(x, 1.0 / theta)                     # u_1 = x / theta
        if self._degree > 1:
            ub.zero(m)                              # u_0 = 0
        for _ in range(self._degree - 1):
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            c, cb = rho_new * rho, 2.0 * rho_new / delta
            self._op.cheb_step_bf16(m, ua, ub, b, 1.0 + c, -c, cb)
            ua, ub = ub, ua
            rho = rho_new
        ua.unpack(y)
        self._work16 = [b, ua, ub]
