"""A stand-in for librlhip.so that runs every C-ABI entry point on HOST memory
through the CPU oracle.  TEST INFRASTRUCTURE ONLY: it lets the CPU-only test
tier exercise the host logic of raleigh_amd (selection windows, strides,
padding, sharding, the block-JCG driver) without a GPU.  "Device" pointers are
plain host addresses.  The product never imports this module.
"""

import ctypes

import numpy as np
import scipy.sparse as sp

from oracle import ops

_DT = {0: np.float32, 1: np.float64, 2: np.complex64, 3: np.complex128}


def _addr(x):
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, ctypes.c_void_p):
        return x.value or 0
    if hasattr(x, 'value'):
        return int(x.value or 0)
    raise TypeError('cannot take the address of %r' % (x,))


def _flat(ptr, dtype, count):
    ptr = _addr(ptr)
    dtype = np.dtype(dtype)
    if count == 0:
        return np.zeros((0,), dtype=dtype)
    buf = (ctypes.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def _block(ptr, code, n, m, ld):
    """(m, n) view (one vector per row) of a column-major n x m block with leading dimension ld."""
    dt = np.dtype(_DT[code])
    if m == 0 or n == 0:
        return np.zeros((m, n), dtype=dt)
    flat = _flat(ptr, dt, (m - 1) * ld + n)
    return np.lib.stride_tricks.as_strided(flat, shape=(m, n), strides=(ld * dt.itemsize, dt.itemsize))


class _Csr:
    def __init__(self, mat, code):
        self.mat, self.code = mat, code


class FakeLib:
    def __init__(self):
        self._mem = {}
        self._err = b''
        self._csr = {}
        self._next_handle = 1
        self.calls = {}
        self.initialised = None
        self.bf16_refused = False          # (tests: a shard whose layout the bfloat16 staging cannot take)

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def _fail(self, msg):
        self._err = msg.encode()
        return 1

    # ---- context
    def rlh_version(self):
        return 100

    def rlh_last_error(self):
        return self._err

    def rlh_device_count(self, p):
        p._obj.value = 1
        return 0

    def rlh_init(self, device):
        self.initialised = device
        return 0

    def rlh_finalize(self):
        return 0

    def rlh_set_stream(self, s):
        return 0

    def rlh_sync(self):
        return 0

    def rlh_mem_info(self, f, t):
        return 0

    # ---- memory
    def rlh_malloc(self, pp, nbytes):
        buf = ctypes.create_string_buffer(max(int(nbytes), 1))
        addr = ctypes.addressof(buf)
        self._mem[addr] = buf
        pp._obj.value = addr
        return 0

    def rlh_free(self, p):
        self._mem.pop(_addr(p), None)
        return 0

    def rlh_memset(self, p, value, nbytes):
        ctypes.memset(_addr(p), value, int(nbytes))
        return 0

    def rlh_h2d(self, d, h, nbytes):
        if int(nbytes) > 4096:
            self._count('block_transfer')
        ctypes.memmove(_addr(d), _addr(h), int(nbytes))
        return 0

    rlh_d2h = rlh_h2d

    def rlh_d2d(self, d, s, nbytes):
        ctypes.memmove(_addr(d), _addr(s), int(nbytes))
        return 0

    def rlh_fetch(self, h, d, nbytes):
        self._count('fetch')
        self._count('sync')
        ctypes.memmove(_addr(h), _addr(d), int(nbytes))
        return 0

    def rlh_copy2d(self, dst, dpitch, src, spitch, width, rows, kind):
        if kind != 2:
            self._count('block_transfer')         # a block crossing the host / device boundary
        d, s = _addr(dst), _addr(src)
        for r in range(int(rows)):
            ctypes.memmove(d + r * dpitch, s + r * spitch, int(width))
        return 0

    # ---- reductions
    def rlh_gram(self, code, n, mx, X, ldx, my, Y, ldy, d_out, h_out):
        self._count('gram')
        if mx == 0 or my == 0:
            return 0
        if ldx < n or ldy < n:
            return self._fail('rlh_gram: leading dimension smaller than n')
        g = ops.gram(_block(X, code, n, mx, ldx), _block(Y, code, n, my, ldy)).astype(_DT[code])
        for out in (d_out, h_out):
            if _addr(out):
                _flat(out, _DT[code], my * mx)[:] = g.ravel()
        if _addr(h_out):
            self._count('sync')
        return 0

    def rlh_gram_multi(self, code, n, nx, X, ldx, mx, ny, Y, ldy, my, d_out, h_out):
        self._count('gram_multi')
        lx, mxs = _flat(ldx, np.int64, nx), _flat(mx, np.int64, nx)
        ly, mys = _flat(ldy, np.int64, ny), _flat(my, np.int64, ny)
        xs = np.concatenate([_block(X[k], code, n, int(mxs[k]), int(lx[k])) for k in range(nx)], axis=0)
        ys = np.concatenate([_block(Y[k], code, n, int(mys[k]), int(ly[k])) for k in range(ny)], axis=0)
        g = ops.gram(xs, ys).astype(_DT[code])
        for out in (d_out, h_out):
            if _addr(out):
                _flat(out, _DT[code], g.size)[:] = g.ravel()
        if _addr(h_out):
            self._count('sync')
        return 0

    def rlh_dots(self, code, n, m, X, ldx, Y, ldy, d_out, h_out):
        self._count('dots')
        if m == 0:
            return 0
        v = ops.dots(_block(X, code, n, m, ldx), _block(Y, code, n, m, ldy)).astype(_DT[code])
        for out in (d_out, h_out):
            if _addr(out):
                _flat(out, _DT[code], m)[:] = v
        if _addr(h_out):
            self._count('sync')
        return 0

    def rlh_dots_transp(self, code, n, m, X, ldx, Y, ldy, d_out):
        self._count('dots_transp')
        if n == 0:
            return 0
        w = ops.dots_transp(_block(X, code, n, m, ldx), _block(Y, code, n, m, ldy))
        _flat(d_out, _DT[code], n)[:] = w
        return 0

    # ---- updates
    def rlh_block_update(self, code, n, k, X, ldx, m, Out, ldo, q, q_rs, q_cs, alpha, beta):
        self._count('block_update')
        if n == 0 or m == 0:
            return 0
        dt = np.dtype(_DT[code])
        a = _flat(alpha, np.float64, 2)
        al = complex(a[0], a[1]) if dt.kind == 'c' else a[0]
        out = _block(Out, code, n, m, ldo)
        if k == 0:
            if not beta:
                out[:, :] = 0
            return 0
        span = (k - 1) * q_rs + (m - 1) * q_cs + 1
        qf = _flat(q, dt, span)
        qm = np.lib.stride_tricks.as_strided(qf, shape=(k, m),
                                             strides=(q_rs * dt.itemsize, q_cs * dt.itemsize))
        x = _block(X, code, n, k, ldx)
        upd = (al * qm).T @ x
        out[:, :] = (out + upd if beta else upd).astype(dt)
        return 0

    def rlh_block_update2(self, code, n, k1, X1, ldx1, q1, q1_rs, q1_cs, k2, X2, ldx2, q2, q2_rs, q2_cs, m, Out, ldo,
                          alpha, beta):
        self._count('block_update2')
        if n == 0 or m == 0:
            return 0
        dt = np.dtype(_DT[code])
        a = _flat(alpha, np.float64, 2)
        al = complex(a[0], a[1]) if dt.kind == 'c' else a[0]

        def qmat(q, k, rs, cs):
            qf = _flat(q, dt, (k - 1) * rs + (m - 1) * cs + 1)
            return np.lib.stride_tricks.as_strided(qf, shape=(k, m), strides=(rs * dt.itemsize, cs * dt.itemsize))
        upd = (al * qmat(q1, k1, q1_rs, q1_cs)).T @ _block(X1, code, n, k1, ldx1) \
            + (al * qmat(q2, k2, q2_rs, q2_cs)).T @ _block(X2, code, n, k2, ldx2)
        out = _block(Out, code, n, m, ldo)
        out[:, :] = (out + upd if beta else upd).astype(dt)
        return 0

    def rlh_block_update2x2(self, code, n, k1, X1, ldx1, q1, q1_rs, q1_cs, k2, X2, ldx2, q2, q2_rs, q2_cs, ma, OutA, ldoa,
                            mb, OutB, ldob):
        self._count('block_update2x2')
        if n == 0:
            return 0
        dt = np.dtype(_DT[code])
        m = ma + mb

        def qmat(q, k, rs, cs):
            qf = _flat(q, dt, (k - 1) * rs + (m - 1) * cs + 1)
            return np.lib.stride_tricks.as_strided(qf, shape=(k, m), strides=(rs * dt.itemsize, cs * dt.itemsize))
        upd = qmat(q1, k1, q1_rs, q1_cs).T @ _block(X1, code, n, k1, ldx1) \
            + qmat(q2, k2, q2_rs, q2_cs).T @ _block(X2, code, n, k2, ldx2)
        _block(OutA, code, n, ma, ldoa)[:, :] = upd[:ma].astype(dt)
        _block(OutB, code, n, mb, ldob)[:, :] = upd[ma:].astype(dt)
        return 0

    def rlh_lincomb_cols(self, code, n, m, a, A, lda, b, B, ldb, Out, ldo):
        self._count('lincomb_cols')
        if n == 0 or m == 0:
            return 0
        av, bv = _flat(a, _DT[code], m), _flat(b, _DT[code], m)
        res = av[:, None] * _block(A, code, n, m, lda) + bv[:, None] * _block(B, code, n, m, ldb)
        _block(Out, code, n, m, ldo)[:, :] = res.astype(_DT[code])
        return 0

    def rlh_axpy(self, code, n, m, alpha, X, ldx, Y, ldy):
        self._count('axpy')
        dt = np.dtype(_DT[code])
        a = _flat(alpha, np.float64, 2)
        al = complex(a[0], a[1]) if dt.kind == 'c' else a[0]
        y = _block(Y, code, n, m, ldy)
        y[:, :] = ops.axpy(y, _block(X, code, n, m, ldx), dt.type(al))
        return 0

    def rlh_axpy_cols(self, code, n, m, s, X, ldx, Y, ldy):
        self._count('axpy_cols')
        sv = _flat(s, _DT[code], m)
        y = _block(Y, code, n, m, ldy)
        y[:, :] = ops.axpy_cols(y, _block(X, code, n, m, ldx), sv)
        return 0

    def rlh_copy(self, code, n, m, X, ldx, Y, ldy):
        self._count('copy')
        _block(Y, code, n, m, ldy)[:, :] = _block(X, code, n, m, ldx)
        return 0

    def rlh_copy_cols(self, code, n, m, ind, Xall, ldx, Y, ldy):
        self._count('copy_cols')
        idx = _flat(ind, np.int64, m)
        if m == 0:
            return 0
        xa = _block(Xall, code, n, int(idx.max()) + 1, ldx)
        _block(Y, code, n, m, ldy)[:, :] = ops.copy_cols(xa, idx)
        return 0

    def rlh_scale_cols(self, code, n, m, s, mode, X, ldx):
        self._count('scale_cols')
        dt = np.dtype(_DT[code])
        sv = _flat(s, np.float64, 2 * m if dt.kind == 'c' else m)
        if dt.kind == 'c':
            sv = sv.view(np.complex128)
        x = _block(X, code, n, m, ldx)
        x[:, :] = ops.scale_cols(x, sv, bool(mode))
        return 0

    def rlh_convert(self, src, dst, n, m, X, ldx, Y, ldy):
        self._count('convert')
        _block(Y, dst, n, m, ldy)[:, :] = _block(X, src, n, m, ldx).astype(_DT[dst])
        return 0

    def rlh_fill_random(self, code, n, m, X, ldx, seed, row0, col0):
        self._count('fill_random')
        if n and m:
            _block(X, code, n, m, ldx)[:, :] = ops.uniform_block(int(seed), n, m, _DT[code], row0, col0)
        return 0

    def rlh_absmax(self, code, n, m, X, ldx, h_out):
        x = _block(X, code, n, m, ldx)
        val = 0.0 if x.size == 0 else max(float(np.max(np.abs(x.real))), float(np.max(np.abs(x.imag))))
        ctypes.cast(h_out, ctypes.POINTER(ctypes.c_double))[0] = val
        return 0

    def rlh_conj(self, code, n, m, X, ldx):
        if code in (2, 3):
            x = _block(X, code, n, m, ldx)
            x[:, :] = x.conj()
        return 0

    def rlh_gather_rows(self, code, nidx, d_idx, m, X, ldx, Out, ldo):
        self._count('gather_rows')
        if nidx == 0 or m == 0:
            return 0
        idx = _flat(d_idx, np.int64, nidx)
        x = _block(X, code, int(idx.max()) + 1, m, ldx)
        _block(Out, code, nidx, m, ldo)[:, :] = x[:, idx]
        return 0

    # ---- operators
    def rlh_csr_create(self, ph, code, n_rows, n_cols, indptr, indices, values):
        ip = _flat(indptr, np.int64, n_rows + 1).copy()
        nnz = int(ip[-1])
        ix = _flat(indices, np.int32, nnz).copy()
        va = _flat(values, _DT[code], nnz).copy()
        if nnz and (ix.min() < 0 or ix.max() >= n_cols):
            return self._fail('rlh_csr_create: column index out of range')
        h = self._next_handle
        self._next_handle += 1
        self._csr[h] = _Csr(sp.csr_matrix((va, ix, ip), shape=(n_rows, n_cols)), code)
        ph._obj.value = h
        return 0

    def rlh_csr_create_upper(self, ph, code, n, indptr, indices, values):
        ip = _flat(indptr, np.int64, n + 1).copy()
        nnz = int(ip[-1])
        ix = _flat(indices, np.int32, nnz).copy()
        va = _flat(values, _DT[code], nnz).copy()
        if nnz and (ix.min() < 0 or ix.max() >= n):
            return self._fail('rlh_csr_create_upper: column index out of range')
        a = sp.csr_matrix((va, ix, ip), shape=(n, n))
        u, s1 = sp.triu(a, format='csr'), sp.triu(a, k=1, format='csr')
        h = self._next_handle
        self._next_handle += 1
        self._csr[h] = _Csr(sp.csr_matrix(u + s1.conj().T), code)
        ph._obj.value = h
        return 0

    # ---- ILUT (host-only entry points of the real library) and triangular chains (SciPy)
    def _real(self):
        import os
        from raleigh_amd import _lib
        if not hasattr(self, '_dll'):
            self._dll = _lib._load()
        return self._dll

    def rlh_ilut_factor(self, *a):
        rc = self._real().rlh_ilut_factor(*a)
        if rc:
            self._err = self._real().rlh_last_error()
        return rc

    def rlh_factors_nnz(self, *a):
        return self._real().rlh_factors_nnz(*a)

    def rlh_factors_get(self, *a):
        return self._real().rlh_factors_get(*a)

    def rlh_factors_destroy(self, *a):
        return self._real().rlh_factors_destroy(*a)

    def rlh_shm_create(self, *a):
        rc = self._real().rlh_shm_create(*a)
        if rc:
            self._err = self._real().rlh_last_error()
        return rc

    def rlh_shm_unlink(self, *a):
        return self._real().rlh_shm_unlink(*a)

    def rlh_shm_allreduce(self, *a):
        rc = self._real().rlh_shm_allreduce(*a)
        if rc:
            self._err = self._real().rlh_last_error()
        return rc

    def rlh_shm_destroy(self, *a):
        return self._real().rlh_shm_destroy(*a)

    def rlh_ldlt_factor(self, *a):
        rc = self._real().rlh_ldlt_factor(*a)
        if rc:
            self._err = self._real().rlh_last_error()
        return rc

    def rlh_ldlt_info(self, *a):
        return self._real().rlh_ldlt_info(*a)

    def rlh_ldlt_get(self, *a):
        return self._real().rlh_ldlt_get(*a)

    def rlh_ldlt_get_transposed(self, *a):
        return self._real().rlh_ldlt_get_transposed(*a)

    def rlh_ldlt_destroy(self, *a):
        return self._real().rlh_ldlt_destroy(*a)

    def rlh_bdiag_solve(self, code, n, coef, shift, m, X, ldx):
        self._count('rlh_bdiag_solve')
        if n == 0 or m == 0:
            return 0
        c = _flat(coef, _DT[code], 2 * n).reshape(n, 2)
        s = _flat(shift, np.int32, n)
        x = _block(X, code, n, m, ldx)
        old = x.copy()
        x[:, :] = old * c[:, 0] + old[:, np.arange(n) + s] * np.where(s != 0, c[:, 1], 0)
        return 0

    def rlh_sptrsv_create(self, ph, code, n, indptr, indices, values, lower, unit):
        ip = _flat(indptr, np.int64, n + 1).copy()
        nnz = int(ip[-1])
        ix = _flat(indices, np.int32, nnz).copy()
        va = _flat(values, _DT[code], nnz).copy()
        mat = sp.csr_matrix((va, ix, ip), shape=(n, n))
        rows = np.repeat(np.arange(n), np.diff(ip))
        if unit and np.any(rows == ix):
            return self._fail('rlh_sptrsv_create: a unit-diagonal factor must not store its diagonal')
        if np.any((ix > rows) if lower else (ix < rows)):
            return self._fail('rlh_sptrsv_create: entry lies in the wrong triangle')
        if unit:
            mat = mat + sp.identity(n, dtype=_DT[code], format='csr')
        elif np.any(mat.diagonal() == 0):
            return self._fail('rlh_sptrsv_create: zero diagonal')
        h = self._next_handle
        self._next_handle += 1
        self._csr[h] = (sp.csr_matrix(mat), bool(lower), code, nnz - (0 if unit else n))
        ph._obj.value = h
        return 0

    def rlh_sptrsv_info(self, h, nnz, levels, nbytes):
        ctypes.cast(nnz, ctypes.POINTER(ctypes.c_int64))[0] = self._csr[_addr(h)][3]
        ctypes.cast(levels, ctypes.POINTER(ctypes.c_int64))[0] = 1
        ctypes.cast(nbytes, ctypes.POINTER(ctypes.c_int64))[0] = 0
        return 0

    def rlh_sptrsv_solve_chain(self, nops, ops, perm_in, perm_out, m, B, ldb, X, ldx):
        import scipy.sparse.linalg as sla
        self._count('rlh_sptrsv_solve_chain')
        handles = [int(ops[i]) for i in range(nops)]
        mat0, _, code, _ = self._csr[handles[0]]
        n = mat0.shape[0]
        w = _block(B, code, n, m, ldb).T.copy()
        if _addr(perm_in):
            w = w[_flat(perm_in, np.int64, n)]
        for hh in handles:
            mat, lower, _, _ = self._csr[hh]
            w = sla.spsolve_triangular(mat, w, lower=lower)
        out = _block(X, code, n, m, ldx)
        if _addr(perm_out):
            out[:, _flat(perm_out, np.int64, n)] = w.T
        else:
            out[:, :] = w.T
        return 0

    def rlh_sptrsv_destroy(self, h):
        self._csr.pop(_addr(h), None)
        return 0

    def rlh_csr_destroy(self, h):
        self._csr.pop(_addr(h), None)
        return 0

    def rlh_csr_info(self, h, a, b, c, d):
        return 0

    def rlh_csr_layout(self, h, layout, stored, ratio):
        # (the stand-in has no device layout; it reports the windowed one so that the host logic
        # that depends on it -- bfloat16 preconditioner storage -- is exercised on the CPU tier)
        ctypes.cast(layout, ctypes.POINTER(ctypes.c_int))[0] = 1
        ctypes.cast(stored, ctypes.POINTER(ctypes.c_int64))[0] = 0
        ctypes.cast(ratio, ctypes.POINTER(ctypes.c_double))[0] = 0.0
        return 0

    def rlh_csr_bf16_ready(self, h, n_own, ldh, ok):
        c = self._csr[_addr(h)]
        good = c.code == 0 and (n_own == c.mat.shape[1] or (n_own % 8 == 0 and ldh % 8 == 0)) and not self.bf16_refused
        ctypes.cast(ok, ctypes.POINTER(ctypes.c_int))[0] = 1 if good else 0
        return 0

    def rlh_csr_stacks(self, h, stacks, a, b):
        ctypes.cast(stacks, ctypes.POINTER(ctypes.c_int64))[0] = 0
        ctypes.cast(a, ctypes.POINTER(ctypes.c_double))[0] = 0.0
        ctypes.cast(b, ctypes.POINTER(ctypes.c_double))[0] = 0.0
        return 0

    def _rows_of_part(self, c, part, n_own):
        """Row mask of rlh_spmm_part (granularity here: single rows; the library uses 1024-row blocks)."""
        nr = c.mat.shape[0]
        if part == 0:
            return np.ones(nr, dtype=bool)
        csr = c.mat.tocsr()
        far = np.zeros(nr, dtype=bool)
        rows = np.repeat(np.arange(nr), np.diff(csr.indptr))
        far[rows[csr.indices >= n_own]] = True
        return ~far if part == 1 else far

    def rlh_spmm(self, h, m, X, ldx, n_own, H, ldh, Y, ldy):
        return self.rlh_spmm_part(h, 0, m, X, ldx, n_own, H, ldh, Y, ldy)

    def rlh_spmm_part(self, h, part, m, X, ldx, n_own, H, ldh, Y, ldy):
        self._count('spmm' if part == 0 else 'spmm_part%d' % part)
        c = self._csr[_addr(h)]
        nr, ncol = c.mat.shape
        if m == 0 or nr == 0:
            return 0
        x = np.zeros((m, ncol), dtype=_DT[c.code])
        x[:, :n_own] = _block(X, c.code, n_own, m, ldx)
        if ncol > n_own and part != 1:               # part 1 must not depend on the halo block
            x[:, n_own:] = _block(H, c.code, ncol - n_own, m, ldh)
        mask = self._rows_of_part(c, part, n_own)
        y = _block(Y, c.code, nr, m, ldy)
        y[:, mask] = ((c.mat @ x.T).T)[:, mask]
        return 0

    def rlh_spmm_cheb(self, h, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb):
        return self.rlh_spmm_cheb_part(h, 0, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb)

    def rlh_spmm_cheb_part(self, h, part, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb):
        self._count('spmm_cheb' if part != 1 else 'spmm_cheb_part1')
        c = self._csr[_addr(h)]
        nr, ncol = c.mat.shape
        if m == 0 or nr == 0:
            return 0
        x = np.zeros((m, ncol), dtype=_DT[c.code])
        x[:, :n_own] = _block(Y, c.code, n_own, m, ldy)
        if ncol > n_own and part != 1:
            x[:, n_own:] = _block(H, c.code, ncol - n_own, m, ldh)
        t = (c.mat @ x.T).T
        mask = self._rows_of_part(c, part, n_own)
        pv = _block(P, c.code, nr, m, ldp)
        new = cy * _block(Y, c.code, nr, m, ldy) + cp * pv + cb * (_block(B, c.code, nr, m, ldb) - t)
        pv[:, mask] = new[:, mask]
        return 0

    @staticmethod
    def _bf16_block(ptr, n, m, ld):
        flat = _flat(ptr, np.uint16, (m - 1) * ld + n)
        return np.lib.stride_tricks.as_strided(flat, shape=(m, n), strides=(ld * 2, 2))

    def rlh_bf16_pack(self, code, n, m, X, ldx, scale, Y16, ldy):
        if n and m:
            x = _block(X, code, n, m, ldx)
            self._bf16_block(Y16, n, m, ldy)[:, :] = ops.bf16_bits(np.float32(scale) * x.astype(np.float32))
        return 0

    def rlh_bf16_unpack(self, code, n, m, X16, ldx, Y, ldy):
        if n and m:
            _block(Y, code, n, m, ldy)[:, :] = ops.bf16_from_bits(self._bf16_block(X16, n, m, ldx)).astype(_DT[code])
        return 0

    def rlh_spmm_cheb_bf16(self, h, m, Y16, ldy, P16, ldp, B16, ldb, cy, cp, cb):
        ncol = self._csr[_addr(h)].mat.shape[1]
        return self.rlh_spmm_cheb_bf16_part(h, 0, m, Y16, ldy, ncol, None, 0, P16, ldp, B16, ldb, cy, cp, cb)

    def rlh_spmm_cheb_bf16_part(self, h, part, m, Y16, ldy, n_own, H16, ldh, P16, ldp, B16, ldb, cy, cp, cb):
        self._count('spmm_cheb_bf16')
        c = self._csr[_addr(h)]
        nr, ncol = c.mat.shape
        if m == 0 or nr == 0:
            return 0
        x = np.zeros((m, ncol), dtype=np.float32)
        x[:, :n_own] = ops.bf16_from_bits(self._bf16_block(Y16, n_own, m, ldy))
        if ncol > n_own and part != 1:
            x[:, n_own:] = ops.bf16_from_bits(self._bf16_block(H16, ncol - n_own, m, ldh))
        y = x[:, :nr]
        pv = self._bf16_block(P16, nr, m, ldp)
        b = ops.bf16_from_bits(self._bf16_block(B16, nr, m, ldb))
        t = (c.mat.astype(np.float32) @ x.T).T
        new = np.float32(cy) * y + np.float32(cp) * ops.bf16_from_bits(pv) + np.float32(cb) * (b - t)
        mask = self._rows_of_part(c, part, n_own)
        pv[:, mask] = ops.bf16_bits(new)[:, mask]
        return 0

    def rlh_gather_rows_bf16(self, nidx, d_idx, m, X16, ldx, Out16, ldo):
        self._count('gather_rows')
        if nidx == 0 or m == 0:
            return 0
        idx = _flat(d_idx, np.int64, nidx)
        x = self._bf16_block(X16, int(idx.max()) + 1, m, ldx)
        self._bf16_block(Out16, nidx, m, ldo)[:, :] = x[:, idx]
        return 0

    def rlh_dense_apply(self, code, M, N, A, lda, order, transp, m, X, ldx, Y, ldy):
        return self.rlh_dense_apply_r1(code, M, N, A, lda, order, transp, m, X, ldx, Y, ldy, None, None)

    def rlh_dense_apply_r1(self, code, M, N, A, lda, order, transp, m, X, ldx, Y, ldy, d_u, d_c):
        self._count('dense_apply')
        if order == 0:
            a = _block(A, code, N, M, lda)              # rows of A, shape (M, N)
        else:
            a = _block(A, code, M, N, lda).T            # columns of A, transposed view (M, N)
        nx, ny = (M, N) if transp else (N, M)
        if ldx < nx or ldy < ny:
            return self._fail('rlh_dense_apply: Matrix and vectors dimensions incompatible')
        x = _block(X, code, nx, m, ldx)
        y = ops.dense_apply(a, x, bool(transp))
        if _addr(d_c):
            c = _flat(d_c, _DT[code], m)
            u = _flat(d_u, _DT[code], ny) if _addr(d_u) else np.ones(ny, dtype=_DT[code])
            y = y - c[:, None] * u[None, :]
        _block(Y, code, ny, m, ldy)[:, :] = y
        return 0

    def rlh_timer_start(self):
        return 0

    def rlh_timer_stop(self, p):
        p._obj.value = 0.0
        return 0


def install():
    """Installs a fresh FakeLib as raleigh_amd's library; returns it."""
    from raleigh_amd import _lib
    fake = FakeLib()
    _lib.set_library(fake)
    return fake


def uninstall():
    from raleigh_amd import _lib
    _lib.set_library(None)
