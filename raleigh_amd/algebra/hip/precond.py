"""Device-resident polynomial preconditioner.

The reference's preconditioner is MKL's ILUT applied on the host with two triangular solves
per vector (raleigh/algebra/mkl_wrap.py:279-347); a device counterpart of the same
factorisation is future work.  This class offers what the survey lists as the alternative: a
fixed polynomial p(A) ~ A^-1 (Chebyshev semi-iteration on [lo, hi]) built only from the
operator's ``apply`` and the block operations, so every block stays in HBM.  p(A) is
symmetric positive definite for a positive definite A (0 < 1 - r(x) on (0, hi], r the
Chebyshev residual polynomial), as the solver requires of a preconditioner
(raleigh/interfaces/partial_hevp.py:41-49).  It changes iteration counts relative to ILU, so
runs using it are reported separately from the reference's.
"""

import numpy as np


def gershgorin_upper_bound(matrix):
    """max_i sum_j |a_ij| >= lambda_max for a symmetric / Hermitian matrix (SciPy sparse)."""
    a = abs(matrix)
    return float(np.max(np.asarray(a.sum(axis=1)).ravel()))


class ChebyshevPreconditioner:

    def __init__(self, op, hi, ratio=50.0, degree=6, low_precision_op=None):
        """op: operator with apply(x, y); hi: upper bound of its spectrum; the polynomial
        approximates 1/x on [hi / ratio, hi]; degree: number of operator applications.

        low_precision_op: the same operator in float32 / complex64.  When given, the polynomial
        is evaluated in that precision (half the bytes through the gather-bound SpMM); the input
        and the result are converted on the device.  A preconditioner only has to be a fixed,
        (nearly) symmetric positive definite approximation of A^-1, which single precision is."""
        if degree < 1:
            raise ValueError('degree must be at least 1')
        self._op = op if low_precision_op is None else low_precision_op
        self._low = low_precision_op is not None
        self._lo, self._hi = float(hi) / float(ratio), float(hi)
        self._degree = int(degree)
        self._work = None

    def apply(self, x, y):
        m = x.nvec()
        if self._work is None or self._work[0].nvec() < m or self._work[0].dimension() != x.dimension():
            dt = None
            if self._low:
                dt = np.complex64 if x.is_complex() else np.float32
            self._work = [x.new_vectors(m, data_type=dt) for _ in range(5 if self._low else 3)]
        for v in self._work:
            v.select(m)
        r, d, t = self._work[:3]
        if self._low:
            xin, yout = self._work[3], self._work[4]
            x.convert_to(xin)
        else:
            xin, yout = x, y
        theta, delta = 0.5 * (self._hi + self._lo), 0.5 * (self._hi - self._lo)
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        xin.copy(r)                                 # r = x - A*0
        d.lincomb(1.0 / theta, xin, 0.0, xin)       # d = r / theta
        d.copy(yout)                                # y = d
        fused = hasattr(self._op, 'cheb_step')
        for _ in range(self._degree - 1):
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            if fused:       # r -= A d; dn = a d + b r; y += dn -- one pass, dn in the spare block
                self._op.cheb_step(d, r, t, yout, rho_new * rho, 2.0 * rho_new / delta)
                d, t = t, d
            else:
                self._op.apply(d, t)
                r.add(t, -1.0)                      # r -= A d
                d.lincomb(rho_new * rho, d, 2.0 * rho_new / delta, r)
                yout.add(d, 1.0)
            rho = rho_new
        self._work[:3] = [r, d, t]
        if self._low:
            yout.convert_to(y)
