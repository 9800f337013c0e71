"""NumPy ``Vectors`` / ``Matrix`` with the reference's method surface.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  A CPU restatement of
the abstract-vectors contract of raleigh/core/solver.py:22-96 as implemented by
raleigh/algebra/dense_ndarray.py + dense_numpy.py, written on top of
``oracle.ops``.  Storage: one (capacity, dim) C-ordered array, a (first, nv)
selection window; every operation acts on the window.
"""

import numbers

import numpy as np

from . import ops


class Vectors:

    # ---- construction (dense_ndarray.py:52-83, dense_numpy.py:112-113)
    def __init__(self, arg, nvec=0, data_type=None, shallow=False):
        if isinstance(arg, Vectors):
            src = arg.data()
            self._a = src if shallow else src.copy()
        elif isinstance(arg, Matrix):
            a = arg.data()
            if not a.flags['C_CONTIGUOUS']:
                raise ValueError('Vectors data must be C_CONTIGUOUS')
            self._a = a if shallow else a.copy()
        elif isinstance(arg, np.ndarray):
            self._a = arg
        elif isinstance(arg, numbers.Number):
            dt = np.float64 if data_type is None else data_type
            self._a = np.zeros((int(nvec), int(arg)), dtype=dt)
        else:
            raise ValueError('wrong argument %s in constructor' % repr(type(arg)))
        self._sel = (0, self._a.shape[0])

    # ---- the 15 solver-facing methods (solver.py:22-96)
    def new_vectors(self, arg=0, dim=None):
        if isinstance(arg, np.ndarray):
            return Vectors(arg)
        return Vectors(self.dimension() if dim is None else dim, arg,
                       self.data_type())

    def dimension(self):
        return self._a.shape[1]

    def select(self, nv, first=0):
        assert nv <= self._a.shape[0] and first >= 0
        self._sel = (first, nv)

    def selected(self):
        return self._sel

    def clone(self):
        return Vectors(self)

    def append(self, other, axis=0):
        # dense_ndarray.py:39-47
        if axis == 0:
            self._a = np.concatenate((self.data(), other.data()))
            self.select_all()
        else:
            self._a = np.concatenate((self._a, other.all_data()), axis=1)

    def nvec(self):
        return self._sel[1]

    def data_type(self):
        return self._a.dtype.type

    def fill_random(self):
        # dense_ndarray.py:34-37
        f, k = self._sel
        self._a[f:f + k, :] = 2 * np.random.rand(k, self._a.shape[1]) - 1

    def copy(self, other, ind=None):
        if ind is None:
            assert other.nvec() == self.nvec()
            other.data()[:, :] = self.data()
        else:
            j = other.selected()[0]
            other.all_data()[j:j + len(ind), :] = ops.copy_cols(self._a, ind)

    def scale(self, s, multiply=False):
        self.data()[:, :] = ops.scale_cols(self.data(), s[:self.nvec()], multiply)

    def dots(self, other, transp=False):
        if transp:
            return ops.dots_transp(self.data(), other.data())
        return ops.dots(self.data(), other.data())

    def dot(self, other):
        return ops.gram(self.data(), other.data())

    def multiply(self, q, output):
        assert output.nvec() == q.shape[1]
        output.data()[:, :] = ops.multiply(self.data(), q)

    def add(self, other, s, q=None):
        if np.isscalar(s):
            if q is None:
                self.data()[:, :] = ops.axpy(self.data(), other.data(), s)
            else:
                self.data()[:, :] = ops.add_q(self.data(), other.data(), s, q)
        else:
            self.data()[:, :] = ops.axpy_cols(self.data(), other.data(),
                                              np.asarray(s)[:self.nvec()])

    # ---- extras used by the interfaces
    def select_all(self):
        self._sel = (0, self._a.shape[0])

    def is_complex(self):
        return self._a.dtype.kind == 'c'

    def zero(self):
        self.data()[:, :] = 0

    def fill(self, array_or_value):
        self.data()[:, :] = array_or_value

    def all_data(self):
        return self._a

    def data(self, i=None):
        f, k = self._sel
        return self._a[f:f + k, :] if i is None else self._a[f + i, :]

    def reference(self):
        return Vectors(self, shallow=True)

    def orthogonalize(self, other):
        new, q = ops.orthogonalize(self.data(), other.data())
        self.data()[:, :] = new
        return self.new_vectors(q)

    def svd(self):
        w, sigma, vh = ops.svd(self.data())
        self.data()[:, :] = w
        return sigma, vh


class Matrix:
    # dense_ndarray.py:117-151, dense_numpy.py:151-185

    def __init__(self, arg):
        a = arg.data() if isinstance(arg, Vectors) else arg
        if not isinstance(a, np.ndarray):
            raise ValueError('wrong argument %s in Matrix constructor'
                             % repr(type(arg)))
        if a.flags['C_CONTIGUOUS']:
            self._order = 'C_CONTIGUOUS'
        elif a.flags['F_CONTIGUOUS']:
            self._order = 'F_CONTIGUOUS'
        else:
            raise ValueError('Matrix data must be either C- or F-contiguous')
        self._a = a

    def data(self):
        return self._a

    def shape(self):
        return self._a.shape

    def data_type(self):
        return self._a.dtype.type

    def is_complex(self):
        return self._a.dtype.kind == 'c'

    def order(self):
        return self._order

    def apply(self, x, y, transp=False):
        y.data()[:, :] = ops.dense_apply(self._a, x.data(), transp)

    def dots(self):
        v = Vectors(self._a)
        return v.dots(v)

    def new_vectors(self, dim=None, nv=0):
        return Vectors(self._a.shape[1] if dim is None else dim, nv,
                       self.data_type())
