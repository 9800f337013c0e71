"""Dense data matrix bound to the MI355X backend.

Plays the role of the reference's architecture switch (raleigh/algebra/dense_matrix.py:10-64:
``AMatrix(a, arch)`` hands the interfaces an operator, a Vectors view of the rows and a device
handle).  This package has exactly one architecture, so there is nothing to switch: ``arch`` must
name the GPU ('hip', or the reference's 'gpu' / 'gpu!' so existing call sites keep working) and
a missing library or device raises -- there is no CPU fallback.
"""

import numpy as np

_ACCEPTED = ('hip', 'gpu')


class _DeviceHandle:
    """What ``AMatrix.gpu()`` returns: the interfaces only call ``synchronize()`` on it
    (raleigh/interfaces/partial_svd.py:288-289)."""

    @staticmethod
    def synchronize():
        from .. import _lib
        _lib.synchronize()


class AMatrix:
    """A host ndarray uploaded once; ``as_operator()`` is the device Matrix, ``as_vectors()`` a
    shallow Vectors view of its rows."""

    def __init__(self, a, arch='hip', copy_data=False):
        if str(arch)[:3] not in _ACCEPTED:
            raise RuntimeError("raleigh_amd provides only arch='hip' (MI355X); got %r" % (arch,))
        from .hip import Matrix
        self._arch = arch
        self._matrix = Matrix(a)                 # the upload is the copy: copy_data is moot
        self._rows = None
        self._magnitude = None                   # found on the device when first asked for

    # -- what the interfaces ask for
    def as_operator(self):
        return self._matrix

    def as_vectors(self):
        if self._rows is None:
            from .hip import Vectors
            self._rows = Vectors(self._matrix, shallow=True)
        return self._rows

    def gpu(self):
        return _DeviceHandle

    def arch(self):
        return self._arch

    # -- pass-throughs to the operator
    def shape(self):
        return self._matrix.shape()

    def order(self):
        return self._matrix.order()

    def data_type(self):
        return self._matrix.data_type()

    def dots(self):
        """Squared norms of the rows."""
        return self._matrix.dots()

    def frobenius2(self):
        """Squared Frobenius norm (sum of the squared row norms)."""
        return float(np.sum(np.abs(self.dots())))

    def scale(self):
        """Largest entry in modulus of the host data (used to scale error estimates)."""
        if self._magnitude is None:
            self._magnitude = self._matrix.absmax()
        return self._magnitude
