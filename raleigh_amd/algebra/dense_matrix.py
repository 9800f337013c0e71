"""Architecture-aware wrap for a dense matrix (raleigh/algebra/dense_matrix.py:10-64).

The only architecture this package provides is the MI355X one: ``arch`` must be
'hip' (aliases 'gpu', 'gpu!' are accepted so that reference call sites such as
``pca(A, arch='gpu!')`` keep working).  There is no CPU fallback here."""

import numpy


class _Device:
    def synchronize(self):
        from .. import _lib
        _lib.synchronize()


class AMatrix:

    def __init__(self, a, arch='hip', copy_data=False):
        if arch[:3] not in ('hip', 'gpu'):
            raise RuntimeError("raleigh_amd provides only arch='hip' (MI355X); got %r" % arch)
        from .hip import Matrix, Vectors
        self.__arch = arch
        self.__op = Matrix(a)
        self.__gpu = _Device()
        self.__Vectors = Vectors
        self.__vectors = None
        vmin = numpy.amin(a)
        vmax = numpy.amax(a)
        self.__scale = max(abs(vmin), abs(vmax))

    def as_operator(self):
        return self.__op

    def as_vectors(self):
        if self.__vectors is None:
            self.__vectors = self.__Vectors(self.__op, shallow=True)
        return self.__vectors

    def arch(self):
        return self.__arch

    def gpu(self):
        return self.__gpu

    def dots(self):
        return self.__op.dots()

    def frobenius2(self):
        return float(numpy.sum(numpy.abs(self.__op.dots())))

    def data_type(self):
        return self.__op.data_type()

    def shape(self):
        return self.__op.shape()

    def order(self):
        return self.__op.order()

    def scale(self):
        return self.__scale
