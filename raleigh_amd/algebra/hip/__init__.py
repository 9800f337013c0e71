"""raleigh.algebra.hip -- MI355X-native abstract-vectors backend.

``Vectors`` / ``Matrix`` follow raleigh/algebra/dense_cublas.py's surface,
``SparseSymmetricMatrix`` raleigh/algebra/sparse_mkl.py's; see INTEGRATION.md
for the two-line change that plugs them into the reference.
"""

from ..._lib import synchronize, RlhError  # noqa: F401
from .vectors import Vectors  # noqa: F401
from .matrix import Matrix  # noqa: F401
from .sparse import SparseSymmetricMatrix, Operator, CsrOperator  # noqa: F401
