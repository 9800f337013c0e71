"""CPU oracle for the RALEIGH abstract-vectors hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``raleigh_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference's own NumPy backend (``raleigh/algebra/dense_numpy.py``), its MKL
backend where loadable, and its core solver in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks every
function here against those vectors.
"""

from . import ops  # noqa: F401
from .vectors import Vectors, Matrix  # noqa: F401
from .sparse import SparseSymmetricMatrix, lap3d  # noqa: F401
